// Detect head decode for gfx950 (reference models/yolo_test.py:43-65, eval branch).  HBM-bound, fp32 throughout.
// Built with fp contraction off so every expression rounds exactly like the reference's separate torch ops.
#pragma clang fp contract(off)
#include "icaf_common.h"

namespace icaf {

struct Anchors { float v[16]; };   // up to 8 anchors (w, h) in pixels

// One thread per (b, anchor, y, x) cell; the `no` outputs of a cell are contiguous in the conv output.
//   z:      [B][rows_total][no]   rows ordered (anchor, y, x) per level, levels concatenated at row_offset
//   logits: [B][rows_total][no-5] raw class scores
//   raw:    [B][na][ny][nx][no]   pre-sigmoid map in the reference's permuted layout
__global__ __launch_bounds__(256) void detect_decode_kernel(const float* __restrict__ p, int ldp, float* __restrict__ z,
                                                            float* __restrict__ logits, float* __restrict__ raw, int B, int ny, int nx,
                                                            int na, int no, long long rows_total, long long row_offset, float stride,
                                                            Anchors anc) {
    const long long cells = (long long)B * na * ny * nx;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < cells; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % nx);
        long long t = idx / nx;
        const int y = (int)(t % ny);
        t /= ny;
        const int a = (int)(t % na), b = (int)(t / na);
        const float* src = p + (((long long)b * ny + y) * nx + x) * ldp + a * no;
        const long long row = row_offset + ((long long)a * ny + y) * nx + x;
        float* zr = z + ((long long)b * rows_total + row) * no;
        float* rr = raw ? raw + idx * no : nullptr;
        float* lr = logits ? logits + ((long long)b * rows_total + row) * (no - 5) : nullptr;
        const float aw = anc.v[2 * a], ah = anc.v[2 * a + 1];
        for (int o = 0; o < no; ++o) {
            const float v = src[o];
            if (rr) rr[o] = v;
            if (lr && o >= 5) lr[o - 5] = v;
            const float sg = 1.0f / (1.0f + expf(-v));
            float out = sg;
            if (o == 0) out = ((sg * 2.0f - 0.5f) + (float)x) * stride;
            else if (o == 1) out = ((sg * 2.0f - 0.5f) + (float)y) * stride;
            else if (o == 2) { const float d = sg * 2.0f; out = (d * d) * aw; }
            else if (o == 3) { const float d = sg * 2.0f; out = (d * d) * ah; }
            zr[o] = out;
        }
    }
}

}  // namespace icaf

using namespace icaf;

extern "C" int icaf_detect_decode(const float* p, int ldp, float* z, float* logits, float* raw, int B, int ny, int nx, int na, int no,
                                  long long rows_total, long long row_offset, float stride, const float* anchors_px, icaf_stream_t s) {
    if (!p || !z || !anchors_px) return fail(ICAF_ERR_ARG, "icaf_detect_decode: null pointer");
    if (na < 1 || na > 8 || no < 6 || ldp < na * no) return fail(ICAF_ERR_ARG, "icaf_detect_decode: bad na/no/ldp");
    if (row_offset + (long long)na * ny * nx > rows_total) return fail(ICAF_ERR_ARG, "icaf_detect_decode: level does not fit in z");
    Anchors anc;
    for (int i = 0; i < 16; ++i) anc.v[i] = i < 2 * na ? anchors_px[i] : 0.0f;
    const long long cells = (long long)B * na * ny * nx;
    long long blocks = (cells + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(detect_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, S(s), p, ldp, z, logits, raw, B, ny, nx, na, no,
                       rows_total, row_offset, stride, anc);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}
