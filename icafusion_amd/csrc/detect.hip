// Detect head decode for gfx950 (reference models/yolo_test.py:43-65, eval branch).  HBM-bound, fp32 throughout.
// Built with fp contraction off so every expression rounds exactly like the reference's separate torch ops.
#pragma clang fp contract(off)
#include "icaf_common.h"

namespace icaf {

struct Anchors { float v[16]; };   // up to 8 anchors (w, h) in pixels

// One thread per OUTPUT ELEMENT (b, anchor, y, x, o), o fastest: z, logits and raw are written fully coalesced (their
// rows are contiguous over (x, o)); the conv output is read in `no`-float runs at pixel stride ldp, each 128-byte line
// being reused by the na anchors out of L2.
//   z:      [B][rows_total][no]   rows ordered (anchor, y, x) per level, levels concatenated at row_offset
//   logits: [B][rows_total][no-5] raw class scores
//   raw:    [B][na][ny][nx][no]   pre-sigmoid map in the reference's permuted layout
struct DetectDiv { FastDiv no, nx, ny, na; };

// I32: the flat element index fits 31 bits (every configuration in use: 9.7 M elements at 1280 x 1280, batch 16, no = 14) and is
// taken apart with FastDiv; otherwise with 64-bit divisions.
template <bool I32>
__global__ __launch_bounds__(256) void detect_decode_kernel(const float* __restrict__ p, int ldp, float* __restrict__ z,
                                                            float* __restrict__ logits, float* __restrict__ raw, int B, int ny, int nx,
                                                            int na, int no, long long rows_total, long long row_offset, float stride,
                                                            Anchors anc, DetectDiv dv) {
    const long long total = (long long)B * na * ny * nx * no;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        int o, x, y, a, b;
        if constexpr (I32) {
            unsigned int q, r;
            fd_divmod((unsigned int)e, dv.no, q, r);       // q = cell index (b, a, y, x)
            o = (int)r;
            fd_divmod(q, dv.nx, q, r);
            x = (int)r;
            fd_divmod(q, dv.ny, q, r);
            y = (int)r;
            fd_divmod(q, dv.na, q, r);
            a = (int)r;
            b = (int)q;
        } else {
            o = (int)(e % no);
            const long long idx = e / no;
            x = (int)(idx % nx);
            long long t = idx / nx;
            y = (int)(t % ny);
            t /= ny;
            a = (int)(t % na);
            b = (int)(t / na);
        }
        const float v = p[(((long long)b * ny + y) * nx + x) * ldp + a * no + o];
        const long long row = (long long)b * rows_total + row_offset + ((long long)a * ny + y) * nx + x;
        if (raw) raw[e] = v;
        if (logits && o >= 5) logits[row * (no - 5) + (o - 5)] = v;
        const float sg = 1.0f / (1.0f + expf(-v));
        float out = sg;
        if (o == 0) out = ((sg * 2.0f - 0.5f) + (float)x) * stride;
        else if (o == 1) out = ((sg * 2.0f - 0.5f) + (float)y) * stride;
        else if (o == 2 || o == 3) {
            float an = anc.v[o - 2];                       // anchor (w, h) of `a`, selected without dynamic indexing
#pragma unroll
            for (int i = 1; i < 8; ++i) an = (a == i) ? (o == 2 ? anc.v[2 * i] : anc.v[2 * i + 1]) : an;
            const float d = sg * 2.0f;
            out = (d * d) * an;
        }
        z[row * no + o] = out;
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
    const long long cells = (long long)B * na * ny * nx * no;
    long long blocks = (cells + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (B < 1 || ny < 1 || nx < 1) return fail(ICAF_ERR_ARG, "icaf_detect_decode: empty level");
    const DetectDiv dv{make_fastdiv((unsigned)no), make_fastdiv((unsigned)nx), make_fastdiv((unsigned)ny), make_fastdiv((unsigned)na)};
    if (cells < (1ll << 31))
        hipLaunchKernelGGL(detect_decode_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, S(s), p, ldp, z, logits, raw, B, ny, nx, na,
                           no, rows_total, row_offset, stride, anc, dv);
    else
        hipLaunchKernelGGL(detect_decode_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, S(s), p, ldp, z, logits, raw, B, ny, nx, na,
                           no, rows_total, row_offset, stride, anc, dv);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}
