// Detect head decode for gfx950 (reference models/yolo_test.py:43-65, eval branch).  HBM-bound, fp32 throughout.
// Built with fp contraction off so every expression rounds exactly like the reference's separate torch ops.
#pragma clang fp contract(off)
#include "icaf_common.h"
#include "stream_core.h"
#include <cstdlib>

namespace icaf {

struct Anchors { float v[16]; };   // up to 8 anchors (w, h) in pixels
int conv_prepare(const icaf_conv_args* a, ConvP& p);      // igemm.hip

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

// One thread per PIXEL (b, y, x) for the common heads (3 anchors; no = 6 / 8 / 14): the NA * NO conv outputs of the pixel are one contiguous
// run (8-byte loads), the index is taken apart once per pixel instead of once per element, and each anchor's z / raw row leaves as
// 8-byte stores.  Same expressions, same rounding as the element kernel above (which remains the general path).
template <int NA, int NO>
__global__ __launch_bounds__(256) void detect_pixel_kernel(const float* __restrict__ p, int ldp, float* __restrict__ z,
                                                           float* __restrict__ logits, float* __restrict__ raw, int B, int ny, int nx,
                                                           long long rows_total, long long row_offset, float stride, Anchors anc,
                                                           FastDiv dnx, FastDiv dny) {
    constexpr int NV = NA * NO;                            // floats per pixel (18 / 24 / 42), NV % 2 == 0
    const unsigned int total = (unsigned int)B * ny * nx;
    for (unsigned int pix = blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += gridDim.x * blockDim.x) {
        unsigned int t, x, b, y;
        fd_divmod(pix, dnx, t, x);
        fd_divmod(t, dny, b, y);
        const float* src = p + (long long)pix * ldp;
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; i += 2) {                 // 8-byte loads: the plan's conv output is dense (ldp = NV: 72 / 96 / 168-byte pixels)
            const float2 q = *(const float2*)(src + i);
            v[i] = q.x; v[i + 1] = q.y;
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const long long cell = ((long long)a * ny + y) * nx + x;
            const long long row = (long long)b * rows_total + row_offset + cell;
            float o[NO];
#pragma unroll
            for (int k = 0; k < NO; ++k) {
                const float sg = 1.0f / (1.0f + expf(-v[a * NO + k]));
                float out = sg;
                if (k == 0) out = ((sg * 2.0f - 0.5f) + (float)x) * stride;
                else if (k == 1) out = ((sg * 2.0f - 0.5f) + (float)y) * stride;
                else if (k == 2 || k == 3) {
                    const float d = sg * 2.0f;
                    out = (d * d) * anc.v[2 * a + (k - 2)];
                }
                o[k] = out;
            }
            float* zr = z + row * NO;
#pragma unroll
            for (int k = 0; k < NO; k += 2) *(float2*)(zr + k) = float2{o[k], o[k + 1]};
            if (raw) {
                float* rr = raw + (((long long)b * NA) * ny * nx + cell) * NO;
#pragma unroll
                for (int k = 0; k < NO; k += 2) *(float2*)(rr + k) = float2{v[a * NO + k], v[a * NO + k + 1]};
            }
            if (logits) {
#pragma unroll
                for (int k = 5; k < NO; ++k) logits[row * (NO - 5) + (k - 5)] = v[a * NO + k];
            }
        }
    }
}

template <int NA, int NO>
static int launch_detect_pixel(const float* p, int ldp, float* z, float* logits, float* raw, int B, int ny, int nx, long long rows_total,
                               long long row_offset, float stride, const Anchors& anc, hipStream_t s) {
    const long long pixels = (long long)B * ny * nx;
    long long blocks = (pixels + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL((detect_pixel_kernel<NA, NO>), dim3((unsigned)blocks), dim3(256), 0, s, p, ldp, z, logits, raw, B, ny, nx, rows_total,
                       row_offset, stride, anc, make_fastdiv((unsigned)nx), make_fastdiv((unsigned)ny));
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Detect level in ONE launch: the 1x1 output conv (models/yolo_test.py:50) as the persistent streaming GEMM of stream_core.h with
// the decode (:52-63) as its epilogue.  The conv's fp32 map never reaches HBM (it was written and read back: 2 x 14.7 MB at P3,
// batch 32) and a level is one launch instead of two.  The accumulator tile (+ bias) is staged as fp32 in the workgroup's LDS
// buffer — exactly the values the two-launch form stores — and one thread per (pixel, anchor) decodes its NO values with the
// expressions of detect_pixel_kernel above (this file is built with fp contraction off): z, logits and raw are bit-identical.
// ---------------------------------------------------------------------------------------------------------------
template <int NA, int NO>
struct DetectEpi {
    static constexpr int SO = 64 * 4 + 16;                 // fp32 staging rows of the 128 x 64 tile
    float* z; float* logits; float* raw;
    int M, ny, nx; long long rows_total, row_offset; float stride; Anchors anc; FastDiv dnx, dny;
    template <int TM>
    __device__ __forceinline__ void stage(const f32x16 (&acc)[TM], const f32x4 (&bq)[4], unsigned char* stg, int row0, int col0, int l31, int hi, int, int) const {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int nl = col0 + 8 * qd + 4 * hi;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int ml = row0 + b * 32 + l31;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[b][4 * qd + j] + bq[qd][j];          // (igemm's epilogue with ACT_NONE, alpha 1)
                *(f32x4*)(stg + ml * SO + nl * 4) = v;
            }
        }
    }
    __device__ __forceinline__ void flush(const unsigned char* stg, int m0, int n0, int tid) const {
        if (tid >= 128 * NA) return;                        // one thread per (pixel of the tile, anchor)
        const int row = tid / NA, a = tid - row * NA;
        const unsigned int pix = (unsigned int)(m0 + row);
        if ((int)pix >= M) return;
        unsigned int t, x, b, y;
        fd_divmod(pix, dnx, t, x);
        fd_divmod(t, dny, b, y);
        const float* src = (const float*)(stg + row * SO) + a * NO;
        float v[NO];
#pragma unroll
        for (int k = 0; k < NO; k += 2) { const float2 q2 = *(const float2*)(src + k); v[k] = q2.x; v[k + 1] = q2.y; }
        const long long cell = ((long long)a * ny + y) * nx + x;
        const long long zrow = (long long)b * rows_total + row_offset + cell;
        float o[NO];
#pragma unroll
        for (int k = 0; k < NO; ++k) {
            const float sg = 1.0f / (1.0f + expf(-v[k]));
            float out = sg;
            if (k == 0) out = ((sg * 2.0f - 0.5f) + (float)x) * stride;
            else if (k == 1) out = ((sg * 2.0f - 0.5f) + (float)y) * stride;
            else if (k == 2 || k == 3) {
                const float d = sg * 2.0f;
                out = (d * d) * anc.v[2 * a + (k - 2)];
            }
            o[k] = out;
        }
        float* zr = z + zrow * NO;
#pragma unroll
        for (int k = 0; k < NO; k += 2) *(float2*)(zr + k) = float2{o[k], o[k + 1]};
        if (raw) {
            float* rr = raw + (((long long)b * NA) * ny * nx + cell) * NO;
#pragma unroll
            for (int k = 0; k < NO; k += 2) *(float2*)(rr + k) = float2{v[k], v[k + 1]};
        }
        if (logits) {
#pragma unroll
            for (int k = 5; k < NO; ++k) logits[zrow * (NO - 5) + (k - 5)] = v[k];
        }
    }
};

template <int DT, int NA, int NO>
__global__ __launch_bounds__(512) void detect_conv_kernel(const ConvP p, const DetectEpi<NA, NO> epi) {
    stream_gemm<DT, 64, 1>(p, epi);
}

template <int DT, int NA, int NO>
static int launch_detect_conv(const ConvP& p, const DetectEpi<NA, NO>& epi, hipStream_t s) {
    constexpr int LDS = 3 * (128 + 64) * 128 + 128 * DetectEpi<NA, NO>::SO;
    ConvP q = p;
    q.mtiles = (p.M + 127) / 128;
    q.ntiles = 1;
    q.nchunks = p.K / 64;
    int dev = 0, cus = 256;
    ICAF_HIP(hipGetDevice(&dev));
    ICAF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = cus & ~7;                     // one workgroup per CU, 8 XCDs; workgroups beyond the pixel tiles exit at once
    ICAF_LDS_OPTIN((detect_conv_kernel<DT, NA, NO>), LDS);
    detect_conv_kernel<DT, NA, NO><<<dim3((unsigned)grid, 1, 1), dim3(512), LDS, s>>>(q, epi);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
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
    // per-pixel kernel: 3 anchors, even `no` in use, 8-byte aligned pixel runs and output rows
    const bool aligned = ldp % 2 == 0 && ((uintptr_t)p & 7) == 0 && ((uintptr_t)z & 7) == 0 && (!raw || ((uintptr_t)raw & 7) == 0);
    if (na == 3 && aligned && (long long)B * ny * nx < (1ll << 31) && !g_opt.detect_elementwise) {
        if (no == 6) return launch_detect_pixel<3, 6>(p, ldp, z, logits, raw, B, ny, nx, rows_total, row_offset, stride, anc, S(s));
        if (no == 8) return launch_detect_pixel<3, 8>(p, ldp, z, logits, raw, B, ny, nx, rows_total, row_offset, stride, anc, S(s));
        if (no == 14) return launch_detect_pixel<3, 14>(p, ldp, z, logits, raw, B, ny, nx, rows_total, row_offset, stride, anc, S(s));
    }
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

extern "C" int icaf_detect_conv(const icaf_conv_args* a, float* z, float* logits, float* raw, int na, int no, long long rows_total,
                                long long row_offset, float stride, const float* anchors_px, icaf_stream_t s) {
    if (!a || !z || !anchors_px) return fail(ICAF_ERR_ARG, "icaf_detect_conv: null pointer");
    ConvP p;
    int st = conv_prepare(a, p);
    if (st) return st;
    if (a->dtype == ICAF_F32) return fail(ICAF_ERR_UNSUPPORTED, "icaf_detect_conv: 16-bit feature maps (the fp32 build runs icaf_conv2d + icaf_detect_decode)");
    if (a->kh != 1 || a->kw != 1 || a->sh != 1 || a->sw != 1 || a->ph || a->pw || a->groups != 1 || a->res || a->pre || a->w2 || a->act != ICAF_ACT_NONE)
        return fail(ICAF_ERR_ARG, "icaf_detect_conv: a plain 1x1 convolution without activation is expected");
    if ((a->Cin * 2) % 128 || p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "icaf_detect_conv: Cin * 2 bytes must be a multiple of 128 (Cin = %d)", a->Cin);
    if (na != 3 || (no != 6 && no != 8 && no != 14) || a->Cout != na * no) return fail(ICAF_ERR_UNSUPPORTED, "icaf_detect_conv: built for 3 anchors, no in {6, 8, 14} (na = %d, no = %d, Cout = %d)", na, no, a->Cout);
    if (row_offset + (long long)na * a->Ho * a->Wo > rows_total) return fail(ICAF_ERR_ARG, "icaf_detect_conv: level does not fit in z");
    if (((uintptr_t)z & 7) || (raw && ((uintptr_t)raw & 7))) return fail(ICAF_ERR_ARG, "icaf_detect_conv: z / raw must be 8-byte aligned");
    if ((long long)a->B * a->Ho * a->Wo >= (1ll << 31)) return fail(ICAF_ERR_UNSUPPORTED, "icaf_detect_conv: too many pixels");
    Anchors anc;
    for (int i = 0; i < 16; ++i) anc.v[i] = i < 2 * na ? anchors_px[i] : 0.0f;
    const FastDiv dnx = make_fastdiv((unsigned)a->Wo), dny = make_fastdiv((unsigned)a->Ho);
#define ICAF_DETECT_CONV(NO_)                                                                                                              \
    {                                                                                                                                      \
        DetectEpi<3, NO_> epi{z, logits, raw, p.M, a->Ho, a->Wo, rows_total, row_offset, stride, anc, dnx, dny};                           \
        return a->dtype == ICAF_BF16 ? launch_detect_conv<ICAF_BF16, 3, NO_>(p, epi, S(s)) : launch_detect_conv<ICAF_F16, 3, NO_>(p, epi, S(s)); \
    }
    if (no == 6) ICAF_DETECT_CONV(6)
    if (no == 8) ICAF_DETECT_CONV(8)
    ICAF_DETECT_CONV(14)
#undef ICAF_DETECT_CONV
}
