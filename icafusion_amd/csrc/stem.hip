// Stem of both backbones for gfx950 (MI355X): input staging + the 6x6 / stride 2 / pad 2 convolution (+BN+SiLU) of yaml
// rows 0 and 10 in ONE persistent kernel (reference: detect_twostream.py:70-80 / test.py:116-123 for the staging,
// models/common.py:48-60 for the layer).
//
// The two-kernel form (icaf_preprocess_* then a 3x3 convolution over the space-to-depth tensor) writes and re-reads a
// 16-channel bf16 copy of the images (2 x 105 MB at batch 32) and spends most of its time in workgroup prologues: the
// layer has K = 144 and 25600 output patches.  Here
//   * a workgroup is PERSISTENT: it owns the folded weights (Cout x 144, resident in LDS for its whole life) and walks
//     patches p = blockIdx.x, += gridDim.x of 8 x 32 output pixels;
//   * the 10 x 34 space-to-depth halo patch of patch p+1 is fetched straight from the NCHW image (fp32, or the
//     dataloader's uint8 6-channel batch with the /255 fused) into REGISTERS while patch p is being multiplied, then
//     converted and written to the other LDS patch buffer — the image is read once, nothing intermediate reaches HBM;
//   * the arithmetic is ctile.hip's: same swizzled patch layout, same (tap, channel) K order and MFMA step, so the
//     result is bit-identical to the two-kernel form (tested);
//   * the epilogue is the shared one (bias + SiLU in registers, LDS-staged 16-byte NHWC stores).
#include "conv_common.h"
#include <cstring>

namespace icaf {

constexpr int ST_TH = 8, ST_TW = 32, ST_HH = ST_TH + 2, ST_HW = ST_TW + 2, ST_NIDX = ST_HH * ST_HW;      // 340 entries
constexpr int ST_PATCH_BYTES = ((ST_NIDX * 2 + 63) / 64) * 1024;                                        // 11 KiB

struct StemP {
    const void* img;            // fp32 [nstreams*B][3][H][W]   or   uint8 [B][ctot][H][W]
    int ctot;                   // uint8: channels per image (6); stream s reads channels [3s, 3s+3)
    int B, H, W, nstreams;
    int tiles_x, tiles_y, npatch;
    ConvP c;                    // y, bias, w, strides, Ho, Wo, Cout, ldy, Kp, alphas (the epilogue's view of the layer)
};

// 12 image values of one space-to-depth entry: (dy, dx, c) = 2x2 pixels x 3 channels; zero outside the image
template <bool U8>
__device__ __forceinline__ void load_entry(const StemP& q, int stream, int b, int gy, int gx, float (&v)[12]) {
    const bool in = (unsigned)gy < (unsigned)(q.H >> 1) && (unsigned)gx < (unsigned)(q.W >> 1);
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = 0.0f;
    if (!in) return;
    const long long plane = (long long)q.H * q.W;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const long long off = (long long)(2 * gy + dy) * q.W + 2 * gx;
            float a0, a1;
            if constexpr (U8) {
                const unsigned char* s = (const unsigned char*)q.img + ((long long)b * q.ctot + 3 * stream + c) * plane + off;
                const unsigned short two = *(const unsigned short*)s;          // W even, 2*gx even: 2-byte aligned
                a0 = (float)(two & 0xff) / 255.0f;
                a1 = (float)(two >> 8) / 255.0f;
            } else {
                const float* s = (const float*)q.img + ((long long)(stream * q.B + b) * 3 + c) * plane + off;
                const float2 two = *(const float2*)s;                          // 8-byte aligned
                a0 = two.x;
                a1 = two.y;
            }
            v[(dy * 2 + 0) * 3 + c] = a0;                                      // channel order (dy, dx, c) of ops.s2d_conv_weight
            v[(dy * 2 + 1) * 3 + c] = a1;
        }
}

template <int DT>
__device__ __forceinline__ void store_entry(unsigned char* patch, int idx, const float (&v)[12]) {
    float lo[8], hi4[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { lo[i] = v[i]; hi4[i] = i < 4 ? v[8 + i] : 0.0f; }
    const int key = (idx >> 3) & 1;                                            // ctile's swizzle for 2 slots per pixel
    *(u32x4*)(patch + (((idx << 1) + (0 ^ key)) << 4)) = pack16<DT>(lo);
    *(u32x4*)(patch + (((idx << 1) + (1 ^ key)) << 4)) = pack16<DT>(hi4);
}

template <int DT, int BN, bool U8>
__global__ __launch_bounds__(NTHREADS) void stem_kernel(const StemP q) {
    using E = Elem<DT>;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int RB = 128, TM = 2, TN = BN / 32, BM = ST_TH * ST_TW, WM = BM / 4;
    constexpr int NBW = BN / 32;
    constexpr int NSLICE = 3;                                                  // Kp = 192 elements = 3 slices of 128 bytes
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* patch0 = lds;
    unsigned char* patch1 = lds + ST_PATCH_BYTES;
    unsigned char* wbuf = lds + 2 * ST_PATCH_BYTES;                            // NSLICE x BN rows x 128 bytes
    unsigned char* stage = wbuf + NSLICE * BN * RB;                            // epilogue staging (aliasing it with the idle
    // patch buffers would allow 4 instead of 2 workgroups per CU but costs two barriers per patch: measured 288 vs 270 us)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ConvP& p = q.c;

    auto decode = [&](int pt, int& stream, int& b, int& y0, int& x0) {
        const int per_img = q.tiles_x * q.tiles_y, per_stream = per_img * q.B;
        stream = pt / per_stream;
        int r = pt - stream * per_stream;
        b = r / per_img;
        r -= b * per_img;
        const int ty = r / q.tiles_x;
        y0 = ty * ST_TH;
        x0 = (r - ty * q.tiles_x) * ST_TW;
    };
    // each thread stages entries tid and tid + 256 of a patch
    auto fetch = [&](int pt, float (&v0)[12], float (&v1)[12]) {
        int stream, b, y0, x0;
        decode(pt, stream, b, y0, x0);
        const int hy0 = tid / ST_HW, hx0 = tid - hy0 * ST_HW;
        load_entry<U8>(q, stream, b, y0 - 1 + hy0, x0 - 1 + hx0, v0);
        const int i1 = tid + NTHREADS;
        if (i1 < ST_NIDX) {
            const int hy1 = i1 / ST_HW, hx1 = i1 - hy1 * ST_HW;
            load_entry<U8>(q, stream, b, y0 - 1 + hy1, x0 - 1 + hx1, v1);
        }
    };
    auto commit = [&](unsigned char* patch, const float (&v0)[12], const float (&v1)[12]) {
        store_entry<DT>(patch, tid, v0);
        if (tid + NTHREADS < ST_NIDX) store_entry<DT>(patch, tid + NTHREADS, v1);
    };

    auto load_weights = [&](int stream) {
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const typename E::type*)p.w + stream * p.w_gs), 0, p.w_bytes, 0x00020000);
        const int rsub = lane >> 3, dkey = ((wave & 1) << 2) | (rsub >> 1), lslot = (lane & 7) ^ dkey;
        const unsigned w_off0 = ((unsigned)(wave * 8 + rsub) * (unsigned)p.Kp + (unsigned)(lslot * E::VEC)) * E::BYTES;
#pragma unroll
        for (int c = 0; c < NSLICE; ++c)
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                // (offset in a variable: a template-dependent expression passed inline as the builtin's offset operand
                // makes clang's host pass silently drop the kernel's stub)
                const unsigned voff = w_off0 + (unsigned)c * RB + (unsigned)(32 * i) * (unsigned)p.Kp * E::BYTES;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(wbuf + c * BN * RB + (wave + 4 * i) * 1024), 16, voff, 0, 0, 0);
            }
        wait_vmcnt<0>();
    };

    // XCD-aware patch walk: workgroups are dispatched round-robin over the 8 XCDs, each with a private L2.  XCD x walks
    // the x-th contiguous eighth of the patches, so patches that share halo rows / columns (and the 128-byte lines
    // their 8-byte-misaligned rows straddle) meet in ONE L2 instead of being fetched from HBM once per XCD.
    const int xcd = blockIdx.x & 7, per_xcd = (q.npatch + 7) >> 3, pstride = gridDim.x >> 3;
    const int pend = min((xcd + 1) * per_xcd, q.npatch);
    int pt = xcd * per_xcd + (blockIdx.x >> 3);
    if (pt >= pend) return;
    float v0[12], v1[12];
    fetch(pt, v0, v1);

    // fragment / lane constants (as ctile.hip, TW = 32, stride 1, 2 slots per pixel)
    int lbase[TM];
#pragma unroll
    for (int bb = 0; bb < TM; ++bb) lbase[bb] = (wave * TM + bb) * ST_HW + l31;
    const int fkey = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);

    int cur = 0, wstream = -1;
    commit(patch0, v0, v1);
    while (true) {
        int stream, b, y0, x0;
        decode(pt, stream, b, y0, x0);
        if (stream != wstream) {                   // (re)load this stream's weights (at most twice in a workgroup's life)
            __syncthreads();                       // nobody still reads the previous weights
            load_weights(stream);
            wstream = stream;
        }
        const int pn = pt + pstride;
        const bool more = pn < pend;
        if (more) fetch(pn, v0, v1);               // next patch's image reads stay in flight during the MFMAs below
        __syncthreads();                           // current patch (and weights) visible
        const unsigned char* patch = cur ? patch1 : patch0;
        f32x16 acc[TN][TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int bb = 0; bb < TM; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {        // K = (tap, 16 channels): one MFMA step per tap
            const int ky = tap / 3, kx = tap - 3 * ky, toff = ky * ST_HW + kx;
            u32x4 fp[TM], fw[TN];
#pragma unroll
            for (int bb = 0; bb < TM; ++bb) {
                const int idx = lbase[bb] + toff;
                fp[bb] = *(const u32x4*)(patch + (((idx << 1) + (hi ^ ((idx >> 3) & 1))) << 4));
            }
#pragma unroll
            for (int a = 0; a < TN; ++a) fw[a] = *(const u32x4*)(wbuf + (tap >> 2) * BN * RB + (a * 32) * RB + foff[tap & 3]);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int bb = 0; bb < TM; ++bb) mma_step<DT>(acc[a][bb], fw[a], fp[bb]);
        }
        epilogue<DT, DT, BM, BN, WM, BN, ICAF_ACT_SILU, false>(acc, stage, p, stream, [&](int row) {
            const int gy = y0 + (row >> 5), gx = x0 + (row & 31);
            return (gy < p.Ho && gx < p.Wo) ? (b * p.Ho + gy) * p.Wo + gx : -1;
        }, 0);
        if (!more) break;
        commit(cur ? patch0 : patch1, v0, v1);     // (every wave passed the epilogue's barrier: nobody reads that buffer)
        cur ^= 1;
        pt = pn;
    }
}

}  // namespace icaf

using namespace icaf;

template <int DT, int BN, bool U8>
static int launch_stem(const StemP& q, hipStream_t s) {
    constexpr int EB = Elem<DT>::BYTES;
    const int lds = 2 * ST_PATCH_BYTES + 3 * BN * 128 + ST_TH * ST_TW * (BN * EB + 16);
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int per_cu = (160 * 1024) / lds < 4 ? (160 * 1024) / lds : 4;
    int grid = cus * (per_cu < 1 ? 1 : per_cu);
    if (grid > q.npatch) grid = q.npatch;
    grid = (grid + 7) & ~7;                         // the patch walk is per XCD (8 of them)
    static bool attr = false;
    if (lds > 64 * 1024 && !attr) {
        ICAF_HIP(hipFuncSetAttribute((const void*)stem_kernel<DT, BN, U8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr = true;
    }
    stem_kernel<DT, BN, U8><<<dim3((unsigned)grid), dim3(NTHREADS), lds, s>>>(q);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

extern "C" int icaf_stem(const void* img, int img_u8, int ctot, const void* w, const float* bias, void* y, int ldy, int dtype, int nstreams,
                         int B, int H, int W, int Cout, int Kp, long long w_gs, long long bias_gs, long long y_gs, icaf_stream_t s) {
    if (!img || !w || !y) return fail(ICAF_ERR_ARG, "icaf_stem: null pointer");
    if (dtype != ICAF_BF16 && dtype != ICAF_F16) return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem: 16-bit types only");
    if ((H | W) & 1) return fail(ICAF_ERR_ARG, "icaf_stem: H and W must be even");
    if (Cout != 32 && Cout != 64) return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem: built for 32 or 64 output channels (got %d)", Cout);
    if (Kp != 192) return fail(ICAF_ERR_ARG, "icaf_stem: the packed space-to-depth weights must have Kp = 192 (K = 9 x 16), got %d", Kp);
    if (nstreams < 1 || nstreams > 2 || (img_u8 && ctot < 3 * nstreams)) return fail(ICAF_ERR_ARG, "icaf_stem: bad nstreams / ctot");
    if (ldy % 8 || ((uintptr_t)y & 15) || (y_gs * 2) % 16) return fail(ICAF_ERR_ARG, "icaf_stem: y must be 16-byte aligned with ldy %% 8 == 0");
    StemP q;
    memset(&q, 0, sizeof(q));
    q.img = img; q.ctot = ctot; q.B = B; q.H = H; q.W = W; q.nstreams = nstreams;
    ConvP& p = q.c;
    p.w = w; p.bias = bias; p.y = y; p.res = nullptr;
    p.w_gs = w_gs; p.bias_gs = bias_gs; p.y_gs = y_gs;
    p.B = B; p.H = H / 2; p.W = W / 2; p.Ho = H / 2; p.Wo = W / 2; p.Cin = 16; p.Cout = Cout; p.ldy = ldy; p.Kp = Kp;
    p.M = B * p.Ho * p.Wo; p.K = 144; p.act = ICAF_ACT_SILU;
    p.vec_y = 1; p.vec_r = 0;
    p.w_bytes = (unsigned)(128LL * Kp * 2);
    p.alpha_acc[0] = p.alpha_acc[1] = 1.0f; p.alpha_res[0] = p.alpha_res[1] = 0.0f;
    q.tiles_x = (p.Wo + ST_TW - 1) / ST_TW; q.tiles_y = (p.Ho + ST_TH - 1) / ST_TH;
    q.npatch = nstreams * B * q.tiles_x * q.tiles_y;
    hipStream_t hs = S(s);
    if (dtype == ICAF_BF16) {
        if (Cout == 32) return img_u8 ? launch_stem<ICAF_BF16, 32, true>(q, hs) : launch_stem<ICAF_BF16, 32, false>(q, hs);
        return img_u8 ? launch_stem<ICAF_BF16, 64, true>(q, hs) : launch_stem<ICAF_BF16, 64, false>(q, hs);
    }
    if (Cout == 32) return img_u8 ? launch_stem<ICAF_F16, 32, true>(q, hs) : launch_stem<ICAF_F16, 32, false>(q, hs);
    return img_u8 ? launch_stem<ICAF_F16, 64, true>(q, hs) : launch_stem<ICAF_F16, 64, false>(q, hs);
}
