// Bandwidth-bound layout / pooling kernels (HBM roofline): input staging, SPPF max pools, nearest upsample, channel
// copy.  All use 16-byte vectors along the contiguous NHWC channel axis; one thread = one (pixel, channel-vector).
#include "icaf_common.h"
#include <cstdlib>
#include <algorithm>
#include <cstdint>

namespace icaf {

// ---- NCHW fp32 image -> NHWC (mode 0) or space-to-depth NHWC (mode 1) ------------------------------------------
// Consecutive threads own consecutive output pixels of one row, so the per-channel-plane reads are coalesced along
// W; each thread writes whole 16-byte channel vectors of its pixel.
template <int DT>
__global__ __launch_bounds__(256) void preprocess_kernel(const float* __restrict__ img, typename Elem<DT>::type* __restrict__ out,
                                                         int B, int C, int H, int W, int Cpad, int mode) {
    using E = Elem<DT>;
    const int Ho = mode ? H / 2 : H, Wo = mode ? W / 2 : W;
    const long long npix = (long long)B * Ho * Wo;
    const int nv = Cpad / E::VEC;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(pix % Wo);
        const long long t = pix / Wo;
        const int ho = (int)(t % Ho), b = (int)(t / Ho);
        for (int v = 0; v < nv; ++v) {
            float f[E::VEC];
#pragma unroll
            for (int j = 0; j < E::VEC; ++j) {
                const int ch = v * E::VEC + j;
                float val = 0.0f;
                if (mode == 0) {
                    if (ch < C) val = img[(((long long)b * C + ch) * H + ho) * W + wo];
                } else if (ch < 4 * C) {
                    const int sub = ch / C, c = ch - sub * C, dy = sub >> 1, dx = sub & 1;
                    val = img[(((long long)b * C + c) * H + 2 * ho + dy) * W + 2 * wo + dx];
                }
                f[j] = val;
            }
            *(u32x4*)(out + pix * Cpad + v * E::VEC) = pack16<DT>(f);
        }
    }
}

// ---- uint8 NCHW (the dataloader's 6-channel RGB+IR batch, reference test.py:116-123) -> NHWC / space-to-depth NHWC --
// Fuses .to(device).float(), `/= 255.0` and the RGB / IR channel split: stream s of `nstreams` takes channels
// [c0 + s*C, c0 + (s+1)*C) of every image and lands in out[s][b] (a pair act), so one launch stages both backbones from
// a quarter of the bytes the fp32 path reads.  Division (not a reciprocal multiply) to round like the reference.
template <int DT>
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ img, typename Elem<DT>::type* __restrict__ out,
                                                            int B, int Ctot, int c0, int C, int nstreams, int H, int W, int Cpad, int mode) {
    using E = Elem<DT>;
    const int Ho = mode ? H / 2 : H, Wo = mode ? W / 2 : W;
    const long long npix = (long long)nstreams * B * Ho * Wo;
    const int nv = Cpad / E::VEC;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(pix % Wo);
        long long t = pix / Wo;
        const int ho = (int)(t % Ho);
        t /= Ho;
        const int b = (int)(t % B), st = (int)(t / B);
        const unsigned char* src = img + ((long long)b * Ctot + c0 + st * C) * H * W;
        for (int v = 0; v < nv; ++v) {
            float f[E::VEC];
#pragma unroll
            for (int j = 0; j < E::VEC; ++j) {
                const int ch = v * E::VEC + j;
                float val = 0.0f;
                if (mode == 0) {
                    if (ch < C) val = (float)src[((long long)ch * H + ho) * W + wo] / 255.0f;
                } else if (ch < 4 * C) {
                    const int sub = ch / C, c = ch - sub * C, dy = sub >> 1, dx = sub & 1;
                    val = (float)src[((long long)c * H + 2 * ho + dy) * W + 2 * wo + dx] / 255.0f;
                }
                f[j] = val;
            }
            *(u32x4*)(out + pix * Cpad + v * E::VEC) = pack16<DT>(f);
        }
    }
}


// ---- SPPF: y1 = maxpool_k(x), y2 = maxpool_k(y1), y3 = maxpool_k(y2), stride 1, -inf padding ----------------------
// Chained k-pools equal direct pools with windows k, 2k-1, 3k-2 clipped at the border, so one pass over the
// (3k-2)^2 neighbourhood produces all three outputs (the feature map is tiny and L2-resident).
template <int DT>
__global__ __launch_bounds__(256) void sppf_kernel(const typename Elem<DT>::type* __restrict__ x, int ldx,
                                                   typename Elem<DT>::type* __restrict__ y1, typename Elem<DT>::type* __restrict__ y2,
                                                   typename Elem<DT>::type* __restrict__ y3, int ldy, int B, int H, int W, int C, int k) {
    using E = Elem<DT>;
    const int nv = C / E::VEC;
    const long long total = (long long)B * H * W * nv;
    const int r1 = k / 2, r2 = 2 * r1, r3 = 3 * r1;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(idx % nv);
        const long long pix = idx / nv;
        const int w = (int)(pix % W);
        const long long t = pix / W;
        const int h = (int)(t % H), b = (int)(t / H);
        float m1[E::VEC], m2[E::VEC], m3[E::VEC];
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) m1[j] = m2[j] = m3[j] = -INFINITY;
        for (int dy = -r3; dy <= r3; ++dy) {
            const int hh = h + dy;
            if ((unsigned)hh >= (unsigned)H) continue;
            const int ady = dy < 0 ? -dy : dy;
            for (int dx = -r3; dx <= r3; ++dx) {
                const int ww = w + dx;
                if ((unsigned)ww >= (unsigned)W) continue;
                const int adx = dx < 0 ? -dx : dx;
                const int rad = ady > adx ? ady : adx;
                float f[E::VEC];
                unpack16<DT>(*(const u32x4*)(x + (((long long)b * H + hh) * W + ww) * ldx + v * E::VEC), f);
#pragma unroll
                for (int j = 0; j < E::VEC; ++j) {
                    m3[j] = fmaxf(m3[j], f[j]);
                    if (rad <= r2) m2[j] = fmaxf(m2[j], f[j]);
                    if (rad <= r1) m1[j] = fmaxf(m1[j], f[j]);
                }
            }
        }
        const long long o = pix * ldy + v * E::VEC;
        *(u32x4*)(y1 + o) = pack16<DT>(m1);
        *(u32x4*)(y2 + o) = pack16<DT>(m2);
        *(u32x4*)(y3 + o) = pack16<DT>(m3);
    }
}


// LDS version (used whenever a whole H x W plane of VPB channel-vectors fits): the chained pools are done exactly as
// the reference chains them — three rounds of a separable (horizontal, then vertical) k-max over the plane held in
// LDS — so each output costs 2k LDS reads instead of (3k-2)^2 global reads.
// 16-bit types never leave their packed form: a bf16 / f16 bit pattern b maps to the unsigned key
// b ^ (b & 0x8000 ? 0xffff : 0x8000), which orders like the value (-inf < ... < -0 < +0 < ... < +inf), so the max of
// eight channels is four `v_pk_max_u16` instead of eight conversions and eight fp32 max per neighbour; the plane is
// keyed once on the way in and un-keyed on the way out (max only ever selects one of its inputs: bit-exact).
__device__ __forceinline__ unsigned int key16x2(unsigned int b) {
    const unsigned int sign = b & 0x80008000u;
    return b ^ ((sign >> 15) * 0x7fffu | 0x80008000u);          // negative halves: ^0xffff, positive: ^0x8000
}
__device__ __forceinline__ u32x4 key_vec(u32x4 v) { return u32x4{key16x2(v[0]), key16x2(v[1]), key16x2(v[2]), key16x2(v[3])}; }
__device__ __forceinline__ unsigned int unkey16x2(unsigned int k) {
    const unsigned int pos = k & 0x80008000u;                     // keys of non-negative values have the top bit set
    return k ^ (((pos ^ 0x80008000u) >> 15) * 0x7fffu | 0x80008000u);
}
__device__ __forceinline__ u32x4 unkey_vec(u32x4 v) { return u32x4{unkey16x2(v[0]), unkey16x2(v[1]), unkey16x2(v[2]), unkey16x2(v[3])}; }
__device__ __forceinline__ unsigned int max_u16x2(unsigned int a, unsigned int b) {
    using u16x2 = __attribute__((ext_vector_type(2))) unsigned short;
    const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    const u16x2 m = __builtin_elementwise_max(x, y);
    return __builtin_bit_cast(unsigned int, m);
}

template <int DT> struct PlaneMax {        // element-wise max of two 16-byte vectors in the representation held in LDS
    static __device__ __forceinline__ u32x4 in(u32x4 v) { return key_vec(v); }
    static __device__ __forceinline__ u32x4 out(u32x4 v) { return unkey_vec(v); }
    static __device__ __forceinline__ u32x4 mx(u32x4 a, u32x4 b) {
        return u32x4{max_u16x2(a[0], b[0]), max_u16x2(a[1], b[1]), max_u16x2(a[2], b[2]), max_u16x2(a[3], b[3])};
    }
};
template <> struct PlaneMax<ICAF_F32> {
    static __device__ __forceinline__ u32x4 in(u32x4 v) { return v; }
    static __device__ __forceinline__ u32x4 out(u32x4 v) { return v; }
    static __device__ __forceinline__ u32x4 mx(u32x4 a, u32x4 b) {
        u32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __float_as_uint(fmaxf(__uint_as_float(a[j]), __uint_as_float(b[j])));
        return r;
    }
};

template <int DT>
__global__ __launch_bounds__(256) void sppf_lds_kernel(const typename Elem<DT>::type* __restrict__ x, int ldx,
                                                       typename Elem<DT>::type* __restrict__ y1, typename Elem<DT>::type* __restrict__ y2,
                                                       typename Elem<DT>::type* __restrict__ y3, int ldy, int H, int W, int C, int k,
                                                       int vpb) {
    using E = Elem<DT>;
    using PM = PlaneMax<DT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* cur = (u32x4*)smem;
    u32x4* tmp = cur + (size_t)H * W * vpb;
    const int nvb = (C / E::VEC) / vpb;                 // vector groups per image
    const int b = blockIdx.x / nvb, vg = blockIdx.x - b * nvb;
    const int n = H * W * vpb, r = k / 2;
    const long long pix0 = (long long)b * H * W;
    // Item i = (pixel p, vector v) -> (v, x, y) is the same in all seven passes: taken apart ONCE per thread (the launcher caps the
    // plane at 60 KB = 1920 items, i.e. at most MAXIT items per thread) instead of four integer divisions per item and pass, which
    // cost more instructions than the pass's LDS traffic.
    constexpr int MAXIT = 8;
    short iv[MAXIT], ix[MAXIT], iy[MAXIT];
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int i = threadIdx.x + t * 256;
        const int ii = i < n ? i : 0;
        const int v = ii % vpb, pp = ii / vpb;
        iv[t] = (short)v; ix[t] = (short)(pp % W); iy[t] = (short)(pp / W);
    }
#pragma unroll
    for (int t = 0; t < MAXIT; ++t) {
        const int i = threadIdx.x + t * 256;
        if (i < n) cur[i] = PM::in(*(const u32x4*)(x + (pix0 + iy[t] * W + ix[t]) * ldx + (vg * vpb + iv[t]) * E::VEC));
    }
    __syncthreads();
    typename E::type* outs[3] = {y1, y2, y3};
#pragma unroll
    for (int stage = 0; stage < 3; ++stage) {
#pragma unroll
        for (int t = 0; t < MAXIT; ++t) {
            const int i = threadIdx.x + t * 256;
            if (i >= n) continue;
            const int xx = ix[t];
            u32x4 m = cur[i];
            for (int d = -r; d <= r; ++d) {
                if (d == 0 || (unsigned)(xx + d) >= (unsigned)W) continue;
                m = PM::mx(m, cur[i + d * vpb]);
            }
            tmp[i] = m;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < MAXIT; ++t) {
            const int i = threadIdx.x + t * 256;
            if (i >= n) continue;
            const int yy = iy[t];
            u32x4 m = tmp[i];
            for (int d = -r; d <= r; ++d) {
                if (d == 0 || (unsigned)(yy + d) >= (unsigned)H) continue;
                m = PM::mx(m, tmp[i + d * W * vpb]);
            }
            cur[i] = m;
            *(u32x4*)(outs[stage] + (pix0 + yy * W + ix[t]) * ldy + (vg * vpb + iv[t]) * E::VEC) = PM::out(m);
        }
        __syncthreads();
    }
}

// ---- nearest-neighbour integer upsample (writes a channel slice of the consumer's concat buffer) ---------------
struct UpsampleDiv { FastDiv nv, wo, ho, scale; };

template <bool I32>          // I32: flat vector index below 2^31 -> FastDiv (icaf_common.h) instead of 64-bit divisions
__global__ __launch_bounds__(256) void upsample_kernel(const u32x4* __restrict__ x, int ldxv, u32x4* __restrict__ y, int ldyv,
                                                       int B, int H, int W, int nv, int scale, UpsampleDiv dv) {
    const int Ho = H * scale, Wo = W * scale;
    const long long total = (long long)B * Ho * Wo * nv;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        if constexpr (I32) {
            unsigned int pix, v, t, wo, b, ho;
            fd_divmod((unsigned int)idx, dv.nv, pix, v);
            fd_divmod(pix, dv.wo, t, wo);
            fd_divmod(t, dv.ho, b, ho);
            const unsigned int hi = fd_div(ho, dv.scale), wi = fd_div(wo, dv.scale);
            y[(long long)pix * ldyv + v] = x[(((long long)b * H + hi) * W + wi) * ldxv + v];
        } else {
            const int v = (int)(idx % nv);
            const long long pix = idx / nv;
            const int wo = (int)(pix % Wo);
            const long long t = pix / Wo;
            const int ho = (int)(t % Ho), b = (int)(t / Ho);
            y[pix * ldyv + v] = x[(((long long)b * H + ho / scale) * W + wo / scale) * ldxv + v];
        }
    }
}

__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ x, int ldxv, u32x4* __restrict__ y, int ldyv,
                                                   long long rows, int nv) {
    const long long total = rows * nv;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(idx % nv);
        const long long r = idx / nv;
        y[r * ldyv + v] = x[r * ldxv + v];
    }
}

// ---- y = a * x0 + b * x1 over channel slices (Add fusion block, reference models/common.py:324-331) ----------------
template <int DT>
__global__ __launch_bounds__(256) void axpby_kernel(const typename Elem<DT>::type* __restrict__ x0, int ld0,
                                                    const typename Elem<DT>::type* __restrict__ x1, int ld1,
                                                    typename Elem<DT>::type* __restrict__ y, int ldy, long long rows, int nv, float a, float b) {
    using E = Elem<DT>;
    const long long total = rows * nv;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(idx % nv);
        const long long r = idx / nv;
        float f0[E::VEC], f1[E::VEC], o[E::VEC];
        unpack16<DT>(*(const u32x4*)(x0 + r * ld0 + v * E::VEC), f0);
        unpack16<DT>(*(const u32x4*)(x1 + r * ld1 + v * E::VEC), f1);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) o[j] = f0[j] * a + f1[j] * b;
        *(u32x4*)(y + r * ldy + v * E::VEC) = pack16<DT>(o);
    }
}

static inline unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;       // cap and grid-stride: 16 workgroups per CU keeps HBM queues full
    if (b < 1) b = 1;
    return (unsigned)b;
}

static inline int vec_of(int dtype) { return dtype == ICAF_F32 ? 4 : 8; }

}  // namespace icaf

using namespace icaf;

extern "C" int icaf_preprocess_nchw(const float* img, void* out, int dtype, int B, int C, int H, int W, int Cpad, int mode,
                                    icaf_stream_t s) {
    if (!img || !out) return fail(ICAF_ERR_ARG, "icaf_preprocess_nchw: null pointer");
    if (dtype < 0 || dtype > 2) return fail(ICAF_ERR_ARG, "icaf_preprocess_nchw: bad dtype");
    const int vec = vec_of(dtype);
    if (Cpad % vec) return fail(ICAF_ERR_ARG, "icaf_preprocess_nchw: Cpad %d must be a multiple of %d", Cpad, vec);
    if (mode == 1 && ((H | W) & 1)) return fail(ICAF_ERR_ARG, "icaf_preprocess_nchw: space-to-depth needs even H, W");
    if (mode == 1 ? Cpad < 4 * C : Cpad < C) return fail(ICAF_ERR_ARG, "icaf_preprocess_nchw: Cpad too small");
    const long long total = (long long)B * (mode ? H / 2 : H) * (mode ? W / 2 : W);
    dim3 grid(grid_for(total)), block(256);
    if (dtype == ICAF_F32) hipLaunchKernelGGL(preprocess_kernel<ICAF_F32>, grid, block, 0, S(s), img, (float*)out, B, C, H, W, Cpad, mode);
    else if (dtype == ICAF_BF16) hipLaunchKernelGGL(preprocess_kernel<ICAF_BF16>, grid, block, 0, S(s), img, (unsigned short*)out, B, C, H, W, Cpad, mode);
    else hipLaunchKernelGGL(preprocess_kernel<ICAF_F16>, grid, block, 0, S(s), img, (unsigned short*)out, B, C, H, W, Cpad, mode);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

extern "C" int icaf_preprocess_u8(const unsigned char* img, void* out, int dtype, int B, int Ctot, int c0, int C, int nstreams, int H,
                                  int W, int Cpad, int mode, icaf_stream_t s) {
    if (!img || !out) return fail(ICAF_ERR_ARG, "icaf_preprocess_u8: null pointer");
    if (dtype < 0 || dtype > 2) return fail(ICAF_ERR_ARG, "icaf_preprocess_u8: bad dtype");
    const int vec = vec_of(dtype);
    if (Cpad % vec) return fail(ICAF_ERR_ARG, "icaf_preprocess_u8: Cpad %d must be a multiple of %d", Cpad, vec);
    if (mode == 1 && ((H | W) & 1)) return fail(ICAF_ERR_ARG, "icaf_preprocess_u8: space-to-depth needs even H, W");
    if (mode == 1 ? Cpad < 4 * C : Cpad < C) return fail(ICAF_ERR_ARG, "icaf_preprocess_u8: Cpad too small");
    if (nstreams < 1 || c0 < 0 || c0 + nstreams * C > Ctot) return fail(ICAF_ERR_ARG, "icaf_preprocess_u8: channels [%d, %d) exceed %d", c0, c0 + nstreams * C, Ctot);
    const long long total = (long long)nstreams * B * (mode ? H / 2 : H) * (mode ? W / 2 : W);
    dim3 grid(grid_for(total)), block(256);
    if (dtype == ICAF_F32) hipLaunchKernelGGL(preprocess_u8_kernel<ICAF_F32>, grid, block, 0, S(s), img, (float*)out, B, Ctot, c0, C, nstreams, H, W, Cpad, mode);
    else if (dtype == ICAF_BF16) hipLaunchKernelGGL(preprocess_u8_kernel<ICAF_BF16>, grid, block, 0, S(s), img, (unsigned short*)out, B, Ctot, c0, C, nstreams, H, W, Cpad, mode);
    else hipLaunchKernelGGL(preprocess_u8_kernel<ICAF_F16>, grid, block, 0, S(s), img, (unsigned short*)out, B, Ctot, c0, C, nstreams, H, W, Cpad, mode);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}


extern "C" int icaf_sppf_pool(const void* x, int ldx, void* y1, void* y2, void* y3, int ldy, int dtype, int B, int H, int W, int C,
                              int k, icaf_stream_t s) {
    if (!x || !y1 || !y2 || !y3) return fail(ICAF_ERR_ARG, "icaf_sppf_pool: null pointer");
    if (dtype < 0 || dtype > 2) return fail(ICAF_ERR_ARG, "icaf_sppf_pool: bad dtype");
    const int vec = vec_of(dtype);
    if (C % vec || ldx % vec || ldy % vec || !(k & 1)) return fail(ICAF_ERR_ARG, "icaf_sppf_pool: C/ld must be multiples of %d and k odd", vec);
    const int nv = C / vec;
    // channel vectors per workgroup: the largest group whose plane fits.  A cap of 2 (twice the workgroups) is faster in isolation
    // (lab/probes/sppf_vpb.py, 64 x 20x20 x 256 bf16: 32.7 -> 29.3 us) and slower in the forward (35 -> 40 us, input not cache-resident)
    int vpb = 0, cap = 8;
    if (g_opt.sppf_vpb > 0) cap = g_opt.sppf_vpb;                                         // probe knob (icaf_set_option)
    for (int c : {8, 4, 2, 1})
        if (c <= cap && nv % c == 0 && (size_t)H * W * c * 32 <= 60 * 1024) { vpb = c; break; }
    if (vpb) {
        const size_t lds = (size_t)H * W * vpb * 32;
        dim3 grid((unsigned)(B * (nv / vpb))), block(256);
        if (dtype == ICAF_F32) sppf_lds_kernel<ICAF_F32><<<grid, block, lds, S(s)>>>((const float*)x, ldx, (float*)y1, (float*)y2, (float*)y3, ldy, H, W, C, k, vpb);
        else if (dtype == ICAF_BF16) sppf_lds_kernel<ICAF_BF16><<<grid, block, lds, S(s)>>>((const unsigned short*)x, ldx, (unsigned short*)y1, (unsigned short*)y2, (unsigned short*)y3, ldy, H, W, C, k, vpb);
        else sppf_lds_kernel<ICAF_F16><<<grid, block, lds, S(s)>>>((const unsigned short*)x, ldx, (unsigned short*)y1, (unsigned short*)y2, (unsigned short*)y3, ldy, H, W, C, k, vpb);
        ICAF_LAUNCH_CHECK();
        return ICAF_OK;
    }
    const long long total = (long long)B * H * W * (C / vec);
    dim3 grid(grid_for(total)), block(256);
    if (dtype == ICAF_F32) hipLaunchKernelGGL(sppf_kernel<ICAF_F32>, grid, block, 0, S(s), (const float*)x, ldx, (float*)y1, (float*)y2, (float*)y3, ldy, B, H, W, C, k);
    else if (dtype == ICAF_BF16) hipLaunchKernelGGL(sppf_kernel<ICAF_BF16>, grid, block, 0, S(s), (const unsigned short*)x, ldx, (unsigned short*)y1, (unsigned short*)y2, (unsigned short*)y3, ldy, B, H, W, C, k);
    else hipLaunchKernelGGL(sppf_kernel<ICAF_F16>, grid, block, 0, S(s), (const unsigned short*)x, ldx, (unsigned short*)y1, (unsigned short*)y2, (unsigned short*)y3, ldy, B, H, W, C, k);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

extern "C" int icaf_upsample_nearest(const void* x, int ldx, void* y, int ldy, int dtype, int B, int H, int W, int C, int scale,
                                     icaf_stream_t s) {
    if (!x || !y) return fail(ICAF_ERR_ARG, "icaf_upsample_nearest: null pointer");
    if (dtype < 0 || dtype > 2) return fail(ICAF_ERR_ARG, "icaf_upsample_nearest: bad dtype");
    const int vec = vec_of(dtype);
    if (C % vec || ldx % vec || ldy % vec || scale < 1) return fail(ICAF_ERR_ARG, "icaf_upsample_nearest: bad geometry");
    const long long total = (long long)B * H * scale * W * scale * (C / vec);
    if (total < 1) return fail(ICAF_ERR_ARG, "icaf_upsample_nearest: empty tensor");
    const UpsampleDiv dv{make_fastdiv((unsigned)(C / vec)), make_fastdiv((unsigned)(W * scale)), make_fastdiv((unsigned)(H * scale)),
                         make_fastdiv((unsigned)scale)};
    if (total < (1ll << 31))
        hipLaunchKernelGGL(upsample_kernel<true>, dim3(grid_for(total)), dim3(256), 0, S(s), (const u32x4*)x, ldx / vec, (u32x4*)y,
                           ldy / vec, B, H, W, C / vec, scale, dv);
    else
        hipLaunchKernelGGL(upsample_kernel<false>, dim3(grid_for(total)), dim3(256), 0, S(s), (const u32x4*)x, ldx / vec, (u32x4*)y,
                           ldy / vec, B, H, W, C / vec, scale, dv);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

extern "C" int icaf_axpby(const void* x0, int ld0, const void* x1, int ld1, void* y, int ldy, int dtype, long long rows, int C, float a,
                          float b, icaf_stream_t s) {
    if (!x0 || !x1 || !y) return fail(ICAF_ERR_ARG, "icaf_axpby: null pointer");
    if (dtype < 0 || dtype > 2) return fail(ICAF_ERR_ARG, "icaf_axpby: bad dtype");
    const int vec = vec_of(dtype);
    if (C % vec || ld0 % vec || ld1 % vec || ldy % vec) return fail(ICAF_ERR_ARG, "icaf_axpby: C/ld must be multiples of %d", vec);
    dim3 grid(grid_for(rows * (C / vec))), block(256);
    if (dtype == ICAF_F32) axpby_kernel<ICAF_F32><<<grid, block, 0, S(s)>>>((const float*)x0, ld0, (const float*)x1, ld1, (float*)y, ldy, rows, C / vec, a, b);
    else if (dtype == ICAF_BF16) axpby_kernel<ICAF_BF16><<<grid, block, 0, S(s)>>>((const unsigned short*)x0, ld0, (const unsigned short*)x1, ld1, (unsigned short*)y, ldy, rows, C / vec, a, b);
    else axpby_kernel<ICAF_F16><<<grid, block, 0, S(s)>>>((const unsigned short*)x0, ld0, (const unsigned short*)x1, ld1, (unsigned short*)y, ldy, rows, C / vec, a, b);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

extern "C" int icaf_copy_channels(const void* x, int ldx, void* y, int ldy, int dtype, long long rows, int C, icaf_stream_t s) {
    if (!x || !y) return fail(ICAF_ERR_ARG, "icaf_copy_channels: null pointer");
    if (dtype < 0 || dtype > 2) return fail(ICAF_ERR_ARG, "icaf_copy_channels: bad dtype");
    const int vec = vec_of(dtype);
    if (C % vec || ldx % vec || ldy % vec) return fail(ICAF_ERR_ARG, "icaf_copy_channels: C/ld must be multiples of %d", vec);
    hipLaunchKernelGGL(copy_kernel, dim3(grid_for(rows * (C / vec))), dim3(256), 0, S(s), (const u32x4*)x, ldx / vec, (u32x4*)y, ldy / vec,
                       rows, C / vec);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}
