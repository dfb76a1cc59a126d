// Persistent streaming implicit GEMM (stream_core.h): the plain layer epilogue — bias + activation, rounded to the storage type,
// optional residual (the Bottleneck shortcut / coefficient mixes), 16-byte NHWC stores — and the host side of tile ids 51 / 52.  (detect.hip instantiates the same core with the
// Detect head's decode as its epilogue.)
#include "stream_core.h"

namespace icaf {

template <int DT, int BN, int ACT, bool PRE = false>
struct StoreEpi {
    using E = Elem<DT>;
    static constexpr int SO = BN * E::BYTES + 16;
    typename E::type* yg; float alpha_acc; int M, Cout, ldy;
    const typename E::type* rg; float alpha_res; int ldr;           // residual (nullptr = none); may alias yg (in-place Bottleneck chain)
    // PRE: the pre-activation term of DMFF's fused tail / the folded up-sampling (icaf.h: icaf_conv_args.pre) — a coarse fp32 map
    // resized (bilinear, align_corners = False; or nearest) to the output and added before the activation: the shared epilogue's
    // expressions, fp contraction OFF (conv_common.h explains why), so the result is bit-identical to igemm's.
    const float* pre; int pre_h, pre_w, ldpre, pre_mode, Ho, Wo;
    template <int TM>
    __device__ __forceinline__ void stage(const f32x16 (&acc)[TM], const f32x4 (&bq)[4], unsigned char* stg, int row0, int col0, int l31, int hi, int m0, int n0) const {
        const float* pt[PRE ? TM : 1][4];
        float plx[PRE ? TM : 1], ply[PRE ? TM : 1];
        if constexpr (PRE) {
#pragma clang fp contract(off)
            const float sy = (float)pre_h / (float)Ho, sx = (float)pre_w / (float)Wo;
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int m = m0 + row0 + b * 32 + l31;
                const int mm = m < M ? m : 0;
                const int wo = mm % Wo, t = mm / Wo, ho = t % Ho, bi = t / Ho;
                float fy = ((float)ho + 0.5f) * sy - 0.5f, fx = ((float)wo + 0.5f) * sx - 0.5f;
                fy = fy < 0.0f ? 0.0f : fy;
                fx = fx < 0.0f ? 0.0f : fx;
                int y0 = (int)fy, x0 = (int)fx;
                y0 = y0 < pre_h - 1 ? y0 : pre_h - 1;
                x0 = x0 < pre_w - 1 ? x0 : pre_w - 1;
                int y1 = y0 < pre_h - 1 ? y0 + 1 : y0, x1 = x0 < pre_w - 1 ? x0 + 1 : x0;
                ply[b] = fy - (float)y0;
                plx[b] = fx - (float)x0;
                if (pre_mode == 1) {               // nearest: one tap with weight 1 — the sequence below returns it exactly
                    y0 = y1 = (int)((long long)ho * pre_h / Ho);
                    x0 = x1 = (int)((long long)wo * pre_w / Wo);
                    ply[b] = plx[b] = 0.0f;
                }
                const float* base = pre + (long long)bi * pre_h * pre_w * ldpre;
                pt[b][0] = base + (long long)(y0 * pre_w + x0) * ldpre;
                pt[b][1] = base + (long long)(y0 * pre_w + x1) * ldpre;
                pt[b][2] = base + (long long)(y1 * pre_w + x0) * ldpre;
                pt[b][3] = base + (long long)(y1 * pre_w + x1) * ldpre;
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int nl = col0 + 8 * qd + 4 * hi;                   // tile-local channel of this register quad
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int ml = row0 + b * 32 + l31;
                float pv[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (PRE) if (n0 + nl < Cout) {
                    const f32x4 t00 = *(const f32x4*)(pt[b][0] + n0 + nl), t01 = *(const f32x4*)(pt[b][1] + n0 + nl);
                    const f32x4 t10 = *(const f32x4*)(pt[b][2] + n0 + nl), t11 = *(const f32x4*)(pt[b][3] + n0 + nl);
                    {
#pragma clang fp contract(off)
                        const float wx0 = 1.0f - plx[b], wy0 = 1.0f - ply[b];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float top = t00[j] * wx0 + t01[j] * plx[b];
                            const float bot = t10[j] * wx0 + t11[j] * plx[b];
                            pv[j] = top * wy0 + bot * ply[b];
                        }
                    }
                }
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[b][4 * qd + j] + bq[qd][j] + pv[j];
                apply_act4<ACT, DT>(v, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= alpha_acc;
                u32x2 pk;
                if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                *(u32x2*)(stg + ml * SO + nl * E::BYTES) = pk;
            }
        }
    }
    __device__ __forceinline__ void flush(const unsigned char* stg, int m0, int n0, int tid) const {
        constexpr int VPR = BN / E::VEC;           // 16-byte vectors per staged row
        constexpr int NIT = 128 * VPR / 512;
#pragma unroll
        for (int itv = 0; itv < NIT; ++itv) {
            const int idx = tid + itv * 512;
            const int row = idx / VPR, cv = idx - row * VPR;
            const int m = m0 + row, n = n0 + cv * E::VEC;
            u32x4 sv = *(const u32x4*)(stg + row * SO + cv * 16);
            if (m < M && n < Cout) {
                if (rg) {                              // the shared epilogue's arithmetic (conv_common.h): staged value + alpha_res * residual
                    float v[E::VEC], r[E::VEC];
                    unpack16<DT>(sv, v);
                    unpack16<DT>(*(const u32x4*)(rg + (long long)m * ldr + n), r);
#pragma unroll
                    for (int j = 0; j < E::VEC; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                    sv = pack16<DT>(v);
                }
                *(u32x4*)(yg + (long long)m * ldy + n) = sv;
            }
        }
    }
};

template <int DT, int BN, int ACT, int MODE, bool PRE = false>
__global__ __launch_bounds__(512) void igemm_stream_kernel(const ConvP p) {
    using E = Elem<DT>;
    const int g = blockIdx.z;
    const StoreEpi<DT, BN, ACT, PRE> epi{(typename E::type*)p.y + g * p.y_gs, p.alpha_acc[g], p.M, p.Cout, p.ldy,
                                         p.res ? (const typename E::type*)p.res + g * p.res_gs : nullptr, p.alpha_res[g], p.ldr,
                                         p.pre, p.pre_h, p.pre_w, p.ldpre, p.pre_mode, p.Ho, p.Wo};
    stream_gemm<DT, BN, MODE>(p, epi);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
const char* stream_tag(int shape) { return shape == 1 ? "128x128" : shape == 2 ? "128x64" : "?"; }

// workgroups of the persistent grid per XCD for `groups` side-by-side groups on the current device (grid = 8 * wgx)
static int stream_grid(int groups, int* wgx) {
    int dev = 0, cus = 256;
    ICAF_HIP(hipGetDevice(&dev));
    ICAF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    *wgx = ((cus / (groups > 0 ? groups : 1)) & ~7) >> 3;
    return ICAF_OK;
}

int stream_check(const icaf_conv_args* a, const ConvP& p, int shape) {
    if (shape < 1 || shape > 2) return fail(ICAF_ERR_ARG, "igemm_stream: unknown shape %d", shape);
    if (a->dtype == ICAF_F32 || a->out_dtype != a->dtype) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: 16-bit types, out dtype == dtype");
    if ((a->Cin * 2) % 128) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: Cin * 2 bytes must be a multiple of 128 (Cin = %d)", a->Cin);
    if (a->w2) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: no chained layer");
    if (a->pre && (a->act != ICAF_ACT_SILU || a->kh != 1 || a->kw != 1 || a->sh != 1 || a->sw != 1 || a->ph || a->pw || a->groups != 1))
        return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: the pre-activation term is built for 1x1 SiLU layers (one group)");
    if (a->res && !p.vec_r) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: the residual must take 16-byte vectors");
    if (p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: operand exceeds the 2 GiB buffer-descriptor range");
    if (!p.vec_y || a->Cout % 8) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: y must take 16-byte vectors (ldy %% 8, Cout %% 8, alignment)");
    if (shape == 1 && a->Cout <= 64) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream 128x128: Cout = %d <= 64 (use 128x64)", a->Cout);
    // the persistent grid is one workgroup per CU; an XCD's workgroups are dealt over the channel tiles, so the tile count must divide
    // them (checked HERE, not only at launch: icaf_conv2d_kernel_name and the tuner's cache validation see the same verdict)
    int wgx = 0;
    const int st = stream_grid(a->groups, &wgx);
    if (st) return st;
    const int ntiles = (p.Cout + (shape == 1 ? 128 : 64) - 1) / (shape == 1 ? 128 : 64);
    if (wgx < ntiles || wgx % ntiles) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: %d channel tiles do not divide the %d workgroups of an XCD", ntiles, wgx);
    return ICAF_OK;
}

template <int DT, int BN, int ACT>
static int launch_stream_cfg(const ConvP& p, int groups, hipStream_t s) {
    constexpr int LDS = 3 * (128 + BN) * 128 + 128 * (BN * 2 + 16);
    ConvP q = p;
    q.mtiles = (p.M + 127) / 128;
    q.ntiles = (p.Cout + BN - 1) / BN;
    q.nchunks = p.K / 64;                          // 128-byte slices of a 16-bit type (K = kh * kw * Cin is a multiple of 64)
    int dev = 0, wgx = 0;
    ICAF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= ICAF_MAX_DEVICES) return fail(ICAF_ERR_UNSUPPORTED, "device ordinal %d", dev);
    int st = stream_grid(groups, &wgx);            // one workgroup per CU (LDS), the groups (RGB / IR stream) side by side; 8 XCDs
    if (st) return st;
    const int grid = wgx << 3;
    if (wgx < q.ntiles || wgx % q.ntiles) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: %d channel tiles do not divide the %d workgroups of an XCD", q.ntiles, wgx);
    // (fewer pixel tiles than pixel slots: the surplus workgroups find their range empty and exit)
    const bool plain = q.kh == 1 && q.kw == 1 && q.sh == 1 && q.sw == 1 && q.ph == 0 && q.pw == 0;
    auto go = [&](auto kern) -> int {
        ICAF_LDS_OPTIN(kern, LDS);               // per instantiation AND per device
        kern<<<dim3((unsigned)grid, 1, (unsigned)groups), dim3(512), LDS, s>>>(q);
        ICAF_LAUNCH_CHECK();
        return ICAF_OK;
    };
    if constexpr (ACT == ICAF_ACT_SILU) { if (p.pre) return go(igemm_stream_kernel<DT, BN, ACT, 1, true>); }
    if (plain) return go(igemm_stream_kernel<DT, BN, ACT, 1>);
    return go(igemm_stream_kernel<DT, BN, ACT, 2>);
}

template <int DT, int BN>
static int launch_stream_act(const ConvP& p, int groups, hipStream_t s) {
    if (p.act == ICAF_ACT_SILU) return launch_stream_cfg<DT, BN, ICAF_ACT_SILU>(p, groups, s);
    if (p.act == ICAF_ACT_GELU) return launch_stream_cfg<DT, BN, ICAF_ACT_GELU>(p, groups, s);
    return launch_stream_cfg<DT, BN, ICAF_ACT_NONE>(p, groups, s);
}

int launch_stream(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    int st = stream_check(a, p, shape);
    if (st) return st;
    if (a->dtype == ICAF_BF16) return shape == 1 ? launch_stream_act<ICAF_BF16, 128>(p, a->groups, s) : launch_stream_act<ICAF_BF16, 64>(p, a->groups, s);
    return shape == 1 ? launch_stream_act<ICAF_F16, 128>(p, a->groups, s) : launch_stream_act<ICAF_F16, 64>(p, a->groups, s);
}

}  // namespace icaf
