// Persistent streaming implicit GEMM (stream_core.h): the plain layer epilogue — bias + activation, rounded to the storage type,
// optional residual (the Bottleneck shortcut / coefficient mixes), 16-byte NHWC stores — and the host side of tile ids 51 / 52.  (detect.hip instantiates the same core with the
// Detect head's decode as its epilogue.)
#include "stream_core.h"

namespace icaf {

template <int DT, int BN, int ACT>
struct StoreEpi {
    using E = Elem<DT>;
    static constexpr int SO = BN * E::BYTES + 16;
    typename E::type* yg; float alpha_acc; int M, Cout, ldy;
    const typename E::type* rg; float alpha_res; int ldr;           // residual (nullptr = none); may alias yg (in-place Bottleneck chain)
    template <int TM>
    __device__ __forceinline__ void stage(const f32x16 (&acc)[TM], const f32x4 (&bq)[4], unsigned char* stg, int row0, int col0, int l31, int hi) const {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int nl = col0 + 8 * qd + 4 * hi;                   // tile-local channel of this register quad
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int ml = row0 + b * 32 + l31;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act<ACT, DT>(acc[b][4 * qd + j] + bq[qd][j] + 0.0f) * alpha_acc;
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

template <int DT, int BN, int ACT, int MODE>
__global__ __launch_bounds__(512) void igemm_stream_kernel(const ConvP p) {
    using E = Elem<DT>;
    const int g = blockIdx.z;
    const StoreEpi<DT, BN, ACT> epi{(typename E::type*)p.y + g * p.y_gs, p.alpha_acc[g], p.M, p.Cout, p.ldy,
                                    p.res ? (const typename E::type*)p.res + g * p.res_gs : nullptr, p.alpha_res[g], p.ldr};
    stream_gemm<DT, BN, MODE>(p, epi);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
const char* stream_tag(int shape) { return shape == 1 ? "128x128" : shape == 2 ? "128x64" : "?"; }

int stream_check(const icaf_conv_args* a, const ConvP& p, int shape) {
    if (shape < 1 || shape > 2) return fail(ICAF_ERR_ARG, "igemm_stream: unknown shape %d", shape);
    if (a->dtype == ICAF_F32 || a->out_dtype != a->dtype) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: 16-bit types, out dtype == dtype");
    if ((a->Cin * 2) % 128) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: Cin * 2 bytes must be a multiple of 128 (Cin = %d)", a->Cin);
    if (a->pre || a->w2) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: no pre-activation term / chained layer");
    if (a->res && !p.vec_r) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: the residual must take 16-byte vectors");
    if (p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: operand exceeds the 2 GiB buffer-descriptor range");
    if (!p.vec_y || a->Cout % 8) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: y must take 16-byte vectors (ldy %% 8, Cout %% 8, alignment)");
    if (shape == 1 && a->Cout <= 64) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream 128x128: Cout = %d <= 64 (use 128x64)", a->Cout);
    return ICAF_OK;
}

template <int DT, int BN, int ACT>
static int launch_stream_cfg(const ConvP& p, int groups, hipStream_t s) {
    constexpr int LDS = 3 * (128 + BN) * 128 + 128 * (BN * 2 + 16);
    ConvP q = p;
    q.mtiles = (p.M + 127) / 128;
    q.ntiles = (p.Cout + BN - 1) / BN;
    q.nchunks = p.K / 64;                          // 128-byte slices of a 16-bit type (K = kh * kw * Cin is a multiple of 64)
    int dev = 0, cus = 256;
    ICAF_HIP(hipGetDevice(&dev));
    ICAF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int grid = (cus / groups) & ~7;                // one workgroup per CU (LDS), the groups (RGB / IR stream) side by side; 8 XCDs
    const int wgx = grid >> 3;
    if (wgx < q.ntiles || wgx % q.ntiles) return fail(ICAF_ERR_UNSUPPORTED, "igemm_stream: %d channel tiles do not divide the %d workgroups of an XCD", q.ntiles, wgx);
    // (fewer pixel tiles than pixel slots: the surplus workgroups find their range empty and exit)
    const bool plain = q.kh == 1 && q.kw == 1 && q.sh == 1 && q.sw == 1 && q.ph == 0 && q.pw == 0;
    auto go = [&](auto kern) -> int {
        static std::atomic<bool> attr{false};      // one flag per instantiation (one process drives one GPU)
        if (!attr) {
            ICAF_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
            attr = true;
        }
        kern<<<dim3((unsigned)grid, 1, (unsigned)groups), dim3(512), LDS, s>>>(q);
        ICAF_LAUNCH_CHECK();
        return ICAF_OK;
    };
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
