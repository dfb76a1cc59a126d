// Crossed-attention inner loop shared by cross_attn_kernel (dmff.hip) and dmff_attn_mlp_kernel (dmff_fused.hip):
//   one wavefront = one 32-query tile of one head against all keys of that head, K rows and V^T rows resident in LDS
//   (reference models/common.py:670-681:  softmax(q_other k^T / sqrt(dk)) v).
//
// Round 4: the loop is bound by VALU issue, not by the matrix pipe — at d_k = 16 a 32 x 32 score tile is 3 MFMAs (96 pipe cycles)
// against 16 scores per lane of softmax arithmetic; the round-3 ISA spent ~118 issue slots per tile, of which 32 were
// v_accvgpr_read (scores AND the whole O accumulator, hoisted in front of the rescale branch), 16 the row-sum adds, and one an LDS
// round trip (ds_bpermute) in the middle of the dependency chain.  What changed:
//   * the file is compiled with MFMA results in VGPRs (-mllvm -amdgpu-mfma-vgpr-form, build.py): no accvgpr traffic at all;
//   * the softmax DENOMINATOR comes out of the matrix pipe: rows of O^T that the head dimension leaves unused (d_k = 16, 48: sixteen
//     of the 32 rows of the last O^T tile) read an all-ones V^T row, so O^T[16 + i][q] = sum_k P[q][k] — no adds, no final cross-half
//     shuffle, and the rescale of O rescales it too; at d_k = 32 (no free rows) one extra MFMA per P step with a constant all-ones
//     A operand does the same (the pipe is idle two thirds of the time there); d_k >= 64 keeps the VALU adds (matrix-bound already).
//     The sum is then over the ROUNDED probabilities — numerator and denominator see the same P;
//   * the cross-half maximum is ONE v_permlane32_swap instead of ds_bpermute + s_waitcnt;
//   * deferred maximum: the running maximum only moves (and O is only rescaled) when a tile exceeds it by more than 2^DEFER in the
//     exponent domain — p stays <= 2^DEFER, exact in bf16 / f16 / fp32 alike (softmax is shift-invariant; only roundings move);
//   * the score MFMAs of tile t + 1 are issued BEFORE the softmax arithmetic of tile t (independent accumulators): the matrix pipe
//     works under the wave's own VALU stream instead of in front of it.
// fp32 instantiation (parity build): same code, VALU row sums, no deferral.
#pragma once
#include "icaf_common.h"

namespace icaf {

// max(x, x of lane ^ 32): one VALU half-swap (v_permlane32_swap: lanes 32-63 of the first operand trade places with lanes 0-31 of
// the second), no LDS round trip
__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int DT> __device__ __forceinline__ u32x4 ones_frag() {
    if constexpr (DT == ICAF_F32) return u32x4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
    else if constexpr (DT == ICAF_BF16) return u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    else return u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
}

// P (this lane's 16 probabilities of one query, accumulator order) -> the B operand of P step `st`
template <int DT> __device__ __forceinline__ u32x4 attn_pack_p(const f32x16& s, int st) {
    u32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (DT == ICAF_F32) v[e] = __float_as_uint(s[4 * st + e]);
        else if constexpr (DT == ICAF_BF16) v[e] = pack2_bf16(s[8 * st + 2 * e], s[8 * st + 2 * e + 1]);
        else v[e] = pack2_f16(s[8 * st + 2 * e], s[8 * st + 2 * e + 1]);
    }
    return v;
}

template <int DT, int DKP> struct AttnCore {
    using E = Elem<DT>;
    static constexpr int VEC = E::VEC, EB = E::BYTES;
    static constexpr int KSTEP = 2 * VEC;              // reduction elements consumed per mma_step
    static constexpr int QSTEPS = DKP / KSTEP;         // steps over d for S^T = K Q^T
    static constexpr int TD = (DKP + 31) / 32;         // 32-row d tiles of O^T
    static constexpr int PSTEPS = 32 / KSTEP;          // steps over the 32 keys of a tile for O^T += V^T P^T
    // where the softmax denominator is accumulated
    static constexpr bool FREE = DT != ICAF_F32 && (DKP % 32) == 16;     // unused rows 16-31 of the last O^T tile read a ones row of V^T
    static constexpr bool XTRA = DT != ICAF_F32 && DKP == 32;            // one more accumulator fed by an all-ones A operand
    static constexpr int NO = TD + (XTRA ? 1 : 0);
    static constexpr int DEFER = DT == ICAF_F32 ? 0 : 6;                 // p <= 2^6 between rescales
    // bytes of the ones row a caller appends to a head's V^T block (one row of VS bytes), 0 when not needed
    static __host__ __device__ constexpr bool ones_row() { return FREE; }

    // Ks: K rows of the head, key-major, row stride KS;  Vt: V^T rows (d-major, keys permuted inside 16-key groups as vt_phys), row
    // stride VS;  ones: the all-ones row behind them (FREE only).  qf: this lane's query fragments.  nkt = NP / 32 key tiles, keys
    // >= N are padding.  c = log2(e) / sqrt(dk).  Returns the UNNORMALISED O^T tiles in o[0 .. TD) and the row sum l (complete: both
    // lane halves hold the same value).
    template <int KS>
    static __device__ __forceinline__ void run(const unsigned char* __restrict__ Ks, const unsigned char* __restrict__ Vt, int VS,
                                               const unsigned char* __restrict__ ones, const u32x4 (&qf)[QSTEPS], int nkt, int N, float c,
                                               f32x16 (&o)[TD], float& l_out) {
        const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
        f32x16 acc[NO];
#pragma unroll
        for (int t = 0; t < NO; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        float m = -INFINITY, l = 0.0f;
        const float defer = DEFER ? (float)DEFER / c : 0.0f;        // the exponent-domain slack in score units
        // per-lane V^T row pointers (row clamped as before; FREE: lanes 16-31 of the last tile -> the ones row)
        const unsigned char* vrow[TD];
#pragma unroll
        for (int td = 0; td < TD; ++td) {
            int drow = td * 32 + l31;
            drow = drow < DKP ? drow : DKP - 1;
            vrow[td] = Vt + (size_t)drow * VS + (size_t)hi * VEC * EB;
            if constexpr (FREE) { if (td == TD - 1 && l31 >= 16) vrow[td] = ones + (size_t)hi * VEC * EB; }
        }
        const unsigned char* krow = Ks + (size_t)l31 * KS + hi * 16;
        const u32x4 onesf = ones_frag<DT>();

        auto scores = [&](int kt) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
            for (int st = 0; st < QSTEPS; ++st) {
                const u32x4 kf = *(const u32x4*)(krow + (size_t)kt * 32 * KS + st * 32);
                mma_step<DT>(s, kf, qf[st]);
            }
            return s;
        };
        // softmax + P V^T of one key tile whose scores are in s (consumed)
        auto tile = [&](f32x16& s, int kt) {
            if (kt == nkt - 1 && (nkt << 5) != N) {                  // only the last key tile can hold padding keys
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    s[r] = key < N ? s[r] : -INFINITY;
                }
            }
            float tmax = fmaxf(s[0], s[1]);
#pragma unroll
            for (int r = 2; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
            tmax = xhalf_max(tmax);
            if (__any(tmax > m + defer)) {                           // rare after the first tiles: move the maximum, rescale O (and the sum)
                const float m_new = fmaxf(m, tmax);
                const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
                l *= alpha;
#pragma unroll
                for (int t = 0; t < NO; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] *= alpha;
                m = m_new;
            }
            const float mc = m * c;
            if constexpr (FREE || XTRA) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -mc));
            } else {
                float psum = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -mc));
                    s[r] = pv;
                    psum += pv;
                }
                l += psum;
            }
#pragma unroll
            for (int st = 0; st < PSTEPS; ++st) {
                const u32x4 pf = attn_pack_p<DT>(s, st);
#pragma unroll
                for (int td = 0; td < TD; ++td) {
                    const u32x4 vf = *(const u32x4*)(vrow[td] + (size_t)(kt * 32 + st * KSTEP) * EB);
                    mma_step<DT>(acc[td], vf, pf);
                }
                if constexpr (XTRA) mma_step<DT>(acc[TD], onesf, pf);
            }
        };
        // two tiles per trip, the score registers ping-pong: the NEXT tile's score MFMAs are issued before this tile's softmax
        // arithmetic (independent accumulators), so the matrix pipe works under the wave's own VALU stream instead of in front of it
        f32x16 sa = scores(0), sb;
        for (int kt = 0; kt < nkt; kt += 2) {
            const bool two = kt + 1 < nkt;
            if (two) sb = scores(kt + 1);
            tile(sa, kt);
            if (two) {
                if (kt + 2 < nkt) sa = scores(kt + 2);
                tile(sb, kt + 1);
            }
        }
        if constexpr (FREE) l_out = acc[TD - 1][8];                  // rows 16 .. 31 of the last tile: every one of them is the row sum
        else if constexpr (XTRA) l_out = acc[TD][0];
        else l_out = l + __shfl_xor(l, 32);
#pragma unroll
        for (int td = 0; td < TD; ++td) o[td] = acc[td];
    }
};

}  // namespace icaf
