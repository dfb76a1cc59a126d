// Shared pieces of the convolution kernels (igemm.hip: implicit GEMM; ctile.hip: 3x3 direct convolution from an LDS
// halo tile): launch parameter block, epilogue (bias + activation + residual, LDS-staged 16-byte write-back), helpers.
#pragma once
#include "icaf_common.h"

namespace icaf {

struct ConvP {
    const void* x; const void* w; const float* bias; void* y; const void* res;
    long long x_gs, w_gs, bias_gs, y_gs, res_gs;
    int B, H, W, Cin, ldx, Ho, Wo, Cout, ldy, kh, kw, sh, sw, ph, pw, ldr, Kp, act;
    int M, K, nchunks, mtiles, ntiles;
    int vec_y, vec_r;
    unsigned int x_bytes, w_bytes;      // buffer-descriptor ranges (DMA pipeline); 0 = not representable
    float alpha_acc[2], alpha_res[2];
    const float* pre; int pre_h, pre_w, ldpre, pre_mode;   // optional pre-activation term, bilinear / nearest (icaf.h)
    const void* w1; const float* bias1;            // fused Bottleneck (ctile.hip, FUSE1): the 1x1 convolution in front
    long long w1_gs, bias1_gs; int Kp1; unsigned int w1_bytes;
    const void* w2; const float* bias2; void* y2;  // chained 1x1 convolution behind this layer (igemm.hip, CHAIN)
    long long w2_gs, bias2_gs, y2_gs; int Kp2, Cout2, ldy2, vec_y2, keep1; unsigned int w2_bytes;
    const void* x2; long long x2_gs; int ldx2;     // fused Bottleneck + cv3 (ctile.hip, CHAIN3): the cv2 half of cv3's input
};

constexpr int ROWB = 64;        // bytes of K per LDS row per slice
constexpr int ROWS = 80;        // padded LDS row stride in bytes (register-staged pipeline)
constexpr int NTHREADS = 256;

// DT = the layer's storage type: the 16-bit builds evaluate GELU's erf by a 1.5e-7-accurate polynomial (gelu_fast_f), fp32 by erff
template <int ACT, int DT = ICAF_F32> __device__ __forceinline__ float apply_act(float v) {
    if constexpr (ACT == ICAF_ACT_SILU) return silu_f(v);
    else if constexpr (ACT == ICAF_ACT_GELU) return DT == ICAF_F32 ? gelu_f(v) : gelu_fast_f(v);
    else return v;
}
// four values at once: SiLU goes through silu4_f (packed middle steps, same bits), everything else value by value
template <int ACT, int DT = ICAF_F32> __device__ __forceinline__ void apply_act4(const float (&x)[4], float (&y)[4]) {
    if constexpr (ACT == ICAF_ACT_SILU) silu4_f(x, y);
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = apply_act<ACT, DT>(x[j]);
    }
}

__device__ __forceinline__ int xcd_tile(int ntile_total) {
    // bijective remap: consecutive logical tiles stay on one XCD (blocks are dispatched round-robin over 8 XCDs)
    const int bid = blockIdx.x, q = ntile_total >> 3, r = ntile_total & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

#ifndef ICAF_EPI_FAST
#define ICAF_EPI_FAST 1          // 1: the restructured write-back (round 5: adopted after the whole GPU suite ran green on the variant library and a same-box
                                 //    A/B: default workload 16,141 -> 16,191 pairs/s, forward 2.191 -> 2.169 ms; yolov5l shard 3,726 -> 3,803, 9.34 -> 9.10 ms); 0: round 4's loop
#endif
#ifndef ICAF_PRE_WB
#define ICAF_PRE_WB 1            // 1: the pre-activation term is added in the write-back phase from an fp32-staged tile (0: round 4's per-lane taps; A/B builds)
#endif
// ---- epilogue shared by both pipelines: bias + activation in registers, LDS staging, 16-byte write-back --------
// row_to_m(tile_row) -> linear output pixel index (b, ho, wo), or -1 when the tile row lies outside the tensor.
// PRE = true compiles the pre-activation bilinear term in (it costs ~40 registers, so only the few instantiations that
// serve DMFF's fused tail carry it).
// SECOND = true writes the tile as the output of the chained second layer (p.bias2 / y2 / ldy2 / Cout2, no residual):
// selecting the fields here keeps ConvP in scalar registers — a modified copy of the struct would live in scratch memory.
// WB = true writes every final output vector (residual included) back into the staged LDS tile as well: igemm's chained
// 1x1 then consumes, as its pixel operand, exactly what this layer stores.
// FULLVEC = true: the caller has checked that whole BN-channel tiles are written (Cout % BN == 0) in 16-byte vectors without a residual — the
// write-back is then one vector load and one store per thread and round; the general loop below compiles four fall-back paths in (a short last vector, a
// scalar residual, unaligned rows), ~30 instructions of branching per round even when none is taken (icaf_bottleneck + cv3: eight rounds per workgroup).
template <int DT, int ODT, int BM, int BN, int WM, int WN, int ACT, bool PRE, bool SECOND = false, bool WB = false, bool FULLVEC = false, typename RowMap>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[WN / 32][WM / 32], unsigned char* lds, const ConvP& p, int g, RowMap row_to_m, int n0) {
    using E = Elem<DT>;
    using EO = Elem<ODT>;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_M = BM / WM;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;   // threads of the workgroup
    constexpr int VO = 16 / EO::BYTES;       // output elements per 16-byte vector
    constexpr int SO = BN * EO::BYTES + 16;  // staging row stride (bytes)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const float alpha_acc = SECOND ? 1.0f : p.alpha_acc[g], alpha_res = p.alpha_res[g];
    const float* __restrict__ bias = SECOND ? (p.bias2 ? p.bias2 + g * p.bias2_gs : nullptr) : (p.bias ? p.bias + g * p.bias_gs : nullptr);
    const int Cout = SECOND ? p.Cout2 : p.Cout, ldy = SECOND ? p.ldy2 : p.ldy, vec_y = SECOND ? p.vec_y2 : p.vec_y;
    if constexpr (PRE && ICAF_PRE_WB) {
        // Round 5 — the pre-activation term is added in the WRITE-BACK phase.  With the lane = pixel mapping of the accumulators every tap load was 64
        // lanes x 16 bytes in 64 different rows of the fp32 map (a row is Cout * 4 = 512 - 2048 bytes): uncoalesced, 16 * TN * TM such loads per lane —
        // the three fuse convolutions of yolov5s took 154 us where plain 1x1 layers of their shapes take ~60.  Here the tile is staged as FP32
        // (acc + bias, exactly the sum the old path formed first), and the thread that writes a 16-byte output vector adds the term: the VPR threads of
        // a pixel read 4 taps x (BN * 4) CONTIGUOUS bytes.  Same expressions in the same order ((acc + bias) + term, fp contraction off in the
        // bilinear arithmetic) => the same bits as before (tests: every configuration of a pre-term launch against the others and against torch).
        constexpr int SOF = BN * 4 + 16;                   // fp32 staging row stride (bytes)
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * WN + a * 32 + 8 * q + 4 * hi;
                const bool okn = bias && n < Cout;
                const float* bp = okn ? bias + n : (const float*)p.w;
                const f32x4 t = *(const f32x4*)bp;
                bq[q] = okn ? t : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * WN + a * 32 + 8 * q + 4 * hi;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const int ml = wm * WM + b * 32 + l31;
                    f32x4 t;
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = acc[a][b][4 * q + j] + bq[q][j];
                    *(f32x4*)(lds + ml * SOF + nl * 4) = t;
                }
            }
        }
        __syncthreads();
        typename EO::type* __restrict__ yg2 = SECOND ? (typename EO::type*)p.y2 + g * p.y2_gs : (typename EO::type*)p.y + g * p.y_gs;
        const typename E::type* __restrict__ rg2 = (!SECOND && p.res) ? (const typename E::type*)p.res + g * p.res_gs : nullptr;
        constexpr int VPR2 = BN / VO, NVEC2 = BM * VPR2, NIT2 = (NVEC2 + NT - 1) / NT;
        const float sy = (float)p.pre_h / (float)p.Ho, sx = (float)p.pre_w / (float)p.Wo;
#pragma unroll 2
        for (int it = 0; it < NIT2; ++it) {
            const int idx = tid + it * NT;
            if (idx >= NVEC2) break;
            const int row = idx / VPR2, cv = idx - row * VPR2;
            const int m = row_to_m(row), n = n0 + cv * VO;
            if (m < 0 || n >= Cout) continue;
            const float* pt4[4];
            float lx, ly;
            {
#pragma clang fp contract(off)
                const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, bi = t / p.Ho;
                float fy = ((float)ho + 0.5f) * sy - 0.5f, fx = ((float)wo + 0.5f) * sx - 0.5f;
                fy = fy < 0.0f ? 0.0f : fy;
                fx = fx < 0.0f ? 0.0f : fx;
                int y0 = (int)fy, x0 = (int)fx;
                y0 = y0 < p.pre_h - 1 ? y0 : p.pre_h - 1;
                x0 = x0 < p.pre_w - 1 ? x0 : p.pre_w - 1;
                int y1 = y0 < p.pre_h - 1 ? y0 + 1 : y0, x1 = x0 < p.pre_w - 1 ? x0 + 1 : x0;
                ly = fy - (float)y0;
                lx = fx - (float)x0;
                if (p.pre_mode == 1) {             // nearest: one tap with weight 1 — the sequence below returns it exactly
                    y0 = y1 = (int)((long long)ho * p.pre_h / p.Ho);
                    x0 = x1 = (int)((long long)wo * p.pre_w / p.Wo);
                    ly = lx = 0.0f;
                }
                const float* base = p.pre + (long long)bi * p.pre_h * p.pre_w * p.ldpre + n;
                pt4[0] = base + (long long)(y0 * p.pre_w + x0) * p.ldpre;
                pt4[1] = base + (long long)(y0 * p.pre_w + x1) * p.ldpre;
                pt4[2] = base + (long long)(y1 * p.pre_w + x0) * p.ldpre;
                pt4[3] = base + (long long)(y1 * p.pre_w + x1) * p.ldpre;
            }
            const int nvalid = (Cout - n) < VO ? (Cout - n) : VO;
            float v[VO];
#pragma unroll
            for (int k4 = 0; k4 < VO / 4; ++k4) {
                const f32x4 t = *(const f32x4*)(lds + row * SOF + (cv * VO + 4 * k4) * 4);
                float pv[4] = {0.f, 0.f, 0.f, 0.f};
                if (n + 4 * k4 < Cout) {           // (the old path's guard, per accumulator quad: ldpre >= Cout rounded up to 4)
                    const f32x4 t00 = *(const f32x4*)(pt4[0] + 4 * k4), t01 = *(const f32x4*)(pt4[1] + 4 * k4);
                    const f32x4 t10 = *(const f32x4*)(pt4[2] + 4 * k4), t11 = *(const f32x4*)(pt4[3] + 4 * k4);
                    {
#pragma clang fp contract(off)
                        const float wx0 = 1.0f - lx, wy0 = 1.0f - ly;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float top = t00[j] * wx0 + t01[j] * lx;
                            const float bot = t10[j] * wx0 + t11[j] * lx;
                            pv[j] = top * wy0 + bot * ly;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * k4 + j] = apply_act<ACT, DT>(t[j] + pv[j]) * alpha_acc;     // (write-back phase of the pre-term layers: not a hot SiLU site)
            }
            u32x4 o = pack16<ODT>(v);                      // rounded to the storage type, as the staged tile of the other path is
            typename EO::type* yp = yg2 + (long long)m * ldy + n;
            if (rg2) {
                unpack16<ODT>(o, v);
                const typename E::type* rp = rg2 + (long long)m * p.ldr + n;
                for (int j = 0; j < nvalid; ++j) v[j] = __builtin_fmaf(alpha_res, E::ld(rp + j), v[j]);
                o = pack16<ODT>(v);
            }
            if (vec_y && nvalid == VO) *(u32x4*)yp = o;
            else {
                unpack16<ODT>(o, v);
                for (int j = 0; j < nvalid; ++j) EO::st(yp + j, v[j]);
            }
        }
        return;
    }
    // pre-activation bilinear term: the four source taps and weights of this lane's TM pixels (align_corners=False:
    // src = max(0, (dst + 0.5) * in/out - 0.5), neighbours clamped), exactly as upsample_merge_kernel computes them
    const float* pt[PRE ? TM : 1][4];
    float plx[PRE ? TM : 1], ply[PRE ? TM : 1];
    if constexpr (PRE) {
        // No fp contraction in the bilinear arithmetic (here and where the term is evaluated below): left to the compiler,
        // different instantiations (tiles) fused these products differently, so the same layer rounded differently from one
        // tile shape to the next and a batch shard stopped being bit-identical to the same rows of the full batch.  (Writing
        // the fmas out with __builtin_fmaf instead produced a 128x64 / 64-byte-pipeline kernel with wrong, run-to-run
        // varying results at large grids — lab/probes/pre_check.py.)
#pragma clang fp contract(off)
        const float sy = (float)p.pre_h / (float)p.Ho, sx = (float)p.pre_w / (float)p.Wo;
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int m = row_to_m(wm * WM + b * 32 + l31);
            const int mm = m < 0 ? 0 : m;
            const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, bi = t / p.Ho;
            float fy = ((float)ho + 0.5f) * sy - 0.5f, fx = ((float)wo + 0.5f) * sx - 0.5f;
            fy = fy < 0.0f ? 0.0f : fy;
            fx = fx < 0.0f ? 0.0f : fx;
            int y0 = (int)fy, x0 = (int)fx;
            y0 = y0 < p.pre_h - 1 ? y0 : p.pre_h - 1;
            x0 = x0 < p.pre_w - 1 ? x0 : p.pre_w - 1;
            int y1 = y0 < p.pre_h - 1 ? y0 + 1 : y0, x1 = x0 < p.pre_w - 1 ? x0 + 1 : x0;
            ply[b] = fy - (float)y0;
            plx[b] = fx - (float)x0;
            if (p.pre_mode == 1) {                 // nearest: one tap with weight 1 — the fma sequence below returns it exactly
                y0 = y1 = (int)((long long)ho * p.pre_h / p.Ho);
                x0 = x1 = (int)((long long)wo * p.pre_w / p.Wo);
                ply[b] = plx[b] = 0.0f;
            }
            const float* base = p.pre + (long long)bi * p.pre_h * p.pre_w * p.ldpre;
            pt[b][0] = base + (long long)(y0 * p.pre_w + x0) * p.ldpre;
            pt[b][1] = base + (long long)(y0 * p.pre_w + x1) * p.ldpre;
            pt[b][2] = base + (long long)(y1 * p.pre_w + x0) * p.ldpre;
            pt[b][3] = base + (long long)(y1 * p.pre_w + x1) * p.ldpre;
        }
    }
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        // The four bias quads of this 32-channel block are fetched as ONE batch, unconditionally (clamped address +
        // select): a load inside the q loop is a load -> wait -> use chain, i.e. four dependent L2 round trips per block
        // at the end of every workgroup.  (All TN blocks at once would hold 16 * TN more registers next to acc.)
        f32x4 bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * WN + a * 32 + 8 * q + 4 * hi;
            const bool okn = bias && n < Cout;         // (the packed bias is padded to a multiple of 128 >= Cout only)
            const float* bp = okn ? bias + n : (const float*)p.w;      // (any mapped address: the value is discarded)
            const f32x4 t = *(const f32x4*)bp;
            bq[q] = okn ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl = wn * WN + a * 32 + 8 * q + 4 * hi;      // tile-local channel of this register quad
            const float bv[4] = {bq[q][0], bq[q][1], bq[q][2], bq[q][3]};
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                const int ml = wm * WM + b * 32 + l31;
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
                float v[4], xin[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xin[j] = acc[a][b][4 * q + j] + bv[j] + pv[j];
                apply_act4<ACT, DT>(xin, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= alpha_acc;
                unsigned char* dst = lds + ml * SO + nl * EO::BYTES;
                if constexpr (EO::BYTES == 4) {
                    *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    u32x2 pk;
                    if constexpr (ODT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                    else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                    *(u32x2*)dst = pk;
                }
            }
        }
    }
    __syncthreads();

    typename EO::type* __restrict__ yg = SECOND ? (typename EO::type*)p.y2 + g * p.y2_gs : (typename EO::type*)p.y + g * p.y_gs;
    const typename E::type* __restrict__ rg = (!SECOND && p.res) ? (const typename E::type*)p.res + g * p.res_gs : nullptr;
    constexpr int VPR = BN / VO;                       // 16-byte vectors per staged row
    constexpr int NVEC = BM * VPR;
    constexpr int NIT = (NVEC + NT - 1) / NT;
    if constexpr (FULLVEC) {
        static_assert(!WB && NVEC % NT == 0 && (VPR & (VPR - 1)) == 0, "whole rounds of power-of-two rows");
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, row = idx / VPR, cv = idx & (VPR - 1);
            const int m = row_to_m(row);
            if (m >= 0) *(u32x4*)(yg + (long long)m * ldy + n0 + cv * VO) = *(const u32x4*)(lds + row * SO + cv * 16);
        }
        return;
    }
#if ICAF_EPI_FAST
    // The write-back as two checked fast loops — whole tiles in 16-byte vectors, without / with a vector residual — and ONE compact general loop (not unrolled, nothing held
    // across iterations) for everything else, instead of the unrolled general loop below whose four fall-back paths cost ~30 instructions of branching
    // per 16-byte store.  Same arithmetic in every path (staged vector + alpha_res * residual, fma per element).
    if constexpr (VO == E::VEC && NVEC % NT == 0 && (VPR & (VPR - 1)) == 0) {
        if (vec_y && n0 + BN <= Cout && (!rg || p.vec_r)) {            // (workgroup-uniform)
            if (!rg) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int idx = tid + it * NT, row = idx / VPR, cv = idx & (VPR - 1);
                    const int m = row_to_m(row);
                    if (m >= 0) *(u32x4*)(yg + (long long)m * ldy + n0 + cv * VO) = *(const u32x4*)(lds + row * SO + cv * 16);
                }
            } else {
                u32x4 rv[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {                     // all residual vectors in flight first (clamped rows: never used)
                    const int idx = tid + it * NT, row = idx / VPR, cv = idx & (VPR - 1);
                    const int m = row_to_m(row);
                    rv[it] = *(const u32x4*)(rg + (long long)(m < 0 ? 0 : m) * p.ldr + n0 + cv * VO);
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int idx = tid + it * NT, row = idx / VPR, cv = idx & (VPR - 1);
                    const int m = row_to_m(row);
                    if (m < 0) continue;
                    float v[VO], r[VO];
                    unpack16<ODT>(*(const u32x4*)(lds + row * SO + cv * 16), v);
                    unpack16<DT>(rv[it], r);
#pragma unroll
                    for (int j = 0; j < VO; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                    const u32x4 o = pack16<ODT>(v);
                    if constexpr (WB) *(u32x4*)(lds + row * SO + cv * 16) = o;
                    *(u32x4*)(yg + (long long)m * ldy + n0 + cv * VO) = o;
                }
            }
            return;
        }
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT, row = idx / VPR, cv = idx & (VPR - 1);
            const int m = row_to_m(row), n = n0 + cv * VO;
            if (m < 0 || n >= Cout) continue;
            const u32x4 sv = *(const u32x4*)(lds + row * SO + cv * 16);
            const int nvalid = (Cout - n) < VO ? (Cout - n) : VO;
            typename EO::type* yp = yg + (long long)m * ldy + n;
            if (!rg && vec_y && nvalid == VO) { *(u32x4*)yp = sv; continue; }
            float v[VO];
            unpack16<ODT>(sv, v);
            if (rg) {
                const typename E::type* rp = rg + (long long)m * p.ldr + n;
                if (p.vec_r && nvalid == VO) {
                    float r[VO];
                    unpack16<DT>(*(const u32x4*)rp, r);
#pragma unroll
                    for (int j = 0; j < VO; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                } else {
                    for (int j = 0; j < nvalid; ++j) v[j] = __builtin_fmaf(alpha_res, E::ld(rp + j), v[j]);
                }
            }
            if constexpr (WB) *(u32x4*)(lds + row * SO + cv * 16) = pack16<ODT>(v);
            if (vec_y && nvalid == VO) *(u32x4*)yp = pack16<ODT>(v);
            else for (int j = 0; j < nvalid; ++j) EO::st(yp + j, v[j]);
        }
        return;
    }
#endif
    // residual vectors of all of this thread's output vectors first (one batch of loads in flight instead of a
    // load -> wait -> add -> store chain per vector)
    u32x4 rvec[NIT];
    bool rfast[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        rfast[it] = false;
        if constexpr (VO == E::VEC) {
            const int idx = tid + it * NT;
            const int row = idx / VPR, cv = idx - row * VPR;
            const int m = idx < NVEC ? row_to_m(row) : -1, n = n0 + cv * VO;
            rfast[it] = rg && p.vec_r && m >= 0 && n + VO <= Cout;
            if (rfast[it]) rvec[it] = *(const u32x4*)(rg + (long long)m * p.ldr + n);
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NT;
        if (idx >= NVEC) break;
        const int row = idx / VPR, cv = idx - row * VPR;
        const int m = row_to_m(row), n = n0 + cv * VO;
        if (m < 0 || n >= Cout) continue;
        const u32x4 sv = *(const u32x4*)(lds + row * SO + cv * 16);
        const int nvalid = (Cout - n) < VO ? (Cout - n) : VO;
        if (!rg && vec_y && nvalid == VO) {        // common case: no residual — the staged vector is final
            *(u32x4*)(yg + (long long)m * ldy + n) = sv;
            continue;
        }
        float v[VO];
        unpack16<ODT>(sv, v);
        if (rg) {
            const typename E::type* rp = rg + (long long)m * p.ldr + n;
            if (rfast[it]) {
                if constexpr (VO == E::VEC) {
                    float r[VO];
                    unpack16<DT>(rvec[it], r);
#pragma unroll
                    for (int j = 0; j < VO; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                }
            } else if (p.vec_r && nvalid == VO) {   // fp32 output of a 16-bit residual
#pragma unroll
                for (int j = 0; j < VO; ++j) v[j] = __builtin_fmaf(alpha_res, E::ld(rp + j), v[j]);
            } else {
                for (int j = 0; j < nvalid; ++j) v[j] = __builtin_fmaf(alpha_res, E::ld(rp + j), v[j]);
            }
        }
        typename EO::type* yp = yg + (long long)m * ldy + n;
        if constexpr (WB) *(u32x4*)(lds + row * SO + cv * 16) = pack16<ODT>(v);
        if (vec_y && nvalid == VO) {
            *(u32x4*)yp = pack16<ODT>(v);
        } else {
            for (int j = 0; j < nvalid; ++j) EO::st(yp + j, v[j]);
        }
    }
}

template <int DT, int ODT, int BM, int BN>
struct TileLds {
    static constexpr int SO = BN * Elem<ODT>::BYTES + 16;
    static constexpr int OUT_BYTES = BM * SO;
    static constexpr int PRE_BYTES = ICAF_PRE_WB ? BM * (BN * 4 + 16) : OUT_BYTES;      // pre-term launches stage the tile as fp32 (epilogue)
    static constexpr int REG_BYTES = 2 * (BM + BN) * ROWS;
};

// ===============================================================================================================
// LDS-DMA pipeline
// ===============================================================================================================
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 18, "vmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
}

using lds_ptr_t = __attribute__((address_space(3))) void*;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (global loads and stores share that
// counter on gfx9), which would stall a register prefetch of the next tile at every barrier of the current one.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


// Four consecutive channels of one token row (what a lane's accumulator quad holds) in the storage type: 8 bytes for the 16-bit
// types, 16 bytes for fp32.
template <int DT> struct Quad { using type = u32x2; };
template <> struct Quad<ICAF_F32> { using type = u32x4; };
template <int DT> __device__ __forceinline__ typename Quad<DT>::type pack4(float a, float b, float c, float d) {
    typename Quad<DT>::type v;
    if constexpr (DT == ICAF_F32) { v[0] = __float_as_uint(a); v[1] = __float_as_uint(b); v[2] = __float_as_uint(c); v[3] = __float_as_uint(d); }
    else if constexpr (DT == ICAF_BF16) { v[0] = pack2_bf16(a, b); v[1] = pack2_bf16(c, d); }
    else { v[0] = pack2_f16(a, b); v[1] = pack2_f16(c, d); }
    return v;
}
template <int DT> __device__ __forceinline__ void unpack4(const typename Quad<DT>::type& v, float* f) {
    if constexpr (DT == ICAF_F32) {
        f[0] = __uint_as_float(v[0]); f[1] = __uint_as_float(v[1]); f[2] = __uint_as_float(v[2]); f[3] = __uint_as_float(v[3]);
    } else if constexpr (DT == ICAF_BF16) {
        f[0] = __uint_as_float(v[0] << 16); f[1] = __uint_as_float(v[0] & 0xffff0000u);
        f[2] = __uint_as_float(v[1] << 16); f[3] = __uint_as_float(v[1] & 0xffff0000u);
    } else {
        f[0] = f16_to_f32((unsigned short)(v[0] & 0xffffu)); f[1] = f16_to_f32((unsigned short)(v[0] >> 16));
        f[2] = f16_to_f32((unsigned short)(v[1] & 0xffffu)); f[3] = f16_to_f32((unsigned short)(v[1] >> 16));
    }
}


}  // namespace icaf
