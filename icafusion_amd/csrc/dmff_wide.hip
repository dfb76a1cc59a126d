// DMFF block kernels for the WIDE levels (C = 256 / 512, 16-bit types) on gfx950 — one CrossTransformerBlock iteration (reference
// models/common.py:737-759, CrossAttention :641-687, MLP :704-709) as THREE launches:
//
//   dmff_wide_ln_qkv_kernel    LayerNorm (CrossAttention.LN1 / LN2, :661-662) + the six Linear(C, C) projections (:664-669); a workgroup
//                              = 64 rows x three passes of output channels, or ONE pass where the tiles alone do not fill the chip
//   cross_attn_kernel          (dmff.hip) softmax(q_other k^T / sqrt(dk)) v per (image, direction, head) (:670-681)
//   dmff_wide_proj_mlp_kernel  out-projection + coefficient mix (:682-685, :745-746), the block's shared LayerNorm (:749-750),
//                              MLP Linear(C, 4C) -> GELU(erf) -> Linear(4C, C) and the final mix (:704-709, :751-752)
//
// Why not the two-launch form of dmff_fused.hip: with C >= 256 its 64-row tile + K / V^T of two heads leave ONE four-wave
// workgroup per CU, and its weight stream — through LDS, one barrier per 16 KB slice — measured 1200-1850 cycles per slice against
// 256 cycles of MFMAs (phase clocks at batch 32; the same stream by LDS-DMA with a 4-stage ring and eight waves: still 1200: every
// MFMA needs two fragment reads, 64 KB of LDS reads per 16 KB slice).  Here the GEMMs are organised around what is scarce:
//   * the 64-row token tile is RESIDENT in LDS (read-only during a pass);
//   * EIGHT wavefronts, wave w owns output channels [32 w, 32 w + 32) of a 256-channel pass and BOTH 32-row halves of the tile:
//     one weight fragment feeds two MFMAs;
//   * the weights never touch LDS: a wave reads ITS fragments straight from L2 into registers, from a FRAGMENT-MAJOR copy of the
//     packed matrix ([Np/32][Kp/16][64 lanes][8 elements]: one coalesced 16-byte load per lane = one MFMA A operand), four K slices
//     (64 K elements each) ahead, as one continuous per-wave stream across the passes and phases of the kernel — no ring, no DMA
//     issue slots, NO barrier inside a pass;
//   * LDS traffic per 32 KB of weights: 64 KB of token-fragment reads (8 waves x 8 b128 reads), against 128 + 32 KB before.
// Rounding points are those of the per-layer launches (activations rounded to the storage type between layers, fp32 accumulate,
// two-pass LayerNorm statistics on the rounded values); only fp32 summation orders differ.
//
// Round 4 — hidden split (icaf_dmff_wide_proj_mlp_split): at P5 of yolov5s (C = 512, 100 tokens x 32 images) there are only 100 tiles
// of 64 rows for 256 CUs, and every workgroup streams all 4.7 MB of its modality's weights, which do not stay in a 4 MB L2: PMC showed
// 106 MB fetched per launch for 56.6 MB of distinct bytes and 53 GB/s per CU, the latency of the fabric.  With KS = 2 / 4 a tile is
// shared by KS workgroups, each owning 1/KS of the hidden columns (its slices of W1 and W2 — the XCDs of a modality are dealt over
// the hidden slices, so an XCD's L2 holds W_o + ONE slice: 2.6 MB at KS = 2); every one repeats the cheap front (out-projection +
// LayerNorm: 1/9 of the FLOPs), writes its fc2 partial sums in fp32, and a second, tiny launch (dmff_wide_reduce_kernel) adds the
// partials in FIXED order (deterministic, no atomics, no in-kernel fences: the agent-scope release / acquire of round 3's one-launch
// attempt emptied the L2 of the very weights being streamed) and applies bias + coefficient mix.
#include <type_traits>
#include "icaf_common.h"
#include "conv_common.h"

namespace icaf {

struct WideP {
    const void* x;            // tokens [2][rows][C]: LN + QKV input / residual of the attention mix
    const void* att;          // proj_mlp: attention output [2][rows][C]
    float* part;              // hidden split: fc2 partial sums [KS][2][rows][C] fp32
    const float* x32;         // R32 (round 5): the token stream of the PREVIOUS iteration in fp32 [2][rows][C] (NULL: first iteration, x is all there is)
    float* y32;               // R32: this iteration's tokens in fp32 [2][rows][C] (the 16-bit y is written as well: LayerNorm + QKV read it)
    void* qkv;                // ln_qkv: [2][rows][3C]
    void* y;                  // proj_mlp: element (g, row, c) at y + g * y_gs + row * ldy + c
    const void* wqkv; const float* bqkv;      // FRAGMENT-MAJOR weights [2][Np/32][Kp/16][64][8]; biases fp32 [2][Np]
    const void* wo; const float* bo;
    const void* w1; const float* b1;
    const void* w2; const float* b2;
    const float* ln_a_g[2]; const float* ln_a_b[2];
    const float* ln_m_g; const float* ln_m_b;
    long long wqkv_gs, bqkv_gs, wo_gs, bo_gs, w1_gs, b1_gs, w2_gs, b2_gs, x_gs, y_gs;
    long long rows;
    int C, Kp, Kp4, hid, ldy;
    int qkv_npass;            // ln_qkv: passes (of WPASS output channels) per workgroup, 3 or 1
    float eps_a, eps_m;
    float c_res_a[2], c_acc_a[2], c_res_m[2], c_acc_m[2];
};

constexpr int WROWS = 64;            // token rows per workgroup
constexpr int WDEPTH = 4;            // slices in flight per wave (register ring; every pass is a multiple of it)
// Geometry of a build: NW wavefronts per workgroup, wave w owning output channels [32 w, 32 w + 32) of a pass of 32 NW channels, and
// SL MFMA K steps (16 elements each) per slice of the weight stream.
//   wide levels (C = 256 / 512):  NW = 8, SL = 4  -> 256-channel passes, 64-element slices
//   C = 128 (P3 of yolov5s):      NW = 4, SL = 2  -> 128-channel passes, 32-element slices (a pass over K = 128 is again 4 slices)
template <int NW_, int SL_> struct WG {
    static constexpr int NW = NW_, SL = SL_;
    static constexpr int WT = 64 * NW;           // threads per workgroup
    static constexpr int WPASS = 32 * NW;        // output channels per pass
    // (an MFMA K step is 32 BYTES of a row in every type: 16 elements of a 16-bit type, 8 of fp32 — the parity instantiation)
    template <int DT> static constexpr int kstep() { return 32 / Elem<DT>::BYTES; }          // K elements per MFMA step
    template <int DT> static constexpr int ksl() { return kstep<DT>() * SL; }                // K elements per slice
    template <int DT> static constexpr int nsl2() { return WPASS / ksl<DT>(); }              // slices of an fc2 pass over one hidden chunk
    static constexpr int TPR = WT / WROWS;       // threads per row in the tile LayerNorm
};
using WG8 = WG<8, 4>;
using WG4 = WG<4, 2>;

// One wave's weight stream.  A SEGMENT is one pass's fragments for this wave's 32 channels: n slices of WSL consecutive K steps,
// 64 lanes x 16 bytes each, contiguous in the fragment-major copy.  `next` yields the segment after the current one.
struct WCursor {
    const u32x4* base;       // next slice (lane offset included)
    int left, seg;
};

template <int WSL, class NEXT>
__device__ __forceinline__ void wfetch(u32x4 (&w)[WSL], WCursor& c, const NEXT& next) {
    if (c.left == 0) return;                         // end of the kernel's stream (wave-uniform)
#pragma unroll
    for (int k = 0; k < WSL; ++k) w[k] = c.base[k * 64];
    c.base += WSL * 64;
    if (--c.left == 0) { ++c.seg; next(c); }
}

// acc[t] += W[32 wn + i][k] * A[32 t + j][k] over the next n slices (n % WDEPTH == 0) of the wave's stream; A: LDS tile, row stride SA
template <int DT, int WSL, class NEXT>
__device__ __forceinline__ void wpass(f32x16 (&acc)[2], const unsigned char* A, int SA, int n, u32x4 (&wq)[WDEPTH][WSL], WCursor& c, const NEXT& next) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const unsigned char* a0 = A + (size_t)l31 * SA + hi * 16;
    const unsigned char* a1 = a0 + (size_t)32 * SA;
    for (int s = 0; s < n; s += WDEPTH) {
#pragma unroll
        for (int u = 0; u < WDEPTH; ++u) {
#pragma unroll
            for (int k = 0; k < WSL; ++k) {
                const int off = ((s + u) * WSL + k) * 32;
                const u32x4 x0 = *(const u32x4*)(a0 + off);
                const u32x4 x1 = *(const u32x4*)(a1 + off);
                mma_step<DT>(acc[0], wq[u][k], x0);
                mma_step<DT>(acc[1], wq[u][k], x1);
            }
            wfetch<WSL>(wq[u], c, next);
            __builtin_amdgcn_sched_barrier(0);             // (left alone the scheduler hoists the token-fragment reads of all four slices: 128 registers)
        }
    }
}

// LayerNorm of the 64 rows of an LDS tile in place: TPR (8 or 4) threads per row, two-pass statistics in fp32 on the stored values
// (the arithmetic of layernorm_kernel, dmff.hip).
template <int DT, int TPR>
__device__ __forceinline__ void wide_tile_layernorm(unsigned char* tile, int S, int C, const float* __restrict__ gam, const float* __restrict__ bet, float eps) {
    using E = Elem<DT>;
    const int tid = threadIdx.x, row = tid / TPR, part = tid % TPR;
    const int nv = C / E::VEC;
    unsigned char* r = tile + (size_t)row * S;
    float s = 0.0f;
    for (int v = part; v < nv; v += TPR) {
        float t[E::VEC];
        unpack16<DT>(*(const u32x4*)(r + v * 16), t);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) s += t[j];
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    if constexpr (TPR == 8) s += __shfl_xor(s, 4);
    const float mean = s / (float)C;
    float q = 0.0f;
    for (int v = part; v < nv; v += TPR) {
        float t[E::VEC];
        unpack16<DT>(*(const u32x4*)(r + v * 16), t);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) { const float d = t[j] - mean; q += d * d; }
    }
    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2);
    if constexpr (TPR == 8) q += __shfl_xor(q, 4);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    for (int v = part; v < nv; v += TPR) {
        float t[E::VEC], o[E::VEC];
        unpack16<DT>(*(const u32x4*)(r + v * 16), t);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) o[j] = (t[j] - mean) * rstd * gam[v * E::VEC + j] + bet[v * E::VEC + j];
        *(u32x4*)(r + v * 16) = pack16<DT>(o);
    }
}

// workgroup id -> (modality, index): XCDs 0-3 take modality 0, XCDs 4-7 modality 1 (hardware deals consecutive ids round-robin
// over the 8 XCDs), so an XCD's L2 holds ONE modality's weights — at C = 512 that is 4.7 of the 9.4 MB of a block
__device__ __forceinline__ void wide_place(int bid, int& g, int& idx) { g = (bid & 7) >> 2; idx = (bid >> 3) * 4 + (bid & 3); }
// ... and with the hidden columns split KS ways, the 4 XCDs of a modality are dealt over the hidden slices: an XCD's L2 holds W_o and
// ONE slice of W1 / W2
template <int KS> __device__ __forceinline__ void wide_place_ks(int bid, int& g, int& ks, int& idx) {
    const int x = bid & 7, per = 4 / KS;            // XCDs per (modality, slice)
    g = x >> 2;
    ks = (x & 3) / per;
    idx = (bid >> 3) * per + (x & 3) % per;
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm + QKV projection: a workgroup = 64 token rows x three passes (768 output channels with eight wavefronts, 384 with
// four) of one modality.  grid = 8 * ceil(tiles * C / WPASS / 4)
// ---------------------------------------------------------------------------------------------------------------
template <int DT, class G = WG8>
__global__ __launch_bounds__(G::WT) void dmff_wide_ln_qkv_kernel(const WideP p) {
    using E = Elem<DT>;
    using T = typename E::type;
    constexpr int VEC = E::VEC, EB = E::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = p.C, SA = C * EB + 16;
    unsigned char* tile = smem;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npass = p.qkv_npass;                                    // 3: a workgroup writes q, k AND v columns of its group; 1: one pass each
    const int ngrp = 3 * (C / G::WPASS) / npass;                      // column groups per tile
    const int ntiles = (int)((p.rows + WROWS - 1) / WROWS);
    int g, idx;
    wide_place(blockIdx.x, g, idx);
    if (idx >= ntiles * ngrp) return;
    const int tile_i = idx / ngrp, grp = idx - tile_i * ngrp;
    const long long r0 = (long long)tile_i * WROWS;
    const int ks_row = p.Kp / G::template kstep<DT>(), nsl = C / G::template ksl<DT>();

    const u32x4* wf = (const u32x4*)((const T*)p.wqkv + g * p.wqkv_gs) + lane;
    auto next = [&](WCursor& c) {
        if (c.seg >= npass) { c.left = 0; return; }
        c.base = wf + (long long)((grp * npass + c.seg) * G::NW + wn) * ks_row * 64;
        c.left = nsl;
    };
    WCursor cur; cur.seg = 0; next(cur);
    u32x4 wq[WDEPTH][G::SL];
#pragma unroll
    for (int u = 0; u < WDEPTH; ++u) wfetch<G::SL>(wq[u], cur, next);      // the first weight slices travel while the tokens are normalised

    {
        const T* xg = (const T*)p.x + g * p.x_gs;
        const int nv = C / VEC;
        for (int i = tid; i < WROWS * nv; i += G::WT) {               // raw tokens -> LDS (rows beyond the tensor: clamped, never stored)
            const int row = i / nv, v = i - row * nv;
            long long r = r0 + row;
            r = r < p.rows ? r : p.rows - 1;
            *(u32x4*)(tile + (size_t)row * SA + v * 16) = *(const u32x4*)(xg + r * C + v * VEC);
        }
    }
    lds_barrier();
    wide_tile_layernorm<DT, G::TPR>(tile, SA, C, p.ln_a_g[g], p.ln_a_b[g], p.eps_a);
    lds_barrier();

    const float* bias = p.bqkv + g * p.bqkv_gs;
    const int nout = 3 * C;
    T* out = (T*)p.qkv + (long long)g * p.rows * nout;
    for (int j = 0; j < npass; ++j) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        wpass<DT, G::SL>(acc, tile, SA, nsl, wq, cur, next);
        const int nb = (grp * npass + j) * G::WPASS + wn * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long long row = r0 + t * 32 + l31;
            if (row < p.rows) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nb + 8 * q + 4 * hi;
                    const f32x4 b = *(const f32x4*)(bias + n);
                    *(typename Quad<DT>::type*)(out + row * nout + n) =
                        pack4<DT>(acc[t][4 * q] + b[0], acc[t][4 * q + 1] + b[1], acc[t][4 * q + 2] + b[2], acc[t][4 * q + 3] + b[3]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// out-projection + LayerNorm + MLP: a workgroup = 64 token rows of one modality.  grid = 8 * ceil(tiles / 4)
// ---------------------------------------------------------------------------------------------------------------
// R32 (round 5, blocks with loops > 1): the residual chain x -> x_att -> x' stays in FP32 across iterations — x is read from p.x32 where it exists,
// x_att is kept (and parked) in fp32, x' is written to p.y32 beside the 16-bit y.  Without it every iteration rounds the token stream twice to the
// storage type, and three iterations had used up the 16-bit parity margin of the 3-iteration configuration (0.91 x the reference's own bf16 error).
template <int DT, int NPW, int KS = 1, class G = WG8, bool R32 = false>          // NPW = C / WPASS passes per C-wide product; KS = workgroups sharing a tile (hidden split)
__global__ __launch_bounds__(G::WT) void dmff_wide_proj_mlp_kernel(const WideP p) {
    using E = Elem<DT>;
    using T = typename E::type;
    constexpr int VEC = E::VEC, EB = E::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = p.C, SA = C * EB + 16;
    constexpr int WPASS = G::WPASS, WT = G::WT, NSL2 = G::template nsl2<DT>(), KST = G::template kstep<DT>();
    constexpr int SH = WPASS * EB + 16;
    unsigned char* T0 = smem;                                      // attention output -> later the LayerNorm'ed MLP input
    unsigned char* Hb = T0 + (size_t)WROWS * SA;                   // hidden chunk [64][256]
    float* red = (float*)(Hb + (size_t)WROWS * SH);                // [NW][64] row partial sums of the channel groups
    float* b1s = red + G::NW * 64;                                     // fc1 bias (hid floats)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = (int)((p.rows + WROWS - 1) / WROWS);
    int g, tile_i, ksl = 0;
    if constexpr (KS == 1) wide_place(blockIdx.x, g, tile_i);
    else wide_place_ks<KS>(blockIdx.x, g, ksl, tile_i);
    if (tile_i >= ntiles) return;
    const long long r0 = (long long)tile_i * WROWS;
    const int ks_row = p.Kp / KST, ks_row4 = p.Kp4 / KST, nsl = C / G::template ksl<DT>(), nchunk = p.hid / WPASS / KS, chunk0 = ksl * nchunk;

    // the wave's stream: NPW out-projection passes, then per hidden chunk one fc1 pass and NPW fc2 passes
    const u32x4* wof = (const u32x4*)((const T*)p.wo + g * p.wo_gs) + lane;
    const u32x4* w1f = (const u32x4*)((const T*)p.w1 + g * p.w1_gs) + lane;
    const u32x4* w2f = (const u32x4*)((const T*)p.w2 + g * p.w2_gs) + lane;
    auto next = [&](WCursor& c) {
        const int s = c.seg;
        if (s < NPW) { c.base = wof + (long long)(s * G::NW + wn) * ks_row * 64; c.left = nsl; return; }
        const int m = s - NPW, chunk = m / (1 + NPW), r = m - chunk * (1 + NPW);
        if (chunk >= nchunk) { c.left = 0; return; }
        if (r == 0) { c.base = w1f + (long long)((chunk0 + chunk) * G::NW + wn) * ks_row * 64; c.left = nsl; }
        else { c.base = w2f + ((long long)((r - 1) * G::NW + wn) * ks_row4 + (chunk0 + chunk) * (WPASS / KST)) * 64; c.left = NSL2; }
    };
    WCursor cur; cur.seg = 0; next(cur);
    u32x4 wq[WDEPTH][G::SL];
#pragma unroll
    for (int u = 0; u < WDEPTH; ++u) wfetch<G::SL>(wq[u], cur, next);      // the first weight slices travel while the tile is loaded

    {
        const T* att = (const T*)p.att + (long long)g * p.rows * C;
        const int nv = C / VEC;
        for (int i = tid; i < WROWS * nv; i += WT) {               // (rows beyond the tensor: clamped, never stored)
            const int row = i / nv, v = i - row * nv;
            long long r = r0 + row;
            r = r < p.rows ? r : p.rows - 1;
            *(u32x4*)(T0 + (size_t)row * SA + v * 16) = *(const u32x4*)(att + r * C + v * VEC);
        }
        const float* b1 = p.b1 + g * p.b1_gs;
        for (int i = tid; i < p.hid / KS; i += WT) b1s[i] = b1[chunk0 * WPASS + i];
    }
    lds_barrier();

    bool rok[2];
    const T* xres[2];
    T* yrow[2];
    const float* x32row[2];
    float* y32row[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long long row = r0 + t * 32 + l31;
        rok[t] = row < p.rows;
        const long long grow = rok[t] ? row : p.rows - 1;
        xres[t] = (const T*)p.x + g * p.x_gs + grow * C;
        yrow[t] = (T*)p.y + g * p.y_gs + grow * p.ldy;
        x32row[t] = (R32 && p.x32) ? p.x32 + ((long long)g * p.rows + grow) * C : nullptr;
        y32row[t] = R32 ? p.y32 + ((long long)g * p.rows + grow) * C : nullptr;
    }
    using XQ = typename std::conditional<R32, f32x4, typename Quad<DT>::type>::type;      // x_att per (pass, row half, channel quad): fp32, or rounded to the storage type
    auto xq_pack = [](float a, float b, float c, float d) -> XQ {
        if constexpr (R32) return f32x4{a, b, c, d};
        else return pack4<DT>(a, b, c, d);
    };
    auto xq_unpack = [](const XQ& q, float* f) {
        if constexpr (R32) { f[0] = q[0]; f[1] = q[1]; f[2] = q[2]; f[3] = q[3]; }
        else unpack4<DT>(q, f);
    };

    // ---- out-projection + coefficient mix: x_att = c_res * x + c_acc * (att W_o^T + b), rounded to the storage type, in registers ----
    constexpr bool PARK = NPW >= 2 || KS > 1;          // (hidden split: the reduce launch reads x_att from the output rows)
    XQ xatt[NPW][2][4];
    {
        const float* bias = p.bo + g * p.bo_gs;
        const float ca = p.c_acc_a[g], cr = p.c_res_a[g];
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            wpass<DT, G::SL>(acc, T0, SA, nsl, wq, cur, next);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = i * WPASS + wn * 32 + 8 * q + 4 * hi;
                    const f32x4 bv = *(const f32x4*)(bias + n);
                    float rv[4], v[4];
                    if (R32 && x32row[t]) { const f32x4 r4 = *(const f32x4*)(x32row[t] + n); rv[0] = r4[0]; rv[1] = r4[1]; rv[2] = r4[2]; rv[3] = r4[3]; }
                    else unpack4<DT>(*(const typename Quad<DT>::type*)(xres[t] + n), rv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(cr, rv[j], (acc[t][4 * q + j] + bv[j]) * ca);
                    xatt[i][t][q] = xq_pack(v[0], v[1], v[2], v[3]);
                }
        }
    }
    // ---- the block's shared LayerNorm over x_att, from registers: row sums = this lane's channels + the other lane half (shuffle) +
    //      the other channel groups (LDS); the normalised tile overwrites T0 — every wave has passed a barrier after its last
    //      out-projection read by then ----
    {
        float mean[2], rstd[2];
        float sum[2] = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NPW; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
                    xq_unpack(xatt[i][t][q], v);
                    sum[t] += (v[0] + v[1]) + (v[2] + v[3]);
                }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sum[t] += __shfl_xor(sum[t], 32);
            if (hi == 0) red[wn * 64 + t * 32 + l31] = sum[t];
        }
        lds_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < G::NW; ++w) s += red[w * 64 + t * 32 + l31];
            mean[t] = s / (float)C;
        }
        lds_barrier();
        float sq[2] = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < NPW; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
                    xq_unpack(xatt[i][t][q], v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d = v[j] - mean[t]; sq[t] += d * d; }
                }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            sq[t] += __shfl_xor(sq[t], 32);
            if (hi == 0) red[wn * 64 + t * 32 + l31] = sq[t];
        }
        lds_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < G::NW; ++w) s += red[w * 64 + t * 32 + l31];
            rstd[t] = 1.0f / sqrtf(s / (float)C + p.eps_m);
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = i * WPASS + wn * 32 + 8 * q + 4 * hi;
                const f32x4 gv = *(const f32x4*)(p.ln_m_g + n), bv = *(const f32x4*)(p.ln_m_b + n);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float v[4], o[4];
                    xq_unpack(xatt[i][t][q], v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean[t]) * rstd[t] * gv[j] + bv[j];
                    *(typename Quad<DT>::type*)(T0 + (size_t)(t * 32 + l31) * SA + n * EB) = pack4<DT>(o[0], o[1], o[2], o[3]);
                    // C = 512: x_att is needed once more, as the residual of the final mix — parked in the workgroup's own rows of the
                    // output tensor (read back by the same lane) instead of holding 32 more registers through the MLP (spills otherwise)
                    if constexpr (PARK) {
                        if (rok[t] && ksl == 0) {
                            if constexpr (R32) *(f32x4*)(y32row[t] + n) = xatt[i][t][q];          // (fp32 x_att parks in the fp32 output rows)
                            else *(typename Quad<DT>::type*)(yrow[t] + n) = xatt[i][t][q];
                        }
                    }
                }
            }
        lds_barrier();
    }
    // ---- MLP in 256-column hidden chunks: H = GELU(n2 W1_chunk^T + b1) -> LDS, then acc2 += H W2[:, chunk]^T ----
    f32x16 acc2[NPW][2];
#pragma unroll
    for (int i = 0; i < NPW; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][t][r] = 0.0f;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        wpass<DT, G::SL>(acc, T0, SA, nsl, wq, cur, next);
        lds_barrier();                                             // every wave is done with the previous chunk's H
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * 32 + 8 * q + 4 * hi;
                const f32x4 bv = *(const f32x4*)(b1s + chunk * WPASS + nl);
                *(typename Quad<DT>::type*)(Hb + (size_t)(t * 32 + l31) * SH + nl * EB) =
                    pack4<DT>(apply_act<ICAF_ACT_GELU, DT>(acc[t][4 * q] + bv[0]), apply_act<ICAF_ACT_GELU, DT>(acc[t][4 * q + 1] + bv[1]),
                              apply_act<ICAF_ACT_GELU, DT>(acc[t][4 * q + 2] + bv[2]), apply_act<ICAF_ACT_GELU, DT>(acc[t][4 * q + 3] + bv[3]));
            }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < NPW; ++i) wpass<DT, G::SL>(acc2[i], Hb, SH, NSL2, wq, cur, next);
    }
    // ---- output: x' = c_res2 * x_att + c_acc2 * (mlp + b2) ----
    if constexpr (KS > 1) {                            // hidden split: this workgroup's share of the fc2 sums, fp32; dmff_wide_reduce_kernel finishes
        float* part = p.part + ((long long)(ksl * 2 + g) * p.rows) * C;
#pragma unroll
        for (int i = 0; i < NPW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = i * WPASS + wn * 32 + 8 * q + 4 * hi;
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (rok[t])
                        *(f32x4*)(part + (r0 + t * 32 + l31) * C + n) = f32x4{acc2[i][t][4 * q], acc2[i][t][4 * q + 1], acc2[i][t][4 * q + 2], acc2[i][t][4 * q + 3]};
            }
    } else {
        const float* b2 = p.b2 + g * p.b2_gs;
        const float ca = p.c_acc_m[g], cr = p.c_res_m[g];
#pragma unroll
        for (int i = 0; i < NPW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = i * WPASS + wn * 32 + 8 * q + 4 * hi;
                const f32x4 bv = *(const f32x4*)(b2 + n);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float rv[4], v[4];
                    if constexpr (PARK && R32) { const f32x4 r4 = *(const f32x4*)(y32row[t] + n); rv[0] = r4[0]; rv[1] = r4[1]; rv[2] = r4[2]; rv[3] = r4[3]; }
                    else if constexpr (PARK) unpack4<DT>(*(const typename Quad<DT>::type*)(yrow[t] + n), rv);     // (clamped row when !rok: never stored)
                    else xq_unpack(xatt[i][t][q], rv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(cr, rv[j], (acc2[i][t][4 * q + j] + bv[j]) * ca);
                    if (rok[t]) {
                        *(typename Quad<DT>::type*)(yrow[t] + n) = pack4<DT>(v[0], v[1], v[2], v[3]);
                        if constexpr (R32) *(f32x4*)(y32row[t] + n) = f32x4{v[0], v[1], v[2], v[3]};
                    }
                }
            }
    }
}

// y = c_res2 * x_att + c_acc2 * (sum_ks part[ks] + b2): x_att was parked in y by the ks = 0 workgroups; the partial sums are added in
// slice order (a fixed association: the result does not depend on which workgroup finished first).  One thread = 4 channels of one row.
template <int DT, int KS, bool R32 = false>
__global__ __launch_bounds__(256) void dmff_wide_reduce_kernel(const WideP p) {
    using T = typename Elem<DT>::type;
    const int C = p.C, nq = C >> 2;
    const long long total = 2 * p.rows * nq;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int qn = (int)(idx % nq);
        const long long gr = idx / nq;
        const int g = (int)(gr / p.rows);
        const long long row = gr - (long long)g * p.rows;
        const int n = qn * 4;
        f32x4 sum = *(const f32x4*)(p.part + ((long long)g * p.rows + row) * C + n);
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            const f32x4 v = *(const f32x4*)(p.part + ((long long)(k * 2 + g) * p.rows + row) * C + n);
            sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
        }
        const f32x4 bv = *(const f32x4*)(p.b2 + g * p.b2_gs + n);
        T* y = (T*)p.y + g * p.y_gs + row * p.ldy + n;
        float* y32 = R32 ? p.y32 + ((long long)g * p.rows + row) * C + n : nullptr;
        float rv[4], v[4];
        if constexpr (R32) { const f32x4 r4 = *(const f32x4*)y32; rv[0] = r4[0]; rv[1] = r4[1]; rv[2] = r4[2]; rv[3] = r4[3]; }      // x_att parked in fp32
        else unpack4<DT>(*(const typename Quad<DT>::type*)y, rv);
        const float ca = p.c_acc_m[g], cr = p.c_res_m[g];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(cr, rv[j], (sum[j] + bv[j]) * ca);
        *(typename Quad<DT>::type*)y = pack4<DT>(v[0], v[1], v[2], v[3]);
        if constexpr (R32) *(f32x4*)y32 = f32x4{v[0], v[1], v[2], v[3]};
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int wide_fill(const icaf_dmff_args* a, WideP& p, const char* who) {
    if (!a || !a->x) return fail(ICAF_ERR_ARG, "%s: null pointer", who);
    // fp32: the PARITY instantiation of the C = 128 build (v_mfma_f32_32x32x2_f32, erff): the same indexing, masks, LayerNorm and hidden-chunk loop as the
    // 16-bit kernels the bench runs, held to the reference's DMFF goldens at fp32 accuracy (tests/test_gpu_dmff_fused.py)
    if (a->dtype != ICAF_BF16 && a->dtype != ICAF_F16 && !(a->dtype == ICAF_F32 && a->C == 128))
        return fail(ICAF_ERR_UNSUPPORTED, "%s: 16-bit types (fp32: the C = 128 parity build only; dtype %d, C = %d)", who, a->dtype, a->C);
    if (a->B < 1 || a->N < 1) return fail(ICAF_ERR_ARG, "%s: bad B/N", who);
    if (a->C != 128 && a->C != 256 && a->C != 512) return fail(ICAF_ERR_UNSUPPORTED, "%s: C=%d (built for 128 [four wavefronts, 128-channel passes] and 256 / 512 [eight, 256-channel passes])", who, a->C);
    if (a->Kp < a->C || a->Kp % (a->dtype == ICAF_F32 ? 32 : 64)) return fail(ICAF_ERR_ARG, "%s: Kp=%d", who, a->Kp);
    p.x = a->x; p.qkv = a->qkv; p.y = a->y; p.att = nullptr; p.part = nullptr;
    p.x32 = a->x32; p.y32 = a->y32;
    if (a->x32 && !a->y32) return fail(ICAF_ERR_ARG, "%s: x32 without y32 (the fp32 residual stream is read AND written by every iteration but the first)", who);
    if (a->y32 && a->dtype == ICAF_F32) return fail(ICAF_ERR_ARG, "%s: the fp32 residual stream belongs to the 16-bit builds", who);
    if ((a->x32 && ((uintptr_t)a->x32 & 15)) || (a->y32 && ((uintptr_t)a->y32 & 15))) return fail(ICAF_ERR_ARG, "%s: x32 / y32 must be 16-byte aligned", who);
    p.wqkv = a->wqkv; p.bqkv = a->bqkv; p.wo = a->wo; p.bo = a->bo; p.w1 = a->w1; p.b1 = a->b1; p.w2 = a->w2; p.b2 = a->b2;
    p.ln_a_g[0] = a->ln_attn_gamma[0]; p.ln_a_g[1] = a->ln_attn_gamma[1]; p.ln_a_b[0] = a->ln_attn_beta[0]; p.ln_a_b[1] = a->ln_attn_beta[1];
    p.ln_m_g = a->ln_mlp_gamma; p.ln_m_b = a->ln_mlp_beta;
    p.wqkv_gs = a->wqkv_gs; p.bqkv_gs = a->bqkv_gs; p.wo_gs = a->wo_gs; p.bo_gs = a->bo_gs; p.w1_gs = a->w1_gs; p.b1_gs = a->b1_gs;
    p.w2_gs = a->w2_gs; p.b2_gs = a->b2_gs; p.x_gs = a->x_gs; p.y_gs = a->y_gs;
    p.rows = (long long)a->B * a->N; p.C = a->C; p.Kp = a->Kp; p.Kp4 = a->Kp4; p.hid = a->hidden; p.ldy = a->ldy;
    // LN + QKV: three output-channel passes per workgroup by default.  One pass per workgroup (three times the workgroups for the levels whose
    // tiles alone do not fill the chip) was measured and LOSES at every level — P5 of yolov5s 26.7 -> 34.4 us, P4 19.6 -> 25.5, yolov5l P4
    // 48.9 -> 65.5: the repeated LayerNorm and workgroup prologues cost more than the idle CUs.  a->reserved = 1 keeps it reachable for A/B.
    p.qkv_npass = a->reserved == 1 ? 1 : 3;
    p.eps_a = a->eps_attn; p.eps_m = a->eps_mlp;
    for (int g = 0; g < 2; ++g) {
        p.c_res_a[g] = a->coef_res_attn[g]; p.c_acc_a[g] = a->coef_acc_attn[g];
        p.c_res_m[g] = a->coef_res_mlp[g]; p.c_acc_m[g] = a->coef_acc_mlp[g];
    }
    return ICAF_OK;
}

template <int DT, class G>
static int launch_wide_ln_qkv(const WideP& p, hipStream_t s) {
    const size_t lds = (size_t)WROWS * (p.C * Elem<DT>::BYTES + 16);
    ICAF_LDS_OPTIN((dmff_wide_ln_qkv_kernel<DT, G>), lds);        // (size checked on EVERY call, attribute raised per device as needed)
    const long long work = ((p.rows + WROWS - 1) / WROWS) * (3 * (p.C / G::WPASS) / p.qkv_npass);
    hipLaunchKernelGGL((dmff_wide_ln_qkv_kernel<DT, G>), dim3((unsigned)(8 * ((work + 3) / 4))), dim3(G::WT), lds, s, p);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <class G>
static size_t wide_proj_mlp_lds(int C, int hid, int eb = 2) {
    return (size_t)WROWS * (C * eb + 16) + (size_t)WROWS * (G::WPASS * eb + 16) + G::NW * 64 * sizeof(float) + (size_t)hid * sizeof(float);
}

template <int DT, int NPW, class G, bool R32 = false>
static int launch_wide_proj_mlp(const WideP& p, hipStream_t s) {
    const size_t lds = wide_proj_mlp_lds<G>(p.C, p.hid, Elem<DT>::BYTES);
    ICAF_LDS_OPTIN((dmff_wide_proj_mlp_kernel<DT, NPW, 1, G, R32>), lds);
    const long long ntiles = (p.rows + WROWS - 1) / WROWS;
    hipLaunchKernelGGL((dmff_wide_proj_mlp_kernel<DT, NPW, 1, G, R32>), dim3((unsigned)(8 * ((ntiles + 3) / 4))), dim3(G::WT), lds, s, p);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <int DT, int NPW, int KS, bool R32 = false>
static int launch_wide_proj_mlp_split(const WideP& p, hipStream_t s) {
    const size_t lds = wide_proj_mlp_lds<WG8>(p.C, p.hid / KS);
    ICAF_LDS_OPTIN((dmff_wide_proj_mlp_kernel<DT, NPW, KS, WG8, R32>), lds);
    const long long ntiles = (p.rows + WROWS - 1) / WROWS, per = 4 / KS;
    hipLaunchKernelGGL((dmff_wide_proj_mlp_kernel<DT, NPW, KS, WG8, R32>), dim3((unsigned)(8 * ((ntiles + per - 1) / per))), dim3(WG8::WT), lds, s, p);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <int DT>
static int dispatch_wide_split(const WideP& p, int ksplit, hipStream_t s) {
    if (p.y32) {                                   // fp32 residual stream (loops > 1): built for the two-way split only
        if (ksplit != 2) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_wide_proj_mlp_split: the fp32 residual stream is built for ksplit = 2");
        return p.C == 256 ? launch_wide_proj_mlp_split<DT, 1, 2, true>(p, s) : launch_wide_proj_mlp_split<DT, 2, 2, true>(p, s);
    }
    if (p.C == 256) return ksplit == 2 ? launch_wide_proj_mlp_split<DT, 1, 2>(p, s) : launch_wide_proj_mlp_split<DT, 1, 4>(p, s);
    return ksplit == 2 ? launch_wide_proj_mlp_split<DT, 2, 2>(p, s) : launch_wide_proj_mlp_split<DT, 2, 4>(p, s);
}

template <int DT>
static int dispatch_wide_ln_qkv(const WideP& p, hipStream_t s) {
    return p.C == 128 ? launch_wide_ln_qkv<DT, WG4>(p, s) : launch_wide_ln_qkv<DT, WG8>(p, s);
}

template <int DT>
static int dispatch_wide_proj_mlp(const WideP& p, hipStream_t s) {
    if (p.y32) {                                   // fp32 residual stream (loops > 1)
        if (p.C == 128) return launch_wide_proj_mlp<DT, 1, WG4, true>(p, s);
        if (p.C == 256) return launch_wide_proj_mlp<DT, 1, WG8, true>(p, s);
        // (C = 512 unsplit with fp32 x_att: 258 registers — not built; callers keep the 16-bit stream there: ops / CrossTransformerBlock.emit_tokens)
        return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_wide_proj_mlp: the fp32 residual stream at C = 512 is built for the hidden split (ksplit = 2) only");
    }
    if (p.C == 128) return launch_wide_proj_mlp<DT, 1, WG4>(p, s);
    return p.C == 256 ? launch_wide_proj_mlp<DT, 1, WG8>(p, s) : launch_wide_proj_mlp<DT, 2, WG8>(p, s);
}

template <int DT>
static int launch_wide_reduce(const WideP& p, int ksplit, hipStream_t s) {
    const long long items = 2 * p.rows * (p.C / 4);
    long long blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (p.y32) {
        if (ksplit != 2) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_wide_reduce: the fp32 residual stream is built for ksplit = 2");
        hipLaunchKernelGGL((dmff_wide_reduce_kernel<DT, 2, true>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    } else if (ksplit == 2) hipLaunchKernelGGL((dmff_wide_reduce_kernel<DT, 2>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((dmff_wide_reduce_kernel<DT, 4>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

}  // namespace icaf

using namespace icaf;

extern "C" int icaf_dmff_wide_ln_qkv(const icaf_dmff_args* a, icaf_stream_t s) {
    WideP p{};          // (value-initialised: a field a later edit forgets to fill is zero, not stack garbage)
    int st = wide_fill(a, p, "icaf_dmff_wide_ln_qkv");
    if (st) return st;
    if (!a->qkv || !a->wqkv || !a->bqkv || !a->ln_attn_gamma[0] || !a->ln_attn_gamma[1] || !a->ln_attn_beta[0] || !a->ln_attn_beta[1])
        return fail(ICAF_ERR_ARG, "icaf_dmff_wide_ln_qkv: null pointer");
    if (a->dtype == ICAF_F32) return launch_wide_ln_qkv<ICAF_F32, WG4>(p, S(s));
    return a->dtype == ICAF_BF16 ? dispatch_wide_ln_qkv<ICAF_BF16>(p, S(s)) : dispatch_wide_ln_qkv<ICAF_F16>(p, S(s));
}

extern "C" int icaf_dmff_wide_proj_mlp(const icaf_dmff_args* a, const void* att, icaf_stream_t s) {
    WideP p{};          // (value-initialised: a field a later edit forgets to fill is zero, not stack garbage)
    int st = wide_fill(a, p, "icaf_dmff_wide_proj_mlp");
    if (st) return st;
    if (!att || !a->y || !a->wo || !a->bo || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->ln_mlp_gamma || !a->ln_mlp_beta) return fail(ICAF_ERR_ARG, "icaf_dmff_wide_proj_mlp: null pointer");
    const int wpass = a->C == 128 ? WG4::WPASS : WG8::WPASS;
    if (a->hidden % wpass || a->hidden < wpass || a->Kp4 < a->hidden || a->Kp4 % (a->dtype == ICAF_F32 ? 32 : 64)) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_wide_proj_mlp: hidden width %d must be a multiple of %d", a->hidden, wpass);
    if (a->ldy < a->C || a->ldy % 4) return fail(ICAF_ERR_ARG, "icaf_dmff_wide_proj_mlp: ldy=%d", a->ldy);
    p.att = att;
    if (a->dtype == ICAF_F32) return launch_wide_proj_mlp<ICAF_F32, 1, WG4>(p, S(s));
    return a->dtype == ICAF_BF16 ? dispatch_wide_proj_mlp<ICAF_BF16>(p, S(s)) : dispatch_wide_proj_mlp<ICAF_F16>(p, S(s));
}

extern "C" int icaf_dmff_wide_proj_mlp_split(const icaf_dmff_args* a, const void* att, float* partial, int ksplit, icaf_stream_t s) {
    WideP p{};          // (value-initialised: a field a later edit forgets to fill is zero, not stack garbage)
    int st = wide_fill(a, p, "icaf_dmff_wide_proj_mlp_split");
    if (st) return st;
    if (!att || !partial || !a->y || !a->wo || !a->bo || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->ln_mlp_gamma || !a->ln_mlp_beta) return fail(ICAF_ERR_ARG, "icaf_dmff_wide_proj_mlp_split: null pointer");
    if (ksplit != 2 && ksplit != 4) return fail(ICAF_ERR_ARG, "icaf_dmff_wide_proj_mlp_split: ksplit %d (2 or 4)", ksplit);
    if (a->C == 128 || a->dtype == ICAF_F32) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_wide_proj_mlp_split: C = 128 has no hidden split (its weights are 0.3 MB)");
    if (a->hidden % (WG8::WPASS * ksplit) || a->Kp4 < a->hidden || a->Kp4 % 64) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_wide_proj_mlp_split: hidden width %d must be a multiple of %d", a->hidden, WG8::WPASS * ksplit);
    if (a->ldy < a->C || a->ldy % 4) return fail(ICAF_ERR_ARG, "icaf_dmff_wide_proj_mlp_split: ldy=%d", a->ldy);
    p.att = att;
    p.part = partial;
    return a->dtype == ICAF_BF16 ? dispatch_wide_split<ICAF_BF16>(p, ksplit, S(s)) : dispatch_wide_split<ICAF_F16>(p, ksplit, S(s));
}

extern "C" int icaf_dmff_wide_reduce(const icaf_dmff_args* a, const float* partial, int ksplit, icaf_stream_t s) {
    WideP p{};          // (value-initialised: a field a later edit forgets to fill is zero, not stack garbage)
    int st = wide_fill(a, p, "icaf_dmff_wide_reduce");
    if (st) return st;
    if (!partial || !a->y || !a->b2) return fail(ICAF_ERR_ARG, "icaf_dmff_wide_reduce: null pointer");
    if (ksplit != 2 && ksplit != 4) return fail(ICAF_ERR_ARG, "icaf_dmff_wide_reduce: ksplit %d (2 or 4)", ksplit);
    if (a->dtype == ICAF_F32) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_wide_reduce: 16-bit types only");
    if (a->ldy < a->C || a->ldy % 4) return fail(ICAF_ERR_ARG, "icaf_dmff_wide_reduce: ldy=%d", a->ldy);
    p.part = const_cast<float*>(partial);
    return a->dtype == ICAF_BF16 ? launch_wide_reduce<ICAF_BF16>(p, ksplit, S(s)) : launch_wide_reduce<ICAF_F16>(p, ksplit, S(s));
}
