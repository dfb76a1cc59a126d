// DMFF (Dual-Modality Feature Fusion / TransformerFusionBlock, reference models/common.py:762-865) kernels for gfx950:
//   pool_tokens      AdaptivePool2d avg+max, LearnableWeights mix, + positional embedding     (HBM-bound)
//   layernorm        per-token LayerNorm over C                                               (HBM-bound)
//   cross_attention  the two crossed softmax(Q K^T / sqrt(dk)) V products, all heads          (MFMA + VALU)
//   upsample_merge   bilinear resize of the token maps + residual + channel concat            (HBM-bound)
// The Linear layers around them (QKV / out-proj / MLP) run on the implicit-GEMM kernel (igemm.hip) with the
// LearnableCoefficient mixes folded into its epilogue.
#include <cstdlib>
#include "icaf_common.h"
#include "attn_core.h"

namespace icaf {

static inline unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------------------------
// tokens[g][b][n][c] = w1_g * avgpool + w2_g * maxpool + pos_g[n][c]     (reference models/common.py:817-823,868-891)
// ---------------------------------------------------------------------------------------------------------------
// I32: the flat index fits 31 bits (every configuration in use) and is taken apart with FastDiv (icaf_common.h) instead of 64-bit
// divisions; the window is walked row by row instead of dividing the tap index by kw — at 4x4 windows (P3) the index arithmetic
// was a third of the kernel's instructions.
struct PoolDiv { FastDiv nv, N, tw; };
template <int DT, bool I32>
__global__ __launch_bounds__(256) void pool_tokens_kernel(const typename Elem<DT>::type* __restrict__ f0, int ld0,
                                                          const typename Elem<DT>::type* __restrict__ f1, int ld1,
                                                          const float* __restrict__ pos0, const float* __restrict__ pos1,
                                                          typename Elem<DT>::type* __restrict__ tok, int B, int H, int W, int C, int th,
                                                          int tw, int kh, int kw, int sh, int sw, float w1_0, float w2_0, float w1_1,
                                                          float w2_1, PoolDiv dv) {
    using E = Elem<DT>;
    const int nv = C / E::VEC, N = th * tw;
    const long long per_g = (long long)B * N * nv, total = 2 * per_g;
    const float inv_area = 1.0f / (float)(kh * kw);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int g = idx >= per_g;
        const long long r = idx - (g ? per_g : 0);
        int v, n, b, oy, ox;
        if constexpr (I32) {
            unsigned int bn, uv, un, ub, uoy, uox;
            fd_divmod((unsigned int)r, dv.nv, bn, uv);
            fd_divmod(bn, dv.N, ub, un);
            fd_divmod(un, dv.tw, uoy, uox);
            v = (int)uv; n = (int)un; b = (int)ub; oy = (int)uoy; ox = (int)uox;
        } else {
            v = (int)(r % nv);
            const long long bn = r / nv;
            n = (int)(bn % N); b = (int)(bn / N);
            oy = n / tw; ox = n - oy * tw;
        }
        const typename E::type* f = g ? f1 : f0;
        const int ld = g ? ld1 : ld0;
        float sum[E::VEC], mx[E::VEC];
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) { sum[j] = 0.0f; mx[j] = -INFINITY; }
        // the window is walked in (dy, dx) order, four loads in flight at a time: with one load per iteration the
        // 100-pixel windows of P4 (10x10, stride 2) were a chain of 100 L2 round trips per thread
        // (same (dy, dx) order and the same additions as before: the sums are bit-identical)
        const typename E::type* f00 = f + (((long long)b * H + oy * sh) * W + ox * sw) * ld + v * E::VEC;
        for (int dy = 0; dy < kh; ++dy) {
            const typename E::type* frow = f00 + (long long)dy * W * ld;
            for (int dx0 = 0; dx0 < kw; dx0 += 4) {
                u32x4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (dx0 + u < kw) raw[u] = *(const u32x4*)(frow + (long long)(dx0 + u) * ld);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (dx0 + u < kw) {
                        float t[E::VEC];
                        unpack16<DT>(raw[u], t);
#pragma unroll
                        for (int j = 0; j < E::VEC; ++j) { sum[j] += t[j]; mx[j] = fmaxf(mx[j], t[j]); }
                    }
                }
            }
        }
        const float w1 = g ? w1_1 : w1_0, w2 = g ? w2_1 : w2_0;
        const float* pos = (g ? pos1 : pos0) + (long long)n * C + v * E::VEC;
        float o[E::VEC];
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) o[j] = (sum[j] * inv_area) * w1 + mx[j] * w2 + pos[j];
        *(u32x4*)(tok + (((long long)g * B + b) * N + n) * C + v * E::VEC) = pack16<DT>(o);
    }
}

// Overlapping windows (AdaptivePool2d picks kernel = H - (out-1)*stride > stride whenever H is not a multiple of the token
// grid: P4 of a 640x640 input pools 10x10 windows at stride 2): with one thread per output element every input pixel is
// read kh*kw/(sh*sw) times — 25x the tensor through L2 for that level, which took longer than the P3 level four times its
// size.  Sum and max are separable, so a workgroup owning one token ROW (g, b, oy)
//   1. reduces its kh input rows vertically, one (column, 16-byte channel vector) item per thread and coalesced across
//      the vectors of a pixel, into fp32 column sums / maxima in LDS  (each pixel is read kh/sh times, not kh*kw/(sh*sw));
//   2. reduces kw columns per token horizontally from LDS, applies the LearnableWeights mix and the positional embedding.
// Consecutive token rows (which share kh - sh input rows) run on ONE XCD, so the re-reads hit its L2.
// 1024 threads and all R (>= kh for the windows in use) row loads of an item in flight: with 256 threads and four rows at a time a
// workgroup was a chain of 15 dependent L2 round trips (5 items x 3 batches) and the P4 level took 42 us for 52 MB.  The loads are
// unconditional (row index clamped; the duplicates hit L1) — a load under `if (d < kh)` makes the compiler wait for each one.
// TR token rows per workgroup (round 5): consecutive token rows share kh - sh input rows, so a workgroup that owns TR of them loads
// kh + (TR - 1) * sh rows ONCE per item and reduces them into TR column sums / maxima — P4 of a 640x640 input (10x10 windows, stride 2):
// 12 rows for two token rows instead of 2 x 10, half as many workgroups (two rounds of the chip instead of four), and the 1024 outputs of
// the horizontal phase fill the workgroup.  The column maxima are kept in the STORAGE type (a maximum of bf16 / f16 values is one of
// them: exact), 16 bytes per item beside the V fp32 sums: 6 bytes per (column, channel) instead of 8, which is what lets two token rows
// of P4 (and one of yolov5l's 512-channel P4) fit the 160 KB.  Same additions in the same (dy, dx) order per token as before.
template <int DT, int R, int TR>
__global__ __launch_bounds__(1024) void pool_tokens_rows_kernel(const typename Elem<DT>::type* __restrict__ f0, int ld0,
                                                                const typename Elem<DT>::type* __restrict__ f1, int ld1,
                                                                const float* __restrict__ pos0, const float* __restrict__ pos1,
                                                                typename Elem<DT>::type* __restrict__ tok, int B, int H, int W, int C,
                                                                int th, int tw, int kh, int kw, int sh, int sw, float w1_0, float w2_0,
                                                                float w1_1, float w2_1) {
    using E = Elem<DT>;
    constexpr int V = E::VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int nv = C / V, nitem = W * nv;
    float* csum = (float*)lds_raw;                                   // [TR][W * nv][V] fp32
    u32x4* cmax = (u32x4*)(csum + (size_t)TR * nitem * V);           // [TR][W * nv] one packed vector each
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int qd = nblk >> 3, rm = nblk & 7, xcd = bid & 7, ix = bid >> 3;
    const int row = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + ix;      // (as conv_common.h: xcd_tile)
    const int thb = (th + TR - 1) / TR;                              // blocks of TR token rows per image
    const int g = row / (B * thb), r = row - g * (B * thb), b = r / thb, oy0 = (r - b * thb) * TR;
    const int ntr = th - oy0 < TR ? th - oy0 : TR;                   // token rows of this block (the last block of an image may be short)
    const int nrow = kh + (ntr - 1) * sh;                            // input rows the block reads
    const typename E::type* f = g ? f1 : f0;
    const int ld = g ? ld1 : ld0;
    const typename E::type* frow = f + ((long long)b * H + oy0 * sh) * W * ld;
    const long long rstride = (long long)W * ld;
    for (int item = threadIdx.x; item < nitem; item += blockDim.x) {
        const int x = item / nv, v = item - x * nv;
        const typename E::type* p0 = frow + (long long)x * ld + v * V;
        float sum[TR][V], mx[TR][V];
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int j = 0; j < V; ++j) { sum[t][j] = 0.0f; mx[t][j] = -INFINITY; }
        for (int d0 = 0; d0 < nrow; d0 += R) {
            u32x4 raw[R];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int d = d0 + u < nrow ? d0 + u : nrow - 1;
                raw[u] = *(const u32x4*)(p0 + d * rstride);
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                float t[V];
                unpack16<DT>(raw[u], t);
#pragma unroll
                for (int k = 0; k < TR; ++k) {
                    const int dd = d0 + u - k * sh;                  // the row's index inside token row k's window
                    if (dd >= 0 && dd < kh && k < ntr) {             // wave-uniform: a BRANCH (the empty asm keeps the compiler from turning the
                        asm volatile("");                            // block into 2 * V selects per row and token row — with them the kernel was
#pragma unroll                                                       // VALU-bound: 72 instead of 24 instructions per loaded row)
                        for (int j = 0; j < V; ++j) {
                            sum[k][j] += t[j];
                            mx[k][j] = fmaxf(mx[k][j], t[j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < TR; ++k) {
#pragma unroll
            for (int j = 0; j < V; j += 4)
                *(f32x4*)(csum + ((size_t)k * nitem + item) * V + j) = f32x4{sum[k][j], sum[k][j + 1], sum[k][j + 2], sum[k][j + 3]};
            cmax[(size_t)k * nitem + item] = pack16<DT>(mx[k]);
        }
    }
    __syncthreads();
    const float inv_area = 1.0f / (float)(kh * kw);
    const float w1 = g ? w1_1 : w1_0, w2 = g ? w2_1 : w2_0;
    const int N = th * tw, per = tw * nv;
    for (int o = threadIdx.x; o < ntr * per; o += blockDim.x) {
        const int k = o / per, o1 = o - k * per, ox = o1 / nv, v = o1 - ox * nv;
        float sum[V], mx[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { sum[j] = 0.0f; mx[j] = -INFINITY; }
        for (int dx = 0; dx < kw; ++dx) {
            const size_t it = (size_t)k * nitem + (size_t)(ox * sw + dx) * nv + v;
            float m[V];
            unpack16<DT>(cmax[it], m);
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                const f32x4 a = *(const f32x4*)(csum + it * V + j);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sum[j + e] += a[e]; mx[j + e] = fmaxf(mx[j + e], m[j + e]); }
            }
        }
        const int n = (oy0 + k) * tw + ox;
        const float* pos = (g ? pos1 : pos0) + (long long)n * C + v * V;
        float out[V];
#pragma unroll
        for (int j = 0; j < V; ++j) out[j] = (sum[j] * inv_area) * w1 + mx[j] * w2 + pos[j];
        *(u32x4*)(tok + (((long long)g * B + b) * N + n) * C + v * V) = pack16<DT>(out);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wavefront per token row, values held in registers, two-pass statistics
// ---------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void layernorm_kernel(const typename Elem<DT>::type* __restrict__ x, typename Elem<DT>::type* __restrict__ y,
                                                        const float* __restrict__ g0, const float* __restrict__ b0,
                                                        const float* __restrict__ g1, const float* __restrict__ b1,
                                                        long long rows_per_group, int C, int groups, float eps) {
    using E = Elem<DT>;
    constexpr int MAXV = 4;                                   // vectors per lane: C <= 64 * MAXV * VEC
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows_per_group * groups) return;
    const int grp = (int)(row / rows_per_group);
    const float* gam = grp ? g1 : g0;
    const float* bet = grp ? b1 : b0;
    const int nv = C / E::VEC;
    float val[MAXV][E::VEC];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = lane + 64 * i;
        if (v < nv) {
            unpack16<DT>(*(const u32x4*)(x + row * C + v * E::VEC), val[i]);
#pragma unroll
            for (int j = 0; j < E::VEC; ++j) s += val[i][j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = lane + 64 * i;
        if (v < nv) {
#pragma unroll
            for (int j = 0; j < E::VEC; ++j) { const float d = val[i][j] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = lane + 64 * i;
        if (v < nv) {
            float o[E::VEC];
#pragma unroll
            for (int j = 0; j < E::VEC; ++j) o[j] = (val[i][j] - mean) * rstd * gam[v * E::VEC + j] + bet[v * E::VEC + j];
            *(u32x4*)(y + row * C + v * E::VEC) = pack16<DT>(o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Cross attention (reference models/common.py:670-685)
//   out[dir] = softmax( Q_{1-dir} K_dir^T / sqrt(dk) ) V_dir          per (batch, head), dir = 0 (RGB) / 1 (IR)
//
// One workgroup = (q-split, head, batch*dir); its 4 wavefronts each own 32-query tiles.  N <= 400 tokens, so the whole
// K and V^T of the head live in LDS for the workgroup's lifetime (<= ~115 KB even for the fp32 build).
//   S^T = K Q^T     MFMA A operand = K rows from LDS (padded row stride, conflict-free ds_read_b128),
//                   B operand = the lane's own query row, loaded once from HBM into registers.
//                   -> every lane holds 16 of the 32 key scores of ONE query: the softmax row reduction is 15 in-lane
//                      max/adds plus one cross-half __shfl_xor(.., 32) — no LDS round trip, no serial lanes.
//   O^T = V^T P^T   the exponentiated scores feed the MFMA B operand straight from the accumulator registers (packed
//                   to bf16/f16 in-lane); V^T is stored in LDS with the key order permuted inside each 16-key group so
//                   that the A-operand fragment is again one 16-byte read.
// Online softmax (running max / sum, rescale of O per key tile) keeps registers independent of N.
// ---------------------------------------------------------------------------------------------------------------
template <int DT> __device__ __forceinline__ int vt_phys(int key) {
    if constexpr (DT == ICAF_F32) return key;
    else {
        const int k16 = key & 15;
        return (key & ~15) + (((k16 >> 2) & 1) << 3) + (k16 & 3) + ((k16 >> 3) << 2);
    }
}

template <int DT, int DKP>
__global__ __launch_bounds__(256) void cross_attn_kernel(const typename Elem<DT>::type* __restrict__ qkv,
                                                         typename Elem<DT>::type* __restrict__ out, int B, int N, int C, int DK,
                                                         int NP, float scale_l2e, int xq) {
    using E = Elem<DT>;
    using T = typename E::type;
    using AC = AttnCore<DT, DKP>;
    constexpr int VEC = E::VEC, EB = E::BYTES;
    constexpr int KSTEP = AC::KSTEP, QSTEPS = AC::QSTEPS, TD = AC::TD;
    constexpr int KS = DKP * EB + 16;           // K row stride (bytes), odd multiple of 16 -> conflict-free b128 reads
    const int VS = NP * EB + 16;                // V^T row stride (bytes)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;
    unsigned char* Vt = smem + (size_t)NP * KS;
    unsigned char* ones = Vt + (size_t)DKP * VS;                   // (AC::FREE only) one all-ones row: the softmax denominator's V^T row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    // XCD-aware placement (xq > 0: one-dimensional grid of xq * heads * 2B workgroups, xq = query splits, 2B % 8 == 0): all heads and query
    // splits of one (direction, image) run on ONE XCD.  A head's K / V rows are DK * EB-byte pieces of 3C-element rows — 64 bytes of a
    // 128-byte line at d_k = 32, the other half belonging to the neighbouring head — and every query split re-reads them: spread over
    // the XCDs in dispatch order, each line was pulled into several private L2s (PMC at P4: 51.9 MB fetched for 22 MB of distinct qkv).
    int h, qx, qsplit, grp;
    if (xq > 0) {
        qsplit = xq;
        const int heads = C / DK, per = qsplit * heads;
        const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
        const int gi = idx / per, rem = idx - gi * per;
        grp = gi * 8 + xcd;
        h = rem / qsplit;
        qx = rem - h * qsplit;
    } else {
        h = blockIdx.y; qx = blockIdx.x; qsplit = (int)gridDim.x; grp = blockIdx.z;
    }
    const int dir = grp / B, b = grp - dir * B;
    const long long row3 = 3LL * C;
    const T* kvbase = qkv + ((long long)(dir * B + b) * N) * row3 + (long long)h * DK;
    const T* qbase = qkv + ((long long)((1 - dir) * B + b) * N) * row3 + (long long)h * DK;

    // ---- stage K (row-major) and V^T (key-permuted) of this head into LDS ----------------------------------------
    constexpr int NVK = DKP / VEC;
    if constexpr (EB == 2) {
        // Round 5: FOUR consecutive keys per item.  vt_phys keeps the keys 4i .. 4i + 3 of a 16-key group adjacent (8 bytes of a V^T row), so the eight
        // d-rows of a 16-byte V vector take ONE 8-byte store each for four keys — 8 ds_write_b64 + 4 ds_write_b128 (K) per item where the key-by-key
        // loop issued 32 two-byte stores + 4: the staging was half of a workgroup's life (SQ wave-wait 0.48 in round 4).
        for (int idx = tid; idx < (NP >> 2) * NVK; idx += 256) {
            const int k4 = idx / NVK, v = idx - k4 * NVK, key0 = k4 * 4;
            u32x4 kq[4], vq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kq[i] = u32x4{0u, 0u, 0u, 0u};
                vq[i] = u32x4{0u, 0u, 0u, 0u};
                if (key0 + i < N && v * VEC < DK) {
                    kq[i] = *(const u32x4*)(kvbase + (key0 + i) * row3 + C + v * VEC);
                    vq[i] = *(const u32x4*)(kvbase + (key0 + i) * row3 + 2 * C + v * VEC);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(Ks + (size_t)(key0 + i) * KS + v * 16) = kq[i];
            const int pk = vt_phys<DT>(key0);                      // (= the physical position of the group's first key; the other three follow it)
#pragma unroll
            for (int j = 0; j < 8; ++j) {                          // d-row v * 8 + j: element j of each key's vector
                u32x2 w;
                if (j & 1) {
                    w[0] = (vq[0][j >> 1] >> 16) | (vq[1][j >> 1] & 0xffff0000u);
                    w[1] = (vq[2][j >> 1] >> 16) | (vq[3][j >> 1] & 0xffff0000u);
                } else {
                    w[0] = (vq[0][j >> 1] & 0xffffu) | (vq[1][j >> 1] << 16);
                    w[1] = (vq[2][j >> 1] & 0xffffu) | (vq[3][j >> 1] << 16);
                }
                *(u32x2*)(Vt + (size_t)(v * 8 + j) * VS + pk * 2) = w;
            }
        }
    } else
    for (int idx = tid; idx < NP * NVK; idx += 256) {
        const int key = idx / NVK, v = idx - key * NVK;
        u32x4 kvv = {0u, 0u, 0u, 0u}, vvv = {0u, 0u, 0u, 0u};
        if (key < N && v * VEC < DK) {
            kvv = *(const u32x4*)(kvbase + key * row3 + C + v * VEC);
            vvv = *(const u32x4*)(kvbase + key * row3 + 2 * C + v * VEC);
        }
        *(u32x4*)(Ks + (size_t)key * KS + v * 16) = kvv;
        const int pk = vt_phys<DT>(key);
#pragma unroll
        for (int j = 0; j < 4; ++j) *(unsigned int*)(Vt + (size_t)(v * 4 + j) * VS + pk * 4) = vvv[j];
    }
    if constexpr (AC::FREE) {
        const u32x4 of = ones_frag<DT>();
        for (int i = tid; i * 16 < NP * EB; i += 256) *(u32x4*)(ones + i * 16) = of;
    }
    __syncthreads();

    const int nqt = NP >> 5;
    for (int qt = qx + wave * qsplit; qt < nqt; qt += 4 * qsplit) {
        const int q = qt * 32 + l31;
        const bool qok = q < N;
        u32x4 qf[QSTEPS];
#pragma unroll
        for (int st = 0; st < QSTEPS; ++st) {
            const int off = st * KSTEP + hi * VEC;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (qok && off < DK) v = *(const u32x4*)(qbase + q * row3 + off);
            qf[st] = v;
        }
        f32x16 o[TD];
        float l;
        AC::template run<KS>(Ks, Vt, VS, ones, qf, nqt, N, scale_l2e, o, l);
        const float inv = 1.0f / l;
        if (qok) {
            T* orow = out + ((long long)(dir * B + b) * N + q) * C + (long long)h * DK;
#pragma unroll
            for (int td = 0; td < TD; ++td)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d0 = td * 32 + 8 * g4 + 4 * hi;
                    if (d0 < DK) {
                        const float v0 = o[td][4 * g4] * inv, v1 = o[td][4 * g4 + 1] * inv, v2 = o[td][4 * g4 + 2] * inv,
                                    v3 = o[td][4 * g4 + 3] * inv;
                        if constexpr (EB == 4) {
                            *(f32x4*)(orow + d0) = f32x4{v0, v1, v2, v3};
                        } else if constexpr (DT == ICAF_BF16) {
                            u32x2 pk;
                            pk[0] = pack2_bf16(v0, v1);
                            pk[1] = pack2_bf16(v2, v3);
                            *(u32x2*)(orow + d0) = pk;
                        } else {
                            u32x2 pk;
                            pk[0] = pack2_f16(v0, v1);
                            pk[1] = pack2_f16(v2, v3);
                            *(u32x2*)(orow + d0) = pk;
                        }
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// out[b][h][w][g*C + c] = bilinear(tokens[g])(h, w)[c] + fea_g[b][h][w][c]      (reference models/common.py:827-840)
// align_corners=False: src = max(0, (dst + 0.5) * in/out - 0.5), neighbours clamped to the last row / column.
// ---------------------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void upsample_merge_kernel(const typename Elem<DT>::type* __restrict__ tok,
                                                             const typename Elem<DT>::type* __restrict__ f0, int ld0,
                                                             const typename Elem<DT>::type* __restrict__ f1, int ld1,
                                                             typename Elem<DT>::type* __restrict__ out, int ldo, int B, int H, int W, int C,
                                                             int th, int tw) {
    using E = Elem<DT>;
    const int nv = C / E::VEC, N = th * tw;
    const long long total = (long long)B * H * W * 2 * nv;
    const float sy = (float)th / (float)H, sx = (float)tw / (float)W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(idx % nv);
        long long t = idx / nv;
        const int g = (int)(t & 1);
        t >>= 1;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H), b = (int)(t / H);
        float fy = ((float)h + 0.5f) * sy - 0.5f, fx = ((float)w + 0.5f) * sx - 0.5f;
        fy = fy < 0.0f ? 0.0f : fy;
        fx = fx < 0.0f ? 0.0f : fx;
        int y0 = (int)fy, x0 = (int)fx;
        y0 = y0 < th - 1 ? y0 : th - 1;
        x0 = x0 < tw - 1 ? x0 : tw - 1;
        const int y1 = y0 < th - 1 ? y0 + 1 : y0, x1 = x0 < tw - 1 ? x0 + 1 : x0;
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const typename E::type* tb = tok + (((long long)g * B + b) * N) * C + v * E::VEC;
        float a00[E::VEC], a01[E::VEC], a10[E::VEC], a11[E::VEC], r[E::VEC], o[E::VEC];
        unpack16<DT>(*(const u32x4*)(tb + (long long)(y0 * tw + x0) * C), a00);
        unpack16<DT>(*(const u32x4*)(tb + (long long)(y0 * tw + x1) * C), a01);
        unpack16<DT>(*(const u32x4*)(tb + (long long)(y1 * tw + x0) * C), a10);
        unpack16<DT>(*(const u32x4*)(tb + (long long)(y1 * tw + x1) * C), a11);
        const typename E::type* f = g ? f1 : f0;
        const int ld = g ? ld1 : ld0;
        const long long pix = ((long long)b * H + h) * W + w;
        unpack16<DT>(*(const u32x4*)(f + pix * ld + v * E::VEC), r);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) {
            const float top = a00[j] * (1.0f - lx) + a01[j] * lx;
            const float bot = a10[j] * (1.0f - lx) + a11[j] * lx;
            o[j] = (top * (1.0f - ly) + bot * ly) + r[j];
        }
        *(u32x4*)(out + pix * ldo + (long long)g * C + v * E::VEC) = pack16<DT>(o);
    }
}

template <int DT, int DKP>
static int launch_attn(const void* qkv, void* out, int B, int N, int C, int DK, int heads, hipStream_t s) {
    using T = typename Elem<DT>::type;
    constexpr int EB = Elem<DT>::BYTES;
    const int NP = (N + 31) & ~31;
    const size_t lds = (size_t)NP * (DKP * EB + 16) + (size_t)(DKP + (AttnCore<DT, DKP>::ones_row() ? 1 : 0)) * ((size_t)NP * EB + 16);
    if (lds > 160 * 1024) return fail(ICAF_ERR_UNSUPPORTED, "icaf_cross_attention: %zu bytes of LDS needed (N=%d, dk=%d) exceed 160 KiB", lds, N, DK);
    static std::atomic<bool> attr_set[ICAF_MAX_DEVICES];          // per instantiation and per device
    int dev = 0;
    ICAF_HIP(hipGetDevice(&dev));
    if (lds > 64 * 1024 && dev >= 0 && dev < ICAF_MAX_DEVICES && !attr_set[dev]) {
        ICAF_HIP(hipFuncSetAttribute((const void*)cross_attn_kernel<DT, DKP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    const int nqt = NP / 32;
    // Query splits per head: every split stages the head's K / V^T again (25 - 100 KB through scalar LDS transposition writes), so as FEW as
    // the chip needs: two workgroups per CU (512) must exist, beyond that one split — except at d_k <= 16 with many query tiles, where a wave
    // walking seven tiles alone is the longer pole (round 5, same box, batch 32: yolov5s P3 28.5 us with 4 splits, 25.5 with 2, 26.5 with 1;
    // yolov5l P3 44.3 -> 33.3 and P4 27.1 -> 22.1 with ONE split; P5 levels already ran one).  Small batches keep the splits: they are the
    // only parallelism there.
    const int qs_max = (nqt + 3) / 4;
    int qsplit = (DK <= 16 && nqt > 8) ? 2 : 1;
    const long long wgs1 = (long long)2 * B * heads;                  // workgroups with one split
    if (wgs1 * qsplit < 512) qsplit = (int)((512 + wgs1 - 1) / wgs1);
    if (qsplit > qs_max) qsplit = qs_max;
    {   // A/B switch (timing studies): icaf_set_option("attn_qsplit", n) forces n query splits per head (1 = a workgroup stages a head's K / V^T once for ALL its query tiles)
        const int forced = g_opt.attn_qsplit;
        if (forced > 0) qsplit = forced < nqt ? forced : nqt;
    }
    const float scale_l2e = (float)((1.0 / sqrt((double)DK)) * 1.4426950408889634);
    const bool remap = (2 * B) % 8 == 0 && heads * DK == C;          // whole (direction, image) groups per XCD
    dim3 grid((unsigned)qsplit, (unsigned)heads, (unsigned)(2 * B));
    if (remap) grid = dim3((unsigned)(qsplit * heads * 2 * B), 1u, 1u);
    hipLaunchKernelGGL((cross_attn_kernel<DT, DKP>), grid, dim3(256), lds, s, (const T*)qkv, (T*)out, B, N, C, DK, NP, scale_l2e, remap ? qsplit : 0);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <int DT>
static int dispatch_attn(const void* qkv, void* out, int B, int N, int C, int DK, int heads, hipStream_t s) {
    if (DK <= 16) return launch_attn<DT, 16>(qkv, out, B, N, C, DK, heads, s);
    if (DK <= 32) return launch_attn<DT, 32>(qkv, out, B, N, C, DK, heads, s);
    if (DK <= 48) return launch_attn<DT, 48>(qkv, out, B, N, C, DK, heads, s);
    if (DK <= 64) return launch_attn<DT, 64>(qkv, out, B, N, C, DK, heads, s);
    if (DK <= 96) return launch_attn<DT, 96>(qkv, out, B, N, C, DK, heads, s);
    if (DK <= 128) return launch_attn<DT, 128>(qkv, out, B, N, C, DK, heads, s);
    return fail(ICAF_ERR_UNSUPPORTED, "icaf_cross_attention: head dim %d > 128", DK);
}

}  // namespace icaf

using namespace icaf;

namespace {
template <int DT>
int run_pool_tokens(const void* f0, int ld0, const void* f1, int ld1, const float* p0, const float* p1, void* tok, int B, int H, int W, int C,
                    int th, int tw, int kh, int kw, int sh, int sw, float a0, float b0, float a1, float b1, hipStream_t s) {
    using T = typename Elem<DT>::type;
    constexpr int V = Elem<DT>::VEC;
    const size_t row_lds = (size_t)W * (C / V) * (V * sizeof(float) + 16);      // one token row: fp32 column sums + packed column maxima
    if ((kh > sh || kw > sw) && row_lds <= 160 * 1024) {             // overlapping windows: separable, TR token rows per workgroup
        // two token rows per workgroup when both fit the LDS, share input rows (kh > sh), are loaded in one batch and still leave a workgroup per CU
        const int tr = (2 * row_lds <= 160 * 1024 && kh > sh && kh + sh <= 12 && 2 * B * ((th + 1) / 2) >= 256) ? 2 : 1;
        const size_t rows_lds = tr * row_lds;
        static size_t attr_bytes[ICAF_MAX_DEVICES][3][2] = {};   // per device and instantiation
        int dev = 0;
        ICAF_HIP(hipGetDevice(&dev));
        const int nrow = kh + (tr - 1) * sh;
        const int rsel = nrow <= 4 ? 0 : nrow <= 8 ? 1 : 2;      // rows in flight per item: 4 / 8 / 12
        const void* fns[3][2] = {{(const void*)pool_tokens_rows_kernel<DT, 4, 1>, (const void*)pool_tokens_rows_kernel<DT, 4, 2>},
                                 {(const void*)pool_tokens_rows_kernel<DT, 8, 1>, (const void*)pool_tokens_rows_kernel<DT, 8, 2>},
                                 {(const void*)pool_tokens_rows_kernel<DT, 12, 1>, (const void*)pool_tokens_rows_kernel<DT, 12, 2>}};
        const void* fn = fns[rsel][tr - 1];
        if (rows_lds > 64 * 1024 && dev >= 0 && dev < ICAF_MAX_DEVICES && rows_lds > attr_bytes[dev][rsel][tr - 1]) {
            ICAF_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds));
            attr_bytes[dev][rsel][tr - 1] = rows_lds;
        }
        const dim3 grid((unsigned)(2 * B * ((th + tr - 1) / tr))), block(1024);
#define ICAF_POOL_ROWS(RR, TT)                                                                                                              \
    pool_tokens_rows_kernel<DT, RR, TT><<<grid, block, rows_lds, s>>>((const T*)f0, ld0, (const T*)f1, ld1, p0, p1, (T*)tok, B, H, W, C, th, tw, \
                                                                      kh, kw, sh, sw, a0, b0, a1, b1)
        if (rsel == 0) { if (tr == 2) ICAF_POOL_ROWS(4, 2); else ICAF_POOL_ROWS(4, 1); }
        else if (rsel == 1) { if (tr == 2) ICAF_POOL_ROWS(8, 2); else ICAF_POOL_ROWS(8, 1); }
        else { if (tr == 2) ICAF_POOL_ROWS(12, 2); else ICAF_POOL_ROWS(12, 1); }
#undef ICAF_POOL_ROWS
        ICAF_LAUNCH_CHECK();
        return ICAF_OK;
    }
    const long long total = 2LL * B * th * tw * (C / Elem<DT>::VEC);
    const PoolDiv dv{make_fastdiv((unsigned)(C / Elem<DT>::VEC)), make_fastdiv((unsigned)(th * tw)), make_fastdiv((unsigned)tw)};
    if (total < (1ll << 31))
        pool_tokens_kernel<DT, true><<<dim3(grid_for(total)), dim3(256), 0, s>>>((const T*)f0, ld0, (const T*)f1, ld1, p0, p1, (T*)tok, B, H, W, C,
                                                                                  th, tw, kh, kw, sh, sw, a0, b0, a1, b1, dv);
    else
        pool_tokens_kernel<DT, false><<<dim3(grid_for(total)), dim3(256), 0, s>>>((const T*)f0, ld0, (const T*)f1, ld1, p0, p1, (T*)tok, B, H, W, C,
                                                                                   th, tw, kh, kw, sh, sw, a0, b0, a1, b1, dv);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}
template <int DT>
int run_layernorm(const void* x, void* y, const float* g0, const float* b0, const float* g1, const float* b1, long long rpg, int C, int groups,
                  float eps, hipStream_t s) {
    using T = typename Elem<DT>::type;
    const long long rows = rpg * groups;
    layernorm_kernel<DT><<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>((const T*)x, (T*)y, g0, b0, g1, b1, rpg, C, groups, eps);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}
template <int DT>
int run_upsample_merge(const void* tok, const void* f0, int ld0, const void* f1, int ld1, void* out, int ldo, int B, int H, int W, int C, int th,
                       int tw, hipStream_t s) {
    using T = typename Elem<DT>::type;
    const long long total = 2LL * B * H * W * (C / Elem<DT>::VEC);
    upsample_merge_kernel<DT><<<dim3(grid_for(total)), dim3(256), 0, s>>>((const T*)tok, (const T*)f0, ld0, (const T*)f1, ld1, (T*)out, ldo, B,
                                                                           H, W, C, th, tw);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}
}  // namespace

#define DISPATCH_DT(dtype, FN, ...)                                  \
    switch (dtype) {                                                 \
        case ICAF_F32: return FN<ICAF_F32>(__VA_ARGS__);             \
        case ICAF_BF16: return FN<ICAF_BF16>(__VA_ARGS__);           \
        case ICAF_F16: return FN<ICAF_F16>(__VA_ARGS__);             \
        default: return fail(ICAF_ERR_ARG, "bad dtype %d", dtype);   \
    }

extern "C" int icaf_dmff_pool_tokens(const void* fea_rgb, int ld_rgb, const void* fea_ir, int ld_ir, const float* pos_rgb,
                                     const float* pos_ir, void* tokens, int dtype, int B, int H, int W, int C, int th, int tw, int kh,
                                     int kw, int sh, int sw, float w1_rgb, float w2_rgb, float w1_ir, float w2_ir, icaf_stream_t s) {
    if (!fea_rgb || !fea_ir || !pos_rgb || !pos_ir || !tokens) return fail(ICAF_ERR_ARG, "icaf_dmff_pool_tokens: null pointer");
    const int vec = dtype == ICAF_F32 ? 4 : 8;
    if (C % vec || ld_rgb % vec || ld_ir % vec) return fail(ICAF_ERR_ARG, "icaf_dmff_pool_tokens: C/ld must be multiples of %d", vec);
    if ((th - 1) * sh + kh > H || (tw - 1) * sw + kw > W || kh < 1 || kw < 1) return fail(ICAF_ERR_ARG, "icaf_dmff_pool_tokens: window exceeds the feature map");
    DISPATCH_DT(dtype, run_pool_tokens, fea_rgb, ld_rgb, fea_ir, ld_ir, pos_rgb, pos_ir, tokens, B, H, W, C, th, tw, kh, kw, sh, sw, w1_rgb,
                w2_rgb, w1_ir, w2_ir, S(s));
}

extern "C" int icaf_layernorm(const void* x, void* y, const float* gamma0, const float* beta0, const float* gamma1, const float* beta1,
                              int dtype, long long rows_per_group, int C, int groups, float eps, icaf_stream_t s) {
    if (!x || !y || !gamma0 || !beta0) return fail(ICAF_ERR_ARG, "icaf_layernorm: null pointer");
    if (groups < 1 || groups > 2 || (groups == 2 && (!gamma1 || !beta1))) return fail(ICAF_ERR_ARG, "icaf_layernorm: bad groups");
    const int vec = dtype == ICAF_F32 ? 4 : 8;
    if (C % vec || C > 64 * 4 * vec) return fail(ICAF_ERR_ARG, "icaf_layernorm: C=%d must be a multiple of %d and <= %d", C, vec, 64 * 4 * vec);
    DISPATCH_DT(dtype, run_layernorm, x, y, gamma0, beta0, gamma1 ? gamma1 : gamma0, beta1 ? beta1 : beta0, rows_per_group, C, groups, eps, S(s));
}

extern "C" int icaf_cross_attention(const void* qkv, void* out, int dtype, int B, int N, int C, int heads, icaf_stream_t s) {
    if (!qkv || !out) return fail(ICAF_ERR_ARG, "icaf_cross_attention: null pointer");
    if (heads < 1 || C % heads) return fail(ICAF_ERR_ARG, "icaf_cross_attention: C=%d not divisible by heads=%d", C, heads);
    const int DK = C / heads, vec = dtype == ICAF_F32 ? 4 : 8;
    if (DK % vec) return fail(ICAF_ERR_UNSUPPORTED, "icaf_cross_attention: head dim %d must be a multiple of %d for this dtype", DK, vec);
    if (B < 1 || N < 1 || 2LL * B > 65535) return fail(ICAF_ERR_ARG, "icaf_cross_attention: bad B/N");
    DISPATCH_DT(dtype, dispatch_attn, qkv, out, B, N, C, DK, heads, S(s));
}

extern "C" int icaf_dmff_upsample_merge(const void* tokens, const void* fea_rgb, int ld_rgb, const void* fea_ir, int ld_ir, void* out,
                                        int ldo, int dtype, int B, int H, int W, int C, int th, int tw, icaf_stream_t s) {
    if (!tokens || !fea_rgb || !fea_ir || !out) return fail(ICAF_ERR_ARG, "icaf_dmff_upsample_merge: null pointer");
    const int vec = dtype == ICAF_F32 ? 4 : 8;
    if (C % vec || ld_rgb % vec || ld_ir % vec || ldo % vec || ldo < 2 * C) return fail(ICAF_ERR_ARG, "icaf_dmff_upsample_merge: bad strides");
    DISPATCH_DT(dtype, run_upsample_merge, tokens, fea_rgb, ld_rgb, fea_ir, ld_ir, out, ldo, B, H, W, C, th, tw, S(s));
}
