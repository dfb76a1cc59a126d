// Fused DMFF block kernels for gfx950 (16-bit types) — one CrossTransformerBlock iteration (reference models/common.py:737-759,
// CrossAttention :641-687, MLP :704-709) in TWO launches instead of seven:
//
//   dmff_ln_qkv_kernel     LayerNorm (CrossAttention.LN1 on RGB tokens, LN2 on IR tokens, :661-662) as the prologue of the
//                          six Linear(C, C) projections (:664-669) run as one [64 x C] x [C x 3C] GEMM per workgroup.
//   dmff_attn_mlp_kernel   one workgroup = 64 token rows of one (image, modality):
//                            A. crossed attention of its rows, all heads (softmax(q_other k^T / sqrt(dk)) v, :670-681), K and
//                               V^T of two heads at a time in LDS, online softmax in registers (as cross_attn_kernel);
//                               the heads' outputs land in an LDS tile [64 x C] — never in HBM;
//                            B. out-projection + coefficient mix  x_att = c_res * x + c_acc * (att W_o^T + b)  (:682-685,:745-746);
//                            C. the block's shared LayerNorm LN2 over x_att (:749-750), in LDS;
//                            D. MLP  Linear(C,4C) -> GELU(erf) -> Linear(4C,C)  in 128-column hidden chunks: each chunk is
//                               produced into LDS and immediately contracted into the output accumulators, so the [64 x 4C]
//                               hidden tile never exists (:704-709);  x' = c_res2 * x_att + c_acc2 * (mlp + b)  (:751-752).
//
// All four GEMMs share one building block (gemm_pass): the 64-row operand tile is RESIDENT in LDS, weights stream from L2
// through a two-stage LDS ring in 128-byte K slices (global -> registers -> LDS, one LDS-only barrier per slice), MFMA
// 32x32x16 with the weights as the A operand so that a lane owns 4 consecutive channels of ONE token row — LayerNorm
// statistics, bias, GELU, the coefficient mixes and the residual are per-lane register work.
// Rounding points are those of the unfused launches (activations rounded to the storage type between layers, fp32 accumulate).
// An fp32 instantiation (v_mfma_f32_32x32x2_f32, erff GELU, fp32 tiles) exists for C <= 128 — where the LDS plan still fits — so
// that the SAME template the 16-bit bench path runs is checked against the reference's fp32 goldens at 1e-3
// (tests/test_gpu_dmff_fused.py); the fp32 model path uses the per-layer launches unless CrossTransformerBlock.fuse_fp32 is set.
#include "icaf_common.h"
#include "conv_common.h"
#include "attn_core.h"

namespace icaf {

struct DmffP {
    const void* x;            // tokens [2][rows][C]: LN + QKV input / residual of the attention mix
    void* qkv;                // [2][rows][3C]  (q | k | v per modality)
    void* y;                  // output tokens: element (g, row, c) at y + g * y_gs + row * ldy + c
    const void* wqkv; const float* bqkv;      // [2][Np][Kp], [2][Np]   rows = (que | key | val) output channels
    const void* wo; const float* bo;
    const void* w1; const float* b1;
    const void* w2; const float* b2;
    const float* ln_a_g[2]; const float* ln_a_b[2];     // CrossAttention.LN1 (RGB) / LN2 (IR)
    const float* ln_m_g; const float* ln_m_b;           // CrossTransformerBlock.LN2, shared by both modalities
    long long wqkv_gs, bqkv_gs, wo_gs, bo_gs, w1_gs, b1_gs, w2_gs, b2_gs, x_gs, y_gs;
    int B, N, C, heads, dk, Kp, Kp4, hid, ldy;
    float eps_a, eps_m, scale_l2e;
    float c_res_a[2], c_acc_a[2], c_res_m[2], c_acc_m[2];
    long long* dbg;           // optional: workgroup (0, 0, 0) records s_memtime at its phase boundaries (lab/probes/dmff_phases.py)
};

constexpr int FT = 256;              // threads per workgroup (4 wavefronts)
constexpr int TMROWS = 64;           // token rows per workgroup
// One ring stage = 128 output channels x SLB bytes of K, rows padded by 16 bytes (odd multiple of 16: conflict-free b128 reads).
template <int SLB> struct Ring {
    static constexpr int WROW = SLB + 16;
    static constexpr int STAGE = 128 * WROW;
    static constexpr int BYTES = 2 * STAGE;
};

// Weight-slice stream.  The GEMM passes of a phase form ONE continuous sequence of 128-channel x SLB-byte weight slices:
// slice s of a pass is consumed from LDS stage s & 1 while slice s + 1 sits in registers and slice s + 2 is being fetched
// (two slices of L2 latency hidden behind two slices of MFMAs); the last two steps of a pass fetch the first two slices of the
// NEXT pass, so only the first pass of a phase exposes a load.  Every pass has an even number of slices (SLB is chosen for
// that), so stage / register parity is a compile-time property of the code position — no dynamic register indexing.
template <int DT, int SLB> struct WS {
    using E = Elem<DT>;
    using T = typename E::type;
    using R = Ring<SLB>;
    static constexpr int BKE = SLB / E::BYTES;              // K elements per slice
    static constexpr int VPR = SLB / 16;                    // 16-byte vectors per row of a slice
    static constexpr int NV = 128 * VPR / FT;               // vectors per thread per slice
    static __device__ __forceinline__ void fetch(u32x4 (&r)[NV], const T* __restrict__ base, long long ldw) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = threadIdx.x + i * FT, row = v / VPR, kv = v % VPR;
            r[i] = *(const u32x4*)(base + (long long)row * ldw + kv * E::VEC);
        }
    }
    static __device__ __forceinline__ void commit(const u32x4 (&r)[NV], unsigned char* stage) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = threadIdx.x + i * FT, row = v / VPR, kv = v % VPR;
            *(u32x4*)(stage + row * R::WROW + kv * 16) = r[i];
        }
    }
    // acc[t] += W[wn*64 + t*32 + i][k] * A[wm*32 + j][k] over the SLB bytes of K of one slice
    static __device__ __forceinline__ void compute(f32x16 (&acc)[2], const unsigned char* arow, const unsigned char* wrow) {
#pragma unroll
        for (int ks = 0; ks < SLB / 32; ++ks) {
            const u32x4 xf = *(const u32x4*)(arow + ks * 32);
            const u32x4 w0 = *(const u32x4*)(wrow + ks * 32);
            const u32x4 w1 = *(const u32x4*)(wrow + 32 * R::WROW + ks * 32);
            mma_step<DT>(acc[0], w0, xf);
            mma_step<DT>(acc[1], w1, xf);
        }
    }
    // first pass of a phase: slice 0 -> stage 0 (visible after the barrier), slice 1 -> r1 (in flight)
    static __device__ __forceinline__ void start(u32x4 (&r0)[NV], u32x4 (&r1)[NV], const T* W, long long ldw, unsigned char* ring) {
        fetch(r0, W, ldw);
        fetch(r1, W + BKE, ldw);
        commit(r0, ring);
        lds_barrier();
    }
    // One pass over n (even) slices.  Pre: stage 0 holds slice 0, r1 holds slice 1.  Post: the same for the pass at Wn (if any).
    // The caller has made the A tile visible (barrier) before the call.
    static __device__ __forceinline__ void pass(f32x16 (&acc)[2], const unsigned char* A, int SA, int n, const T* __restrict__ W, long long ldw,
                                                const T* __restrict__ Wn, long long ldwn, unsigned char* ring, u32x4 (&r0)[NV], u32x4 (&r1)[NV]) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
        const int wm = wave & 1, wn = wave >> 1;
        const unsigned char* arow = A + (size_t)(wm * 32 + l31) * SA + hi * 16;
        const unsigned char* w0 = ring + (size_t)(wn * 64 + l31) * R::WROW + hi * 16;
        const unsigned char* w1 = w0 + R::STAGE;
        for (int c = 0; c < n; c += 2) {
            if (c + 2 < n) fetch(r0, W + (long long)(c + 2) * BKE, ldw);
            else if (Wn) fetch(r0, Wn, ldwn);
            compute(acc, arow + (size_t)c * SLB, w0);
            commit(r1, ring + R::STAGE);                       // slice c + 1
            lds_barrier();
            if (c + 3 < n) fetch(r1, W + (long long)(c + 3) * BKE, ldw);
            else if (Wn) fetch(r1, Wn + BKE, ldwn);
            compute(acc, arow + (size_t)(c + 1) * SLB, w1);
            if (c + 2 < n || Wn) commit(r0, ring);             // slice c + 2, or slice 0 of the next pass
            lds_barrier();
        }
    }
};

// Weight-slice stream by LDS-DMA, for the wide levels (C >= 256, 16-bit types).  At C = 256 / 512 the register-staged stream above
// is a chain of exposed L2 round trips (phase clocks, batch 32: 1850 cycles per 16 KB slice against 256 cycles of MFMAs — the MLP
// alone 118 k / 461 k cycles at P4 / P5).  Here the slices of the WHOLE out-projection + MLP sequence of a workgroup form one stream
// that `buffer_load ... lds` writes straight into an NS-stage ring, NS - 1 slices ahead of the MFMAs and across pass and phase
// boundaries (the producer cursor walks the passes in the order the kernel consumes them), synchronised like the conv kernels:
// a counted `s_waitcnt vmcnt` (a wave's own 4 instructions per slice, in issue order) + one LDS-only barrier per slice.  Other
// vector-memory operations issued in between (bias / residual loads, parked rows) are younger than the slice a wait targets or
// have been waited for by the compiler — they can only make a counted wait stricter.  Rows are 128 bytes, their 16-byte slots
// XOR-swizzled by (row >> 1) & 7 on the global side (igemm's layout: conflict-free b128 fragment reads without padding).
template <int DT, int NS, int NT = FT> struct WSD {
    using E = Elem<DT>;
    using T = typename E::type;
    static constexpr int SLB = 128, BKE = SLB / E::BYTES, STAGE = 128 * SLB, BYTES = NS * STAGE;
    static constexpr int PER = 128 / 8 / (NT / 64);            // DMA instructions per wave per slice (4; 2 with eight wavefronts)
    struct State {
        const T *Wo, *W1, *W2;                                  // the program: out-projection passes, then per hidden chunk fc1 + fc2 passes
        int Kp, Kp4, nsl, npass, nchunk;
        const T* base; long long ld; int left, seg;             // producer cursor: `left` slices of segment `seg` remain, next one at `base`
        int fill, cons, inflight;                               // ring: next stage to fill / to consume, slices issued but not consumed
    };
    static __device__ __forceinline__ void segment(State& st) {
        const int s = st.seg;
        if (s < st.npass) { st.base = st.Wo + (long long)s * 128 * st.Kp; st.ld = st.Kp; st.left = st.nsl; return; }
        const int m = s - st.npass, chunk = m / (1 + st.npass), r = m - chunk * (1 + st.npass);
        if (chunk >= st.nchunk) { st.left = 0; return; }
        if (r == 0) { st.base = st.W1 + (long long)chunk * 128 * st.Kp; st.ld = st.Kp; st.left = st.nsl; }
        else { st.base = st.W2 + (long long)(r - 1) * 128 * st.Kp4 + chunk * 128; st.ld = st.Kp4; st.left = 128 / BKE; }
    }
    static __device__ __forceinline__ void issue(State& st, unsigned char* ring) {
        if (st.left == 0) return;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)st.base, 0, 0x7fffffff, 0x00020000);
        unsigned char* stage = ring + st.fill * STAGE;
        const int rsub = lane >> 3;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = wave + (NT / 64) * i;
            const int sl = (lane & 7) ^ (((j & 1) << 2) | (rsub >> 1));
            const unsigned voff = (unsigned)(((long long)(j * 8 + rsub) * st.ld + sl * E::VEC) * E::BYTES);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(stage + (j * 8) * SLB), 16, voff, 0, 0, 0);
        }
        st.base += BKE;
        if (--st.left == 0) { ++st.seg; segment(st); }
        st.fill = st.fill + 1 == NS ? 0 : st.fill + 1;
        ++st.inflight;
    }
    static __device__ __forceinline__ void begin(State& st, const DmffP& p, int dir, unsigned char* ring) {
        st.Wo = (const T*)p.wo + dir * p.wo_gs; st.W1 = (const T*)p.w1 + dir * p.w1_gs; st.W2 = (const T*)p.w2 + dir * p.w2_gs;
        st.Kp = p.Kp; st.Kp4 = p.Kp4; st.nsl = p.C / BKE; st.npass = (p.C + 127) / 128; st.nchunk = p.hid / 128;
        st.seg = 0; st.fill = 0; st.cons = 0; st.inflight = 0;
        segment(st);
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue(st, ring);
    }
    // acc[t] += W[wn*64 + t*32 + i][k] * A[wm*32 + j][k] over the next n slices of the stream (the pass the program has reached)
    static __device__ __forceinline__ void pass(f32x16 (&acc)[2], const unsigned char* A, int SA, int n, unsigned char* ring, State& st) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
        const int wm = wave & 1, wn = wave >> 1;
        const unsigned char* arow = A + (size_t)(wm * 32 + l31) * SA + hi * 16;
        const int fkey = (l31 >> 1) & 7;
        const int wbase = (wn * 64 + l31) * SLB;
        for (int c = 0; c < n; ++c) {
            if (st.inflight - 1 >= NS - 2) wait_vmcnt<PER * (NS - 2)>();      // the slice to consume has landed (this wave's part of it)
            else wait_vmcnt<0>();
            lds_barrier();                                                    // ... everyone's part; and the stage consumed last is free
            issue(st, ring);
            const unsigned char* w = ring + st.cons * STAGE + wbase;
            const unsigned char* a = arow + (size_t)c * SLB;
#pragma unroll
            for (int ks = 0; ks < SLB / 32; ++ks) {
                const int off = ((2 * ks + hi) ^ fkey) << 4;
                const u32x4 xf = *(const u32x4*)(a + ks * 32);
                const u32x4 w0 = *(const u32x4*)(w + off);
                const u32x4 w1 = *(const u32x4*)(w + 32 * SLB + off);
                mma_step<DT>(acc[0], w0, xf);
                mma_step<DT>(acc[1], w1, xf);
            }
            st.cons = st.cons + 1 == NS ? 0 : st.cons + 1;
            --st.inflight;
        }
    }
};

// LayerNorm of the 64 rows of an LDS tile (row stride S bytes, C channels), in place or into `dst`: 4 threads per row,
// two-pass statistics in fp32 on the stored (16-bit) values, exactly the arithmetic of layernorm_kernel (dmff.hip).
template <int DT>
__device__ __forceinline__ void tile_layernorm(const unsigned char* src, unsigned char* dst, int S, int C, const float* __restrict__ gam,
                                               const float* __restrict__ bet, float eps) {
    using E = Elem<DT>;
    const int tid = threadIdx.x, row = tid >> 2, part = tid & 3;
    const int nv = C / E::VEC;                           // 16-byte vectors per row; vectors part, part+4, ... belong to this thread
    const unsigned char* r = src + (size_t)row * S;
    float s = 0.0f;
    for (int v = part; v < nv; v += 4) {
        float t[E::VEC];
        unpack16<DT>(*(const u32x4*)(r + v * 16), t);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) s += t[j];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mean = s / (float)C;
    float q = 0.0f;
    for (int v = part; v < nv; v += 4) {
        float t[E::VEC];
        unpack16<DT>(*(const u32x4*)(r + v * 16), t);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) { const float d = t[j] - mean; q += d * d; }
    }
    q += __shfl_xor(q, 1);
    q += __shfl_xor(q, 2);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    unsigned char* w = dst + (size_t)row * S;
    for (int v = part; v < nv; v += 4) {
        float t[E::VEC], o[E::VEC];
        unpack16<DT>(*(const u32x4*)(r + v * 16), t);
#pragma unroll
        for (int j = 0; j < E::VEC; ++j) o[j] = (t[j] - mean) * rstd * gam[v * E::VEC + j] + bet[v * E::VEC + j];
        *(u32x4*)(w + v * 16) = pack16<DT>(o);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm + QKV projection.  grid = (ceil(rows / 64), column groups of 384 output channels, 2 modalities)
// ---------------------------------------------------------------------------------------------------------------
constexpr int QKV_GROUP = 384;       // output channels per workgroup (3 passes of 128): C / 128 groups cover the 3C outputs

template <int DT, int SLB>
__global__ __launch_bounds__(FT) void dmff_ln_qkv_kernel(const DmffP p) {
    using E = Elem<DT>;
    using T = typename E::type;
    using S = WS<DT, SLB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = p.C, SA = C * E::BYTES + 16;
    unsigned char* tile = smem;                                  // [64][SA]
    unsigned char* ring = smem + (size_t)TMROWS * SA;            // Ring<SLB>::BYTES
    const int g = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const long long rows = (long long)p.B * p.N, r0 = (long long)blockIdx.x * TMROWS;
    const int nout = 3 * C, nbeg = blockIdx.y * QKV_GROUP, nend = nbeg + QKV_GROUP < nout ? nbeg + QKV_GROUP : nout;
    const T* W = (const T*)p.wqkv + g * p.wqkv_gs;
    u32x4 r0v[S::NV], r1v[S::NV];
    S::fetch(r0v, W + (long long)nbeg * p.Kp, p.Kp);             // the first weight slices travel while the tokens are normalised
    S::fetch(r1v, W + (long long)nbeg * p.Kp + S::BKE, p.Kp);
    const T* xg = (const T*)p.x + g * p.x_gs;
    const int nv = C / E::VEC;
    for (int idx = tid; idx < TMROWS * nv; idx += FT) {          // raw tokens -> LDS (rows beyond the tensor: clamped, never stored)
        const int row = idx / nv, v = idx - row * nv;
        long long r = r0 + row;
        r = r < rows ? r : rows - 1;
        *(u32x4*)(tile + (size_t)row * SA + v * 16) = *(const u32x4*)(xg + r * C + v * E::VEC);
    }
    S::commit(r0v, ring);
    __syncthreads();
    tile_layernorm<DT>(tile, tile, SA, C, p.ln_a_g[g], p.ln_a_b[g], p.eps_a);
    __syncthreads();
    const float* bias = p.bqkv + g * p.bqkv_gs;
    T* out = (T*)p.qkv + (long long)g * rows * 3 * C;
    const long long row = r0 + wm * 32 + l31;
    const int nsl = C / S::BKE;
    for (int n0 = nbeg; n0 < nend; n0 += 128) {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        const T* Wn = n0 + 128 < nend ? W + (long long)(n0 + 128) * p.Kp : nullptr;
        S::pass(acc, tile, SA, nsl, W + (long long)n0 * p.Kp, p.Kp, Wn, p.Kp, ring, r0v, r1v);
        if (row < rows) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * 64 + t * 32 + 8 * q + 4 * hi;
                    if (n < nout) {
                        const f32x4 b = *(const f32x4*)(bias + n);
                        *(typename Quad<DT>::type*)(out + row * nout + n) = pack4<DT>(acc[t][4 * q] + b[0], acc[t][4 * q + 1] + b[1], acc[t][4 * q + 2] + b[2],
                                                                                      acc[t][4 * q + 3] + b[3]);
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// attention + out-projection + LayerNorm + MLP.  grid = (ceil(N / 64), B, 2 directions)
// ---------------------------------------------------------------------------------------------------------------
// (fp32: identity key order and P taken register-for-register, as cross_attn_kernel's fp32 build — dmff.hip)
template <int DT> __device__ __forceinline__ int vt_phys16(int key) {
    if constexpr (DT == ICAF_F32) return key;
    else {
        const int k16 = key & 15;
        return (key & ~15) + (((k16 >> 2) & 1) << 3) + (k16 & 3) + ((k16 >> 3) << 2);
    }
}
// one 16-byte vector of a V row (channels v * VEC ...) -> VEC rows of V^T at (permuted) key column pk
template <int DT> __device__ __forceinline__ void scatter_vt(unsigned char* Vt, int VS, int v, int pk, const u32x4& vvv) {
    if constexpr (DT == ICAF_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(unsigned int*)(Vt + (size_t)(v * 4 + j) * VS + pk * 4) = vvv[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            *(unsigned short*)(Vt + (size_t)(v * 8 + j) * VS + pk * 2) = (unsigned short)((vvv[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
    }
}

constexpr int DMFF_NSD = 4;          // stages of the DMA weight ring (16 KB each)
template <int DT, int NP2, int SLB> constexpr bool attn_mlp_dma() { return NP2 >= 2 && DT != ICAF_F32 && SLB == 128; }

template <int DT, int DKP, int NP2, int SLB>
__global__ __launch_bounds__(FT, (NP2 == 1 && DKP <= 32 && DT != ICAF_F32) ? 2 : 1) void dmff_attn_mlp_kernel(const DmffP p) {      // C <= 128: two workgroups per CU (<= 256 registers)
    using E = Elem<DT>;
    using T = typename E::type;
    using S = WS<DT, SLB>;
    constexpr bool DMA = attn_mlp_dma<DT, NP2, SLB>();     // C >= 256: weight slices by LDS-DMA, DMFF_NSD-stage ring (WSD)
    using D = WSD<DT == ICAF_F32 ? ICAF_BF16 : DT, DMFF_NSD>;
    constexpr int RING_BYTES = DMA ? D::BYTES : Ring<SLB>::BYTES;
    using AC = AttnCore<DT, DKP>;                          // the attention inner loop shared with cross_attn_kernel (attn_core.h)
    constexpr int VEC = E::VEC, EB = E::BYTES;
    constexpr int KSTEP = AC::KSTEP, QSTEPS = AC::QSTEPS, TD = AC::TD;
    // K row stride: 32-byte rows (dk <= 16) are read as ONE contiguous kilobyte per b128 wave read — no padding needed, and the
    // 13 KB it saves at N = 400 lets two workgroups share a CU (2 waves / SIMD); wider rows keep the odd-multiple-of-16 stride
    constexpr int KS = DKP * EB == 32 ? 32 : DKP * EB + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = p.C, N = p.N, DK = p.dk, NP = (N + 31) & ~31;
    const int SA = C * EB + 16, VS = NP * EB + 16;
    const size_t tile_bytes = (size_t)TMROWS * SA;
    constexpr int SH = 128 * EB + 16;
    const size_t hb_bytes = (size_t)TMROWS * SH;
    unsigned char* T0 = smem;                        // attention output -> later the LayerNorm'ed MLP input
    unsigned char* U = smem + tile_bytes;            // union: {K, V^T of two heads}  |  {H = hidden chunk, weight ring, LayerNorm partial sums}
    const size_t kv_head = (size_t)NP * KS + (size_t)(DKP + (AC::ones_row() ? 1 : 0)) * VS;      // (+ the all-ones V^T row: softmax denominator by MFMA)
    unsigned char* Hb = U;
    unsigned char* ring = U + hb_bytes;
    float* red = (float*)(ring + RING_BYTES);        // [2][64] row partial sums of the two column halves

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    // Workgroup -> (64-row tile, image, direction).  The tiles of one (image, direction) pair all stage the same K / V^T, and the two
    // directions of an image read each other's rows as queries: they are placed on ONE XCD (hardware deals consecutive workgroup ids
    // round-robin over the 8 XCDs), so that those re-reads hit its L2 instead of each XCD fetching its own copy through the fabric
    // (PMC before: 216 MB fetched per launch at yolov5s P3 / batch 32 for ~30 MB of distinct qkv + token bytes).
    const int tiles = (N + TMROWS - 1) / TMROWS, pairs = 2 * p.B, bid = blockIdx.x;
    int pair, tile;
    if ((pairs & 7) == 0) { const int xcd = bid & 7, slot = bid >> 3; pair = xcd + 8 * (slot / tiles); tile = slot - (slot / tiles) * tiles; }
    else { pair = bid / tiles; tile = bid - pair * tiles; }
    const int dir = pair / p.B, b = pair - dir * p.B, q0 = tile * TMROWS;
    const long long rows = (long long)p.B * N, row3 = 3LL * C;
    const T* qkv = (const T*)p.qkv;
    const T* kvb = qkv + ((long long)dir * rows + (long long)b * N) * row3;
    const T* qb = qkv + ((long long)(1 - dir) * rows + (long long)b * N) * row3;

#define DMFF_STAMP(i) do { if (p.dbg && blockIdx.x == 0 && tid == 0) p.dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    DMFF_STAMP(0);
    // ---- A. attention: two heads per round (wave pair = head, wave parity = 32-query tile).  K and V of the NEXT round travel
    //         from L2 into registers while this round computes (its staging loads were the longest serial chain of the kernel:
    //         ~7 dependent L2 round trips per round); V^T is transposed by the 2-byte LDS writes. -----------------------------
    {
        const int hsel = wave >> 1, qt = wave & 1;
        const int q = q0 + qt * 32 + l31;
        const bool qok = q < N;
        const int nkt = NP >> 5;
        constexpr int NVK = DKP / VEC;
        constexpr int MAXI = 4;                               // (key, vector) items per thread and head held in registers
        const int items = NP * NVK;
        const bool pre = items <= MAXI * FT;                  // otherwise (N * dk > 8192): plain staging loop, no prefetch
        u32x4 kreg[2][MAXI], vreg[2][MAXI];
        auto load_round = [&](int h0) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const T* base = kvb + (long long)(h0 + hh) * DK;
#pragma unroll
                for (int i = 0; i < MAXI; ++i) {
                    const int idx = tid + i * FT, key = idx / NVK, v = idx - key * NVK;
                    const bool ok = idx < items && h0 + hh < p.heads && key < N && v * VEC < DK;
                    const T* src = base + (long long)(ok ? key : 0) * row3 + (ok ? v * VEC : 0);
                    const u32x4 a = *(const u32x4*)(src + C), c2 = *(const u32x4*)(src + 2 * C);     // unconditional loads from a clamped address
                    kreg[hh][i] = ok ? a : u32x4{0u, 0u, 0u, 0u};
                    vreg[hh][i] = ok ? c2 : u32x4{0u, 0u, 0u, 0u};
                }
            }
        };
        auto store_round = [&]() {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                unsigned char* Ks = U + hh * kv_head;
                unsigned char* Vt = Ks + (size_t)NP * KS;
#pragma unroll
                for (int i = 0; i < MAXI; ++i) {
                    const int idx = tid + i * FT, key = idx / NVK, v = idx - key * NVK;
                    if (idx < items) {
                        *(u32x4*)(Ks + (size_t)key * KS + v * 16) = kreg[hh][i];
                        scatter_vt<DT>(Vt, VS, v, vt_phys16<DT>(key), vreg[hh][i]);
                    }
                }
            }
        };
        if (pre) { load_round(0); store_round(); }
        if constexpr (AC::ones_row()) {                       // written once: the staging of later rounds never touches it
            const u32x4 of = ones_frag<DT>();
            for (int hh = 0; hh < 2; ++hh) {
                unsigned char* orow = U + hh * kv_head + (size_t)NP * KS + (size_t)DKP * VS;
                for (int i = tid; i * 16 < NP * EB; i += FT) *(u32x4*)(orow + i * 16) = of;
            }
        }
        for (int h0 = 0; h0 < p.heads; h0 += 2) {
            if (!pre) {
                if (h0) __syncthreads();                   // previous round's K / V^T are free again
                for (int hh = 0; hh < 2; ++hh) {
                    if (h0 + hh >= p.heads) break;
                    unsigned char* Ks = U + hh * kv_head;
                    unsigned char* Vt = Ks + (size_t)NP * KS;
                    const T* base = kvb + (long long)(h0 + hh) * DK;
                    for (int idx = tid; idx < items; idx += FT) {
                        const int key = idx / NVK, v = idx - key * NVK;
                        u32x4 kvv = {0u, 0u, 0u, 0u}, vvv = {0u, 0u, 0u, 0u};
                        if (key < N && v * VEC < DK) {
                            kvv = *(const u32x4*)(base + key * row3 + C + v * VEC);
                            vvv = *(const u32x4*)(base + key * row3 + 2 * C + v * VEC);
                        }
                        *(u32x4*)(Ks + (size_t)key * KS + v * 16) = kvv;
                        scatter_vt<DT>(Vt, VS, v, vt_phys16<DT>(key), vvv);
                    }
                }
            }
            __syncthreads();
            if (h0 == 0) DMFF_STAMP(1);
            if (pre && h0 + 2 < p.heads) load_round(h0 + 2);
            const int h = h0 + hsel;
            if (h < p.heads) {
                const unsigned char* Ks = U + hsel * kv_head;
                const unsigned char* Vt = Ks + (size_t)NP * KS;
                u32x4 qf[QSTEPS];
#pragma unroll
                for (int st = 0; st < QSTEPS; ++st) {
                    const int off = st * KSTEP + hi * VEC;
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (qok && off < DK) v = *(const u32x4*)(qb + (long long)q * row3 + (long long)h * DK + off);
                    qf[st] = v;
                }
                f32x16 o[TD];
                float l;
                AC::template run<KS>(Ks, Vt, VS, Vt + (size_t)DKP * VS, qf, nkt, N, p.scale_l2e, o, l);
                const float inv = qok ? 1.0f / l : 0.0f;
                unsigned char* orow = T0 + (size_t)(qt * 32 + l31) * SA + (size_t)h * DK * EB;
#pragma unroll
                for (int td = 0; td < TD; ++td)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int d0 = td * 32 + 8 * g4 + 4 * hi;
                        if (d0 < DK)
                            *(typename Quad<DT>::type*)(orow + d0 * EB) = pack4<DT>(o[td][4 * g4] * inv, o[td][4 * g4 + 1] * inv, o[td][4 * g4 + 2] * inv,
                                                                   o[td][4 * g4 + 3] * inv);
                    }
            }
            if (pre && h0 + 2 < p.heads) {
                __syncthreads();                           // every wave is done with this round's K / V^T
                store_round();
            }
        }
    }
    __syncthreads();              // T0 = attention output of the 64 rows; the K / V^T region is free
    DMFF_STAMP(2);

    // ---- B. out-projection + coefficient mix: x_att = c_res * x + c_acc * (att W_o^T + b), rounded to the storage type (as the
    //         per-layer launch stores it) and kept in REGISTERS, in the accumulator layout of the 128-channel passes --------------
    const int wm = wave & 1, wn = wave >> 1;
    const int lrow = wm * 32 + l31;                        // local token row of this lane
    const int tok = q0 + lrow;
    const bool rok = tok < N;
    const long long grow = (long long)b * N + (rok ? tok : N - 1);      // clamped global row (loads only)
    const T* xres = (const T*)p.x + dir * p.x_gs + grow * C;
    const T* Wo = (const T*)p.wo + dir * p.wo_gs;
    const T* W1 = (const T*)p.w1 + dir * p.w1_gs;
    const T* W2 = (const T*)p.w2 + dir * p.w2_gs;
    const int nsl = C / S::BKE, nsl2 = 128 / S::BKE, npass = (C + 127) / 128;
    T* yrow = (T*)p.y + dir * p.y_gs + ((long long)b * N + (rok ? tok : 0)) * p.ldy;
    u32x4 r0v[DMA ? 1 : S::NV], r1v[DMA ? 1 : S::NV];
    typename D::State dst;
    if constexpr (DMA) D::begin(dst, p, dir, ring);
    else S::start(r0v, r1v, Wo, p.Kp, ring);
    typename Quad<DT>::type xatt[NP2][2][4];
    {
        const float* bias = p.bo + dir * p.bo_gs;
        const float ca = p.c_acc_a[dir], cr = p.c_res_a[dir];
#pragma unroll
        for (int i = 0; i < NP2; ++i) {
            if (i < npass) {
                f32x16 acc[2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
                const T* Wn = i + 1 < npass ? Wo + (long long)(i + 1) * 128 * p.Kp : W1;       // then the first MLP chunk
                if constexpr (DMA) D::pass(acc, T0, SA, nsl, ring, dst);
                else S::pass(acc, T0, SA, nsl, Wo + (long long)i * 128 * p.Kp, p.Kp, Wn, p.Kp, ring, r0v, r1v);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = i * 128 + wn * 64 + t * 32 + 8 * q + 4 * hi;
                        typename Quad<DT>::type pk = pack4<DT>(0.f, 0.f, 0.f, 0.f);
                        if (n < C) {
                            const f32x4 bv = *(const f32x4*)(bias + n);
                            float rv[4];
                            unpack4<DT>(*(const typename Quad<DT>::type*)(xres + n), rv);
                            float v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(cr, rv[j], (acc[t][4 * q + j] + bv[j]) * ca);
                            pk = pack4<DT>(v[0], v[1], v[2], v[3]);
                        }
                        xatt[i][t][q] = pk;
                    }
            }
        }
    }
    DMFF_STAMP(3);
    // ---- C. the block's shared LayerNorm over x_att, from registers: row sums = this lane's channels + the other lane half
    //         (shuffle) + the other column half (two floats per row through LDS); two-pass statistics on the rounded values.
    //         The normalised tile overwrites T0 (every out-projection pass has finished reading it: each ends with a barrier). -----
    {
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < NP2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
                    unpack4<DT>(xatt[i][t][q], v);                     // (channels >= C hold zeros)
                    sum += (v[0] + v[1]) + (v[2] + v[3]);
                }
        sum += __shfl_xor(sum, 32);
        if (hi == 0) red[wn * 64 + lrow] = sum;
        lds_barrier();
        const float mean = (red[lrow] + red[64 + lrow]) / (float)C;
        lds_barrier();
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < NP2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = i * 128 + wn * 64 + t * 32 + 8 * q + 4 * hi;
                    if (n < C) {
                        float v[4];
                        unpack4<DT>(xatt[i][t][q], v);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float d = v[j] - mean; sq += d * d; }
                    }
                }
        sq += __shfl_xor(sq, 32);
        if (hi == 0) red[wn * 64 + lrow] = sq;
        lds_barrier();
        const float rstd = 1.0f / sqrtf((red[lrow] + red[64 + lrow]) / (float)C + p.eps_m);
#pragma unroll
        for (int i = 0; i < NP2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = i * 128 + wn * 64 + t * 32 + 8 * q + 4 * hi;
                    if (n < C) {
                        const f32x4 gv = *(const f32x4*)(p.ln_m_g + n), bv = *(const f32x4*)(p.ln_m_b + n);
                        float v[4], o[4];
                        unpack4<DT>(xatt[i][t][q], v);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = (v[j] - mean) * rstd * gv[j] + bv[j];
                        *(typename Quad<DT>::type*)(T0 + (size_t)lrow * SA + n * EB) = pack4<DT>(o[0], o[1], o[2], o[3]);
                        // x_att is needed once more, as the residual of the MLP mix: park it in this workgroup's own rows of the
                        // output tensor (read back by the same lane at the end) instead of holding 16 * NP2 registers through the MLP
                        if (rok) *(typename Quad<DT>::type*)(yrow + n) = xatt[i][t][q];
                    }
                }
        lds_barrier();
    }
    DMFF_STAMP(4);
    // ---- D. MLP in 128-column hidden chunks: H = GELU(n2 W1_chunk^T + b1) -> LDS, then acc2 += H W2[:, chunk]^T ------------------
    f32x16 acc2[NP2][2];
#pragma unroll
    for (int i = 0; i < NP2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][t][r] = 0.0f;
    {
        const float* b1 = p.b1 + dir * p.b1_gs;
        for (int hc = 0; hc < p.hid; hc += 128) {
            f32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
            if constexpr (DMA) D::pass(acc, T0, SA, nsl, ring, dst);
            else S::pass(acc, T0, SA, nsl, W1 + (long long)hc * p.Kp, p.Kp, W2 + hc, p.Kp4, ring, r0v, r1v);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = wn * 64 + t * 32 + 8 * q + 4 * hi;          // column inside the chunk
                    const f32x4 bv = *(const f32x4*)(b1 + hc + nl);
                    *(typename Quad<DT>::type*)(Hb + (size_t)lrow * SH + nl * EB) =      // (fp32: erff; 16-bit: the 1.5e-7 polynomial — apply_act)
                        pack4<DT>(apply_act<ICAF_ACT_GELU, DT>(acc[t][4 * q] + bv[0]), apply_act<ICAF_ACT_GELU, DT>(acc[t][4 * q + 1] + bv[1]),
                                  apply_act<ICAF_ACT_GELU, DT>(acc[t][4 * q + 2] + bv[2]), apply_act<ICAF_ACT_GELU, DT>(acc[t][4 * q + 3] + bv[3]));
                }
            lds_barrier();
#pragma unroll
            for (int i = 0; i < NP2; ++i)
                if (i < npass) {
                    const T* Wn = i + 1 < npass ? W2 + (long long)(i + 1) * 128 * p.Kp4 + hc
                                                : (hc + 128 < p.hid ? W1 + (long long)(hc + 128) * p.Kp : nullptr);
                    const long long ldn = i + 1 < npass ? p.Kp4 : p.Kp;
                    if constexpr (DMA) D::pass(acc2[i], Hb, SH, nsl2, ring, dst);
                    else S::pass(acc2[i], Hb, SH, nsl2, W2 + (long long)i * 128 * p.Kp4 + hc, p.Kp4, Wn, ldn, ring, r0v, r1v);
                }
            // (all waves are done with Hb before the next chunk overwrites it: every register-staged pass ends with a barrier, and
            //  every step of the DMA stream's next fc1 pass begins with one)
        }
    }
    DMFF_STAMP(5);
    // ---- output: x' = c_res2 * x_att + c_acc2 * (mlp + b2) ----------------------------------------------------------
    if (rok) {
        const float* b2 = p.b2 + dir * p.b2_gs;
        const float ca = p.c_acc_m[dir], cr = p.c_res_m[dir];
#pragma unroll
        for (int i = 0; i < NP2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = i * 128 + wn * 64 + t * 32 + 8 * q + 4 * hi;
                    if (n < C) {
                        const f32x4 bv = *(const f32x4*)(b2 + n);
                        float rv[4];
                        unpack4<DT>(*(const typename Quad<DT>::type*)(yrow + n), rv);   // x_att, parked here after the LayerNorm
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = __builtin_fmaf(cr, rv[j], (acc2[i][t][4 * q + j] + bv[j]) * ca);
                        *(typename Quad<DT>::type*)(yrow + n) = pack4<DT>(v[0], v[1], v[2], v[3]);
                    }
                }
    }
    DMFF_STAMP(6);
}

static inline int slice_bytes(int C) { return C % 128 == 0 ? 128 : 64; }     // every pass needs an even number of slices

static size_t attn_mlp_lds(int C, int N, int dkp, int eb) {      // (eb = 4: always 128-byte slices)
    const int NP = (N + 31) & ~31;
    const size_t tile = (size_t)TMROWS * (C * eb + 16), hb = (size_t)TMROWS * (128 * eb + 16);
    const size_t ks = dkp * eb == 32 ? 32 : dkp * eb + 16;
    const bool ones = eb == 2 && dkp % 32 == 16;                            // AttnCore<>::ones_row(): one all-ones V^T row per head
    const size_t kv2 = 2 * ((size_t)NP * ks + (size_t)(dkp + (ones ? 1 : 0)) * (NP * eb + 16));
    const bool dma = eb == 2 && slice_bytes(C) == 128 && C > 128;         // = attn_mlp_dma<>() of the instantiation dispatch_np2 picks
    const size_t ring = dma ? (size_t)DMFF_NSD * 128 * 128 : (eb == 4 || slice_bytes(C) == 128) ? Ring<128>::BYTES : Ring<64>::BYTES;
    const size_t chain = hb + ring + 2 * 64 * sizeof(float);
    return tile + (kv2 > chain ? kv2 : chain);
}

template <int DT, int DKP, int NP2, int SLB>
static int launch_attn_mlp(const DmffP& p, hipStream_t s) {
    const size_t lds = attn_mlp_lds(p.C, p.N, DKP, Elem<DT>::BYTES);
    if (lds > 160 * 1024) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_attn_mlp: %zu bytes of LDS (C=%d, N=%d, dk=%d) exceed 160 KiB", lds, p.C, p.N, p.dk);
    static std::atomic<bool> attr_set[ICAF_MAX_DEVICES];
    int dev = 0;
    ICAF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= ICAF_MAX_DEVICES) return fail(ICAF_ERR_UNSUPPORTED, "device ordinal %d", dev);
    if (!attr_set[dev]) {
        ICAF_HIP(hipFuncSetAttribute((const void*)dmff_attn_mlp_kernel<DT, DKP, NP2, SLB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    dim3 grid((unsigned)(((p.N + TMROWS - 1) / TMROWS) * p.B * 2));
    hipLaunchKernelGGL((dmff_attn_mlp_kernel<DT, DKP, NP2, SLB>), grid, dim3(FT), lds, s, p);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <int DT, int DKP>
static int dispatch_np2(const DmffP& p, hipStream_t s) {
    const int np2 = (p.C + 127) / 128;
    if (slice_bytes(p.C) == 128) {
        if (np2 <= 1) return launch_attn_mlp<DT, DKP, 1, 128>(p, s);
        if (np2 <= 2) return launch_attn_mlp<DT, DKP, 2, 128>(p, s);
        if (np2 <= 3) return launch_attn_mlp<DT, DKP, 3, 128>(p, s);
        if (np2 <= 4) return launch_attn_mlp<DT, DKP, 4, 128>(p, s);
    } else {                                    // C = 64, 192, 320, 448: 64-byte slices
        if (np2 <= 1) return launch_attn_mlp<DT, DKP, 1, 64>(p, s);
        if (np2 <= 2) return launch_attn_mlp<DT, DKP, 2, 64>(p, s);
        if (np2 <= 4) return launch_attn_mlp<DT, DKP, 4, 64>(p, s);
    }
    return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_attn_mlp: C=%d > 512 (the [64 x C] token tile and K / V^T of two heads exceed the LDS)", p.C);
}

// fp32 (parity build of the same template): C <= 128, head dims 8 .. 32 — what fits the LDS plan with 4-byte tiles
static int dispatch_attn_mlp_f32(const DmffP& p, hipStream_t s) {
    if (p.C > 128 || p.dk > 32) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_attn_mlp: the fp32 instantiation covers C <= 128, head dim <= 32 (C=%d, dk=%d)", p.C, p.dk);
    if (p.dk <= 16) return launch_attn_mlp<ICAF_F32, 16, 1, 128>(p, s);
    return launch_attn_mlp<ICAF_F32, 32, 1, 128>(p, s);
}

template <int DT>
static int dispatch_attn_mlp(const DmffP& p, hipStream_t s) {
    if (p.dk <= 16) return dispatch_np2<DT, 16>(p, s);
    if (p.dk <= 32) return dispatch_np2<DT, 32>(p, s);
    if (p.dk <= 48) return dispatch_np2<DT, 48>(p, s);
    if (p.dk <= 64) return dispatch_np2<DT, 64>(p, s);
    if (p.dk <= 96) return dispatch_np2<DT, 96>(p, s);
    if (p.dk <= 128) return dispatch_np2<DT, 128>(p, s);
    return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_attn_mlp: head dim %d > 128", p.dk);
}

template <int DT, int SLB>
static int launch_ln_qkv_t(const DmffP& p, hipStream_t s) {
    const size_t lds = (size_t)TMROWS * (p.C * Elem<DT>::BYTES + 16) + Ring<SLB>::BYTES;
    if (lds > 160 * 1024) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_ln_qkv: C=%d too wide", p.C);
    static std::atomic<bool> attr_set[ICAF_MAX_DEVICES];
    int dev = 0;
    ICAF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= ICAF_MAX_DEVICES) return fail(ICAF_ERR_UNSUPPORTED, "device ordinal %d", dev);
    if (!attr_set[dev]) {
        ICAF_HIP(hipFuncSetAttribute((const void*)dmff_ln_qkv_kernel<DT, SLB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[dev] = true;
    }
    const long long rows = (long long)p.B * p.N;
    dim3 grid((unsigned)((rows + TMROWS - 1) / TMROWS), (unsigned)((3 * p.C + QKV_GROUP - 1) / QKV_GROUP), 2u);
    hipLaunchKernelGGL((dmff_ln_qkv_kernel<DT, SLB>), grid, dim3(FT), lds, s, p);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}
template <int DT>
static int launch_ln_qkv(const DmffP& p, hipStream_t s) {
    if constexpr (DT == ICAF_F32) return launch_ln_qkv_t<DT, 128>(p, s);                  // (32 K elements per slice: C % 64 == 0 gives an even count)
    else return slice_bytes(p.C) == 128 ? launch_ln_qkv_t<DT, 128>(p, s) : launch_ln_qkv_t<DT, 64>(p, s);
}

static int fill(const icaf_dmff_args* a, DmffP& p, const char* who) {
    if (!a || !a->x || !a->qkv || !a->wqkv || !a->bqkv) return fail(ICAF_ERR_ARG, "%s: null pointer", who);
    if (a->dtype != ICAF_BF16 && a->dtype != ICAF_F16 && a->dtype != ICAF_F32) return fail(ICAF_ERR_ARG, "%s: bad dtype %d", who, a->dtype);
    if (a->dtype == ICAF_F32 && a->C > 128) return fail(ICAF_ERR_UNSUPPORTED, "%s: the fp32 instantiation covers C <= 128 (wider fp32 blocks use the per-layer launches)", who);
    if (a->B < 1 || a->N < 1 || a->heads < 1 || a->C % a->heads || a->B > 65535) return fail(ICAF_ERR_ARG, "%s: bad B/N/heads", who);
    if (a->C % 64 || a->C > 1024) return fail(ICAF_ERR_UNSUPPORTED, "%s: C=%d must be a multiple of 64 and <= 1024", who, a->C);
    if ((a->C / a->heads) % 8) return fail(ICAF_ERR_UNSUPPORTED, "%s: head dim %d must be a multiple of 8", who, a->C / a->heads);
    if (a->Kp < a->C || a->Kp % (a->dtype == ICAF_F32 ? 32 : 64)) return fail(ICAF_ERR_ARG, "%s: Kp=%d", who, a->Kp);
    p.x = a->x; p.qkv = a->qkv; p.y = a->y;
    p.wqkv = a->wqkv; p.bqkv = a->bqkv; p.wo = a->wo; p.bo = a->bo; p.w1 = a->w1; p.b1 = a->b1; p.w2 = a->w2; p.b2 = a->b2;
    p.ln_a_g[0] = a->ln_attn_gamma[0]; p.ln_a_g[1] = a->ln_attn_gamma[1]; p.ln_a_b[0] = a->ln_attn_beta[0]; p.ln_a_b[1] = a->ln_attn_beta[1];
    p.ln_m_g = a->ln_mlp_gamma; p.ln_m_b = a->ln_mlp_beta;
    p.wqkv_gs = a->wqkv_gs; p.bqkv_gs = a->bqkv_gs; p.wo_gs = a->wo_gs; p.bo_gs = a->bo_gs; p.w1_gs = a->w1_gs; p.b1_gs = a->b1_gs;
    p.w2_gs = a->w2_gs; p.b2_gs = a->b2_gs; p.x_gs = a->x_gs; p.y_gs = a->y_gs;
    p.B = a->B; p.N = a->N; p.C = a->C; p.heads = a->heads; p.dk = a->C / a->heads; p.Kp = a->Kp; p.Kp4 = a->Kp4; p.hid = a->hidden; p.ldy = a->ldy;
    p.eps_a = a->eps_attn; p.eps_m = a->eps_mlp;
    p.scale_l2e = (float)((1.0 / sqrt((double)p.dk)) * 1.4426950408889634);
    p.dbg = (long long*)a->debug_clock;
    for (int g = 0; g < 2; ++g) {
        p.c_res_a[g] = a->coef_res_attn[g]; p.c_acc_a[g] = a->coef_acc_attn[g];
        p.c_res_m[g] = a->coef_res_mlp[g]; p.c_acc_m[g] = a->coef_acc_mlp[g];
    }
    return ICAF_OK;
}

}  // namespace icaf

using namespace icaf;

extern "C" int icaf_dmff_attn_mlp_lds_bytes(int C, int N, int heads, int dtype, size_t* bytes) {
    if (!bytes || heads < 1 || C % heads) return fail(ICAF_ERR_ARG, "icaf_dmff_attn_mlp_lds_bytes: bad arguments");
    const int dk = C / heads;
    const int dkp = dk <= 16 ? 16 : dk <= 32 ? 32 : dk <= 48 ? 48 : dk <= 64 ? 64 : dk <= 96 ? 96 : 128;
    if (dtype == ICAF_F32) {
        const size_t need = C % 64 == 0 && C <= 128 && dk % 8 == 0 && dk <= 32 ? attn_mlp_lds(C, N, dkp, 4) : (size_t)-1;
        *bytes = need <= 160 * 1024 ? need : (size_t)-1;
        return ICAF_OK;
    }
    *bytes = (dtype == ICAF_BF16 || dtype == ICAF_F16) && C % 64 == 0 && C <= 512 && dk % 8 == 0 && dk <= 128 ? attn_mlp_lds(C, N, dkp, 2) : (size_t)-1;
    return ICAF_OK;
}

extern "C" int icaf_dmff_ln_qkv(const icaf_dmff_args* a, icaf_stream_t s) {
    DmffP p{};
    int st = fill(a, p, "icaf_dmff_ln_qkv");
    if (st) return st;
    if (!a->ln_attn_gamma[0] || !a->ln_attn_gamma[1] || !a->ln_attn_beta[0] || !a->ln_attn_beta[1]) return fail(ICAF_ERR_ARG, "icaf_dmff_ln_qkv: LayerNorm parameters missing");
    if (a->dtype == ICAF_F32) return launch_ln_qkv<ICAF_F32>(p, S(s));
    return a->dtype == ICAF_BF16 ? launch_ln_qkv<ICAF_BF16>(p, S(s)) : launch_ln_qkv<ICAF_F16>(p, S(s));
}

extern "C" int icaf_dmff_attn_mlp(const icaf_dmff_args* a, icaf_stream_t s) {
    DmffP p{};
    int st = fill(a, p, "icaf_dmff_attn_mlp");
    if (st) return st;
    if (!a->y || !a->wo || !a->bo || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->ln_mlp_gamma || !a->ln_mlp_beta) return fail(ICAF_ERR_ARG, "icaf_dmff_attn_mlp: null pointer");
    if (a->hidden % 128 || a->hidden < 128 || a->Kp4 < a->hidden || a->Kp4 % (a->dtype == ICAF_F32 ? 32 : 64)) return fail(ICAF_ERR_UNSUPPORTED, "icaf_dmff_attn_mlp: hidden width %d must be a multiple of 128", a->hidden);
    if (a->ldy < a->C || a->ldy % 4) return fail(ICAF_ERR_ARG, "icaf_dmff_attn_mlp: ldy=%d", a->ldy);
    if (a->dtype == ICAF_F32) return dispatch_attn_mlp_f32(p, S(s));
    return a->dtype == ICAF_BF16 ? dispatch_attn_mlp<ICAF_BF16>(p, S(s)) : dispatch_attn_mlp<ICAF_F16>(p, S(s));
}
