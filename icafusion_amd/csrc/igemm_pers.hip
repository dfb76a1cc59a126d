// PERSISTENT, BALANCED implicit GEMM for the long-K layers on gfx950 (MI355X) — yolov5l's 3x3 256 -> 256 / 512 -> 512 layers and 1x1 layers with K >= 512
// (reference models/common.py:48-60, 184-227; widths models/transformer/yolov5l_Transfusion_*.yaml), nn.Linear :607-618 / :704-709.  16-bit types;
// same arithmetic and same BITS as igemm.hip / igemm_wreg.hip (K order, MFMA step, bias / activation / residual expressions).
//
// What was wrong with one-tile-per-workgroup launches on these layers (round 4, `igemm_wreg` 128 x 256 with four waves, two workgroups per CU:
// 0.32 of the MFMA roof at 1.7 - 2.0 x the algorithmic bytes, ablation in tools/probes/abl_wreg_l.sh):
//   * ROUNDS: a 40 x 40 layer of yolov5l at batch 32 is 400 - 800 tiles of 128 pixels for 512 workgroup slots — 1.56 rounds, the chip is 78 % full
//     on average and every workgroup's prologue (first slices: a memory round trip with nothing to do) and epilogue are serial;
//   * WEIGHT STREAM: a 128-pixel tile re-uses a weight byte for 128 MACs; both co-resident workgroups stream the same fragments out of L2;
//   * two co-resident workgroups re-fetch each other's taps (PMC 1.97 x).
// Here ONE workgroup of eight waves per CU owns a contiguous SPAN of 32-pixel blocks of one (stream, 256-channel tile), sized so that every CU gets the
// same number of blocks to within one (3200 blocks over 256 CUs = 12 or 13 each: 96 % full instead of 78 %), and walks it in CHUNKS of up to 224 pixels:
//   * wave w = output channels [32 w, 32 w + 32) x ALL pixels of the chunk (TM <= 7 accumulator tiles): a weight fragment, fetched once per workgroup
//     straight into registers (fragment-major copy, igemm_wreg's scheme, one slice ahead), feeds up to seven MFMAs — about half the weight bytes per
//     MAC of the 128-pixel tiles; the pixel operand (<= 28 KiB per 64-element K slice) goes through a four-stage LDS-DMA ring;
//   * the K slices of ALL chunks form ONE stream: the first slices of chunk u + 1 are issued during the last steps of chunk u, and its first weight
//     slice before the epilogue of chunk u — no memory round trip is ever exposed between chunks;
//   * the epilogue is PER WAVE (a 32 x 32 tile is transposed through a private 2.5 KiB LDS buffer into 64-byte runs per pixel): no workgroup barrier, no
//     staging buffer the size of the tile, so the ring keeps its 128 KiB and stays busy through it;
//   * chunk sizes are balanced inside a span (13 blocks = 7 + 6), every TM in 1 .. 7 has its own unrolled loop (selected per chunk, uniform for the
//     workgroup), and every wave issues the SAME number of vector-memory operations per K step whatever TM is (rows beyond the chunk are out-of-range
//     DMA: zero fill, no traffic) — the counted `s_waitcnt vmcnt` below relies on that.
// XCD placement: the two streams (groups = 2) take four XCDs each; an XCD owns a contiguous range of a stream's pixels and runs the workgroups of
// all channel tiles of a span side by side, so taps and halo rows are shared through ONE L2 and the weights of a stream stay in its own XCDs' L2s.
//
// MEASURED (round 5): bit-identical on first run; isolated, 5 - 16 % faster than the round-4 choices on the paired long-K layers
// (0.41 of the MFMA roof on the 3x3 256 -> 256 layer), slower on the single-stream head layers; a forward run alone gets 1 - 2.6 % shorter, but with two
// forwards in flight (the bench) it LOSES 1 %: a workgroup that owns a CU shares it with nothing.  Ablations put the pixel operand's LDS-DMA first
// (16 bytes / clock / CU: the feed limit) — bytes per MAC that no tile height changes.  Hence opt-in: ICAF_PERS_GEMM=1 (ops.conv_candidates).
#include "conv_common.h"

// Ablation switches for timing studies (tools/quick_variant.py <tag> igemm_pers.hip -DICAF_PERS_ABL=n; results are then meaningless):
//   1 = no weight loads, 2 = no pixel DMA, 4 = no LDS fragment reads, 8 = no workgroup barrier in the K step, 16 = no epilogue, 32 = no MFMAs
#ifndef ICAF_PERS_ABL
#define ICAF_PERS_ABL 0
#endif
// 1: every wave waits, at the top of step q, for ITS portion of pixel slice q + 1 as well (two slices in flight instead of three), so the barrier of step q
//    publishes slice q + 1 too and the first fragment group of the next slice is read at the END of a step, under its last MFMAs — nothing but the barrier
//    itself stands between the MFMAs of two slices.  0: slice q only, the first group of a slice is read behind its barrier (A/B builds).
#ifndef ICAF_PERS_EARLY
#define ICAF_PERS_EARLY 1
#endif

namespace icaf {

constexpr int PERS_BN = 256, PERS_TMAX = 7, PERS_NW = 8, PERS_NS = 4, PERS_RB = 128;      // (TMAX = 8 fits alone — 244 registers — but not beside the other seven loop bodies)
constexpr int PERS_STAGE = 8 * 32 * PERS_RB;                         // 32 KiB: room for 256 pixel rows x 128 bytes of K (every wave issues 4 DMA instructions per slice)
constexpr int PERS_EPITCH = 80;                                      // bytes per pixel row of a wave's private transposition buffer (64 + 16)
constexpr int PERS_EBUF = 32 * PERS_EPITCH;                          // 2560 bytes per wave
constexpr int PERS_LDS = PERS_NS * PERS_STAGE + PERS_NW * PERS_EBUF; // 151,552 bytes: one workgroup per CU

// MODE 1: 1x1 / stride 1 / pad 0 (plain row-major pixel matrix); MODE 2: any filter with Cin * bytes a multiple of 128 (a K slice lies inside one
// tap: wave-uniform tap walk) — igemm's address generators.  p.mtiles = 32-pixel blocks per stream, p.ntiles = 256-channel tiles, p.nchunks = K slices.
template <int DT, int ACT, int MODE>
__global__ __launch_bounds__(PERS_NW * 64) void igemm_pers_kernel(const ConvP p, const void* __restrict__ wfrag, const long long wf_gs, const int groups) {
    using E = Elem<DT>;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int RB = PERS_RB, NS = PERS_NS, NW = PERS_NW, STAGE = PERS_STAGE;
    constexpr int VEC = E::VEC, BK = RB / E::BYTES;                  // 8 elements per 16 bytes, 64 K elements per slice
    constexpr int NA = 4;                                            // DMA instructions per wave and slice: 32 x 1 KiB = 256 rows of ring stage — always issued
    constexpr int NSTEP = RB / 32;                                   // 4 MFMA steps (= weight fragments) per slice
    constexpr int PER = NA + NSTEP;                                  // vector-memory operations per wave and K step
    constexpr int NB = 2;                                            // weight register buffers = unroll of the K loop: slices c, c + 1 (one slice = up to
                                                                     // 1024 cycles of this wave's MFMAs of cover; a third buffer costs the fragment double-buffer its registers)
    static_assert(2 * PER <= 18, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- which span of which (stream, channel tile) this workgroup owns (header) --------------------------------------------------------------
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int xg_n = 8 / groups, g = xcd / xg_n, xg = xcd - g * xg_n;
    const int NT = p.ntiles, S = per_xcd / NT;                       // spans per XCD
    if (idx >= S * NT) return;                                       // (workgroup-uniform, before any barrier)
    const int nt = idx % NT, sp = idx / NT;
    const int ST = S * xg_n, s_id = xg * S + sp;                     // spans per stream, this workgroup's span
    const int blk_lo = (int)((long long)s_id * p.mtiles / ST), blk_hi = (int)((long long)(s_id + 1) * p.mtiles / ST);
    const int nblk = blk_hi - blk_lo;
    if (nblk <= 0) return;
    const int nck = (nblk + PERS_TMAX - 1) / PERS_TMAX, ck_base = nblk / nck, ck_rem = nblk - ck_base * nck;   // chunk k: ck_base + (k < ck_rem) blocks
    const int n0 = nt * PERS_BN;
    const int nch = p.nchunks, nchp = (nch + NB - 1) / NB * NB;      // K slices of the layer; steps per chunk (padded with zero slices)

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- pixel operand: lane -> (row, 16-byte slot) of a DMA instruction, igemm's RB = 128 swizzle (key(row) = (row >> 1) & 7) -------------------
    const int rsub = lane >> 3;
    const int dkey = ((wave & 1) << 2) | (rsub >> 1);                // key(row) for row = (wave + NW * i) * 8 + rsub (NW even)
    const int lslot = (lane & 7) ^ dkey;
    // issue cursor: chunk ick (first block iblk), slice ic of it; runs NS - 1 slices ahead of the consume cursor, across chunk boundaries
    int ick = 0, iblk = blk_lo, ic = 0;
    int kc = 0, ky = 0, kx = 0;                                       // MODE 2: K position of the cursor's slice (wave-uniform)
    unsigned a_off[NA];
    unsigned a_mask[NA / 2];                                          // MODE 2: per DMA instruction, bit t = tap t (ky * kw + kx) reads inside the image (16 bits each)
    const int ntaps = p.kh * p.kw;
    int tap = 0;                                                      // ky * kw + kx of the cursor's slice
    auto issue_setup = [&]() {                                        // per-chunk lane offsets of the cursor's chunk (dead rows / no chunk left: OOB)
        const bool live = ick < nck;
        const int rows = live ? (ck_base + (ick < ck_rem ? 1 : 0)) * 32 : 0;
        if constexpr (MODE == 2) a_mask[0] = a_mask[1] = 0u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = (wave + NW * i) * 8 + rsub;
            const int m = iblk * 32 + row;
            const bool ok = row < rows && m < p.M;
            if constexpr (MODE == 1) {
                a_off[i] = ok ? ((unsigned)m * (unsigned)p.ldx + (unsigned)(lslot * VEC)) * E::BYTES : OOB;
            } else {
                const int mm = ok ? m : 0;
                const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, b = t / p.Ho;
                const int h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
                a_off[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES
                         + (unsigned)((h0 * p.W + w0) * p.ldx + lslot * VEC) * E::BYTES;                  // tap (0, 0) (may wrap below 0)
                unsigned mk = 0;
                for (int t2 = 0, yy = 0; yy < p.kh; ++yy)
                    for (int xx = 0; xx < p.kw; ++xx, ++t2)
                        mk |= (ok && (unsigned)(h0 + yy) < (unsigned)p.H && (unsigned)(w0 + xx) < (unsigned)p.W) ? (1u << t2) : 0u;
                a_mask[i >> 1] |= mk << ((i & 1) * 16);
            }
        }
    };
    auto issue_part = [&](int stage, int part) {                      // DMA instruction `part` (of NA = NSTEP) of the cursor's slice
        unsigned char* st = lds + stage * STAGE;
        unsigned tap_delta = 0;
        if constexpr (MODE == 2) tap_delta = (unsigned)((ky * p.W + kx) * p.ldx + kc) * E::BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i != part) continue;
            unsigned voff;
            if constexpr (MODE == 1) {
                voff = (a_off[i] != OOB && ic < nch) ? a_off[i] + (unsigned)ic * RB : OOB;
            } else {
                const bool ok = tap < ntaps && ((a_mask[i >> 1] >> ((i & 1) * 16 + tap)) & 1u);          // (tap == ntaps: a padding slice past K)
                voff = ok ? a_off[i] + tap_delta : OOB;
            }
            if constexpr (!(ICAF_PERS_ABL & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(st + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto issue_advance = [&]() {                                      // (wave-uniform)
        if constexpr (MODE == 2) {
            kc += BK;
            if (kc >= p.Cin) { kc = 0; ++tap; if (++kx == p.kw) { kx = 0; ++ky; } }
        }
        if (++ic == nchp) {
            ic = 0; kc = 0; ky = 0; kx = 0; tap = 0;
            iblk += ck_base + (ick < ck_rem ? 1 : 0);
            ++ick;
            issue_setup();
        }
    };

    // ---- weight operand: fragment-major [channel block of 32][MFMA step of 16 K][lane][8 elements]; this wave's channel block -------------------
    const int ksteps = p.Kp / 16;                                     // MFMA steps per channel block row
    // (a wave-uniform 64-bit base in scalar registers + ONE 32-bit lane offset: global_load ... v_off, s[base] — per-load 64-bit vector addresses cost
    //  this kernel the registers of its fragment double-buffer)
    const unsigned char* __restrict__ wbase = (const unsigned char*)((const typename E::type*)wfrag + g * wf_gs)
                                              + ((long long)(n0 / 32 + wave) * ksteps) * 1024;
    const unsigned wlane = (unsigned)lane * 16u;
    const int last_step = ksteps - 1;
    auto load_w = [&](u32x4 (&dst)[NSTEP], int chunk) {               // unconditional, clamped: past the end the last fragments are re-read
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            int ks = chunk * NSTEP + s;
            ks = ks < last_step ? ks : last_step;
            if constexpr (ICAF_PERS_ABL & 1) dst[s] = u32x4{(unsigned)ks, (unsigned)lane, 0x3f803f80u, 0x3f803f80u};
            else dst[s] = *(const u32x4*)(wbase + (long long)ks * 1024 + wlane);
        }
    };

    const int fkey = (l31 >> 1) & 7;
    int foff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);

    // epilogue constants
    const float alpha_acc = p.alpha_acc[g], alpha_res = p.alpha_res[g];
    const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
    typename E::type* __restrict__ yg = (typename E::type*)p.y + g * p.y_gs;
    const typename E::type* __restrict__ rg = p.res ? (const typename E::type*)p.res + g * p.res_gs : nullptr;
    unsigned char* ebuf = lds + NS * STAGE + wave * PERS_EBUF;        // this wave's transposition buffer
    const int nw0 = n0 + wave * 32;                                   // the wave's first output channel

    // ---- prologue: weight fragments of slice 0 -> registers; pixel slices 0 .. NS - 2 -> ring ---------------------------------------------------
    u32x4 fw[NB][NSTEP];
    load_w(fw[0], 0);
    issue_setup();
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
        for (int part = 0; part < NSTEP; ++part) issue_part(s, part);
        issue_advance();
    }
    wait_vmcnt<0>();                              // (the steps below then always find "everything older than two steps" complete)

    int q = 0;                                    // slices consumed so far by this workgroup: slice q lives in stage q % NS
    int cblk = blk_lo;                            // the consume cursor's chunk: first block

    // One chunk of TM blocks: K loop + per-wave epilogue.
    auto run_chunk = [&](auto TMtag, const bool has_next) {
        constexpr int TM = decltype(TMtag)::value;
        f32x16 acc[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
        // One slice step.  P = c % NB selects the register buffer of the weight fragments of slice c (compile-time: the K loop is unrolled by NB; every
        // chunk starts at P = 0 — its slice 0 was requested into buffer 0 before the previous chunk's epilogue).  Issues, in this order, the weight loads
        // of slice c + 1 and the pixel DMA of stream slice q + NS - 1: exactly PER vector-memory operations per wave.
        constexpr int TA = TM > 4 ? 4 : TM, TB = TM - TA, GPS = TB > 0 ? 2 : 1, NG = NSTEP * GPS;      // fragment groups per step, per slice
        u32x4 fa[TA], fb[TB > 0 ? TB : TA];            // (fa carries the NEXT slice's first group across the step boundary: ICAF_PERS_EARLY)
        auto step = [&](auto Ptag, int c) {
            constexpr int P = decltype(Ptag)::value;
            // stream slice q (EARLY: q + 1) was issued three (two) steps ago or in the prologue: complete once at most the 2 * PER (PER) operations of the
            // last two steps (last step) are outstanding.  Whatever else may be outstanding (a previous chunk's weight pre-loads and epilogue stores) is
            // YOUNGER than that slice, so it can only make this wait stricter, never let it pass early (loads complete in order).
            wait_vmcnt<ICAF_PERS_EARLY ? PER : 2 * PER>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(ICAF_PERS_ABL & 8))
                __builtin_amdgcn_s_barrier();      // (a) slice q (and q + 1) visible to every wave, (b) stage (q - 1) % NS is free
            const unsigned char* a_s = lds + (q & (NS - 1)) * STAGE;
            const unsigned char* a_n = lds + ((q + 1) & (NS - 1)) * STAGE;
            load_w(fw[(P + NB - 1) % NB], c + NB - 1);
            const int sfree = (q + NS - 1) & (NS - 1);
            // fragment reads in GROUPS of up to four 32-pixel blocks, two groups in flight: the reads of group k + 1 are issued before the MFMAs of
            // group k (left to itself under this register budget the compiler read two fragments at a time and waited for each)
            auto rd = [&](auto& dst, auto Ktag, const unsigned char* base) {
                constexpr int k = decltype(Ktag)::value, s = k / GPS, b0 = (k % GPS) * TA, n = (k % GPS) ? TB : TA;
#pragma unroll
                for (int b = 0; b < n; ++b) {
                    if constexpr (ICAF_PERS_ABL & 4) dst[b] = u32x4{(unsigned)(q + b), (unsigned)foff[s], 0x3f803f80u, 0x3f803f80u};
                    else dst[b] = *(const u32x4*)(base + ((b0 + b) * 32) * RB + foff[s]);
                }
            };
            auto mm = [&](auto& src, auto Ktag) {
                constexpr int k = decltype(Ktag)::value, s = k / GPS, b0 = (k % GPS) * TA, n = (k % GPS) ? TB : TA;
#pragma unroll
                for (int b = 0; b < n; ++b) {
                    if constexpr (ICAF_PERS_ABL & 32) acc[b0 + b][0] += __uint_as_float(fw[P][s][0] ^ src[b][0]);
                    else mma_step<DT>(acc[b0 + b], fw[P][s], src[b]);
                }
                if constexpr (k % GPS == GPS - 1) issue_part(sfree, s);
            };
            using K0 = std::integral_constant<int, 0>;
            auto grp = [&](auto Ktag, auto& self) -> void {       // (compile-time recursion over the groups of the slice; NG is even: the last group reads fb)
                constexpr int k = decltype(Ktag)::value;
                if constexpr (k < NG) {
                    using K1 = std::integral_constant<int, k + 1>;
                    if constexpr (k % 2 == 0) { rd(fb, K1{}, a_s); mm(fa, Ktag); }
                    else {
                        if constexpr (k + 1 < NG) rd(fa, K1{}, a_s);
                        else if (ICAF_PERS_EARLY && c + 1 < nchp) rd(fa, K0{}, a_n);          // the next slice's first group (same chunk: same TM)
                        mm(fb, Ktag);
                    }
                    self(K1{}, self);
                }
            };
            if (!ICAF_PERS_EARLY || c == 0) rd(fa, K0{}, a_s);    // (a chunk's first slice: nothing was read ahead)
            grp(K0{}, grp);
            issue_advance();
            ++q;
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        // (NB step bodies per TM and ONE loop exit: a chunk always runs a multiple of NB steps — slices past K are zero-fill DMA against clamped weight
        //  fragments, i.e. exact zeros added to the accumulators: only layers with an odd number of 64-element slices pad, by one)
        for (int c = 0; c < nchp; c += NB) {
            step(P0{}, c);
            step(P1{}, c + 1);
        }

        // the next chunk's first weight slice: in flight through this chunk's epilogue (both buffers are dead now)
        if (has_next) load_w(fw[0], 0);

        // ---- per-wave epilogue: bias + activation in registers (conv_common.h's expressions), 32 x 32 tile -> private LDS buffer -> 64-byte runs ----
        f32x4 bq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float* bp = bias ? bias + nw0 + 8 * qd + 4 * hi : (const float*)p.w;      // (any mapped address: the value is discarded)
            const f32x4 t = *(const f32x4*)bp;
            bq[qd] = bias ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int m_base = cblk * 32;
#pragma unroll
        for (int b = 0; b < ((ICAF_PERS_ABL & 16) ? 1 : TM); ++b) {
            // this tile's two output vectors per lane: pixel row r_ = (lane + 64 j) >> 2, 16-byte channel vector cv = lane & 3
            int mrow[2];
            u32x4 rv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m_base + b * 32 + ((lane + 64 * j) >> 2);
                mrow[j] = m < p.M ? m : -1;
                if (rg) rv[j] = *(const u32x4*)(rg + (long long)(m < p.M ? m : 0) * p.ldr + nw0 + (lane & 3) * VEC);     // (clamped row: never used)
            }
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float pv = 0.0f;                               // (the shared epilogue's pre-activation term: the SAME expression keeps the same bits)
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[b][4 * qd + j] + bq[qd][j] + pv;
                apply_act4<ACT, DT>(v, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= alpha_acc;
                u32x2 pk;
                if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                *(u32x2*)(ebuf + l31 * PERS_EPITCH + (8 * qd + 4 * hi) * E::BYTES) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: its LDS operations execute in order; this only orders the compiler's view)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r_ = (lane + 64 * j) >> 2, cv = lane & 3;
                u32x4 o = *(const u32x4*)(ebuf + r_ * PERS_EPITCH + cv * 16);
                if (rg) {
                    float v[VEC], r[VEC];
                    unpack16<DT>(o, v);
                    unpack16<DT>(rv[j], r);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] = __builtin_fmaf(alpha_res, r[e], v[e]);
                    o = pack16<DT>(v);
                }
                if (mrow[j] >= 0) *(u32x4*)(yg + (long long)mrow[j] * p.ldy + nw0 + cv * VEC) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the buffer is re-written by the next tile
        }
        cblk += TM;
    };

    for (int k = 0; k < nck; ++k) {
        const int tm = ck_base + (k < ck_rem ? 1 : 0);
        const bool has_next = k + 1 < nck;
        switch (tm) {                              // (workgroup-uniform)
            case 1: run_chunk(std::integral_constant<int, 1>{}, has_next); break;
            case 2: run_chunk(std::integral_constant<int, 2>{}, has_next); break;
            case 3: run_chunk(std::integral_constant<int, 3>{}, has_next); break;
            case 4: run_chunk(std::integral_constant<int, 4>{}, has_next); break;
            case 5: run_chunk(std::integral_constant<int, 5>{}, has_next); break;
            case 6: run_chunk(std::integral_constant<int, 6>{}, has_next); break;
            default: run_chunk(std::integral_constant<int, 7>{}, has_next); break;
        }
    }
    wait_vmcnt<0>();                               // zero-fill slices issued past the end of the stream
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
static int pers_grid() {                           // one workgroup per CU, a multiple of 8 (XCD-aware index arithmetic); per device
    static std::atomic<int> cus[ICAF_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ICAF_MAX_DEVICES) return 256;
    int n = cus[dev];
    if (!n) {
        hipDeviceProp_t pr;
        n = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256;
        n = n / 8 * 8;
        if (n < 8) n = 8;
        cus[dev] = n;
    }
    return n;
}

int pers_check(const icaf_conv_args* a, const ConvP& p) {
    if (!a->wf) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: no fragment-major weights (icaf_conv_args.wf)");
    if (a->dtype == ICAF_F32 || a->out_dtype != a->dtype) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: 16-bit types, out dtype == dtype");
    if ((a->Cin * 2) % 128) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: Cin * 2 bytes must be a multiple of 128 (Cin = %d)", a->Cin);
    if (a->pre || a->w2) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: no pre-activation term / chained layer");
    if (p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: operand exceeds the 2 GiB buffer-descriptor range");
    if (a->Kp % 64) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: Kp must be a multiple of 64");
    if (((uintptr_t)a->wf & 15) || (a->wf_gs * 2) % 16) return fail(ICAF_ERR_ARG, "igemm_pers: wf must be 16-byte aligned");
    if (a->Cout % PERS_BN) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: Cout = %d is not a multiple of %d (whole channel tiles only)", a->Cout, PERS_BN);
    if (!p.vec_y || (a->res && !p.vec_r)) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: y / res must be 16-byte aligned with ld %% 8 == 0");
    if (a->bias && ((uintptr_t)a->bias & 15)) return fail(ICAF_ERR_ARG, "igemm_pers: bias must be 16-byte aligned");
    if (a->Cout / PERS_BN > 32) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: more than 32 channel tiles (Cout = %d)", a->Cout);
    if (a->kh * a->kw > 16) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: at most 16 filter taps (%d x %d)", a->kh, a->kw);
    if (a->act == ICAF_ACT_GELU && !(a->kh == 1 && a->kw == 1 && a->sh == 1 && a->sw == 1 && a->ph == 0 && a->pw == 0))
        return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: GELU is built for 1x1 layers (Linear) only");
    return ICAF_OK;
}

template <int DT, int ACT>
static int launch_pers_act(const icaf_conv_args* a, const ConvP& p, hipStream_t s) {
    ConvP q = p;
    q.mtiles = (p.M + 31) / 32;                    // 32-pixel blocks per stream
    q.ntiles = p.Cout / PERS_BN;
    q.nchunks = (p.K + 63) / 64;
    const int grid = pers_grid();
    if (q.ntiles > grid / 8) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: %d channel tiles for %d workgroups per XCD", q.ntiles, grid / 8);
    const bool plain = q.kh == 1 && q.kw == 1 && q.sh == 1 && q.sw == 1 && q.ph == 0 && q.pw == 0;
    auto go = [&](auto kern) -> int {
        ICAF_LDS_OPTIN(kern, PERS_LDS);
        kern<<<dim3((unsigned)grid), dim3(PERS_NW * 64), PERS_LDS, s>>>(q, a->wf, a->wf_gs, a->groups);
        ICAF_LAUNCH_CHECK();
        return ICAF_OK;
    };
    if (plain) return go(igemm_pers_kernel<DT, ACT, 1>);
    if constexpr (ACT == ICAF_ACT_GELU) return fail(ICAF_ERR_UNSUPPORTED, "igemm_pers: GELU is built for 1x1 layers (Linear) only");
    else return go(igemm_pers_kernel<DT, ACT, 2>);
}

template <int DT>
static int launch_pers_dt(const icaf_conv_args* a, const ConvP& p, hipStream_t s) {
    if (p.act == ICAF_ACT_SILU) return launch_pers_act<DT, ICAF_ACT_SILU>(a, p, s);
    if (p.act == ICAF_ACT_GELU) return launch_pers_act<DT, ICAF_ACT_GELU>(a, p, s);
    return launch_pers_act<DT, ICAF_ACT_NONE>(a, p, s);
}

int launch_pers(const icaf_conv_args* a, const ConvP& p, hipStream_t s) {
    int st = pers_check(a, p);
    if (st) return st;
    return a->dtype == ICAF_BF16 ? launch_pers_dt<ICAF_BF16>(a, p, s) : launch_pers_dt<ICAF_F16>(a, p, s);
}

}  // namespace icaf
