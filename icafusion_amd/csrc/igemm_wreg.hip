// Implicit-GEMM convolution with the WEIGHTS FED FROM REGISTERS for gfx950 (MI355X) — same layers, same arithmetic and same
// bits as igemm.hip (reference models/common.py:48-60, 184-194; nn.Linear :607-618 / :704-709), 16-bit types.
//
// What bounds igemm.hip on every layer with a long K loop (3x3 layers at 40x40 / 20x20, 1x1 layers with K >= 512) is neither HBM
// nor the matrix pipe but the L2 -> LDS feed of `buffer_load ... lds`: measured 10-12 TB/s for the whole chip (18-20 bytes / clock /
// CU), and a 128 x 128 tile needs 64 bytes of operands per 4096 MACs — half of them WEIGHTS, which every tile of the layer re-reads.
// Here the weight operand does not pass through LDS at all:
//   * the packed weights exist a second time in FRAGMENT-MAJOR order (icafusion_amd.ops.frag_weights): for every block of 32 output
//     channels and every MFMA step of 16 K elements, the 64 lanes' 16-byte A-operand fragments are 1 KiB of consecutive memory, so a
//     wave fetches its fragment with ONE fully coalesced global_load_dwordx4 (8 cache lines; L1 / L2 resident: the layer's weights
//     are shared by all workgroups) two K slices ahead of its use;
//   * a wavefront owns 32 output channels x ALL 128 pixels of the tile (TM = 4 accumulator tiles): no two waves of a workgroup
//     fetch the same weights, and one weight fragment feeds four MFMAs;
//   * only the pixel operand travels by LDS-DMA (igemm's ring: 128-byte slices, XOR-swizzled through the source address, one
//     barrier per slice, counted vmcnt) — 16 KiB per slice instead of 32, in a 4-stage ring: three slices in flight;
//   * K order, MFMA step (32x32x16), bias / activation / residual epilogue (conv_common.h) are igemm's: bit-identical results.
// Workgroup = NWV wavefronts = 128 pixels x 32 * NWV * TN channels (NWV = 4: 128 x 128, NWV = 8: 128 x 256 with TN = 1).
//
// Round 4 — TN = 2: a wavefront owns TWO 32-channel blocks (64 channels x 128 pixels: eight accumulator tiles), so a pixel fragment read
// from LDS feeds two MFMAs and the tile is twice as wide for the same pixel feed: 128 x 512 with eight wavefronts, 128 x 256 with four
// (two workgroups per CU).  The pixel operand's LDS-DMA bytes per MAC are 2 / BN: at BN = 256 a K slice is 16 KiB of feed (~850 cycles at
// the measured 18-20 bytes / clock / CU) against 1024 cycles of MFMAs per SIMD — co-limiting; at BN = 512 it is the same feed against 2048
// cycles.  For the layers with 512 or more output channels (yolov5l's P4 / P5 rows, yolov5s's 20 x 20 rows).  Register budget: 128 accumulator
// registers leave room for TWO weight buffers (one slice ahead: 32 MFMAs = 1024 cycles of cover) instead of three, so the counted wait of a step
// leaves only that step's own pixel DMA outstanding.  Same K order / MFMA step / epilogue: bit-identical to every other configuration.
#include "conv_common.h"

// Ablation switches for timing studies (tools/quick_variant.py -DICAF_WREG_ABL=n; results are then meaningless):
//   1 = no weight loads, 2 = no pixel DMA, 4 = no LDS fragment reads
#ifndef ICAF_WREG_ABL
#define ICAF_WREG_ABL 0
#endif

namespace icaf {

// MODE 1: 1x1 / stride 1 / pad 0 (plain row-major pixel matrix); MODE 2: any filter with Cin * bytes a multiple of 128 (a K slice
// lies inside one tap: wave-uniform tap walk).  igemm.hip's address generators, pixel operand only.
template <int DT, int NWV, int ACT, int MODE, int TN = 1, int BM_ = 128>
__global__ __launch_bounds__(NWV * 64, (NWV == 4 && TN == 2) ? 2 : 1) void igemm_wreg_kernel(const ConvP p, const void* __restrict__ wfrag, const long long wf_gs) {
    using E = Elem<DT>;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int BM = BM_, BN = 32 * NWV * TN, RB = 128, NS = 4, TM = BM / 32;      // (BM = 64: twice the workgroups for the layers with few pixels)
    constexpr int NB = TN == 1 ? 3 : 2;                                // weight register buffers: slices c, c + 1 (, c + 2)
    constexpr int VEC = E::VEC, BK = RB / E::BYTES;                  // 8, 64
    constexpr int RPI = 8, AI = BM / RPI, NA = AI / NWV;             // DMA instructions per slice: 16 per workgroup, 4 / 2 per wave
    constexpr int NSTEP = RB / 32;                                   // 4 MFMA steps (= weight fragments) per slice
    constexpr int PER = NA + NSTEP * TN;                             // vector-memory operations per wave and slice step
    constexpr int STAGE = BM * RB;                                   // 16 KiB (8 KiB at BM = 64)
    static_assert(AI % NWV == 0 && (TN == 1 ? 2 * PER : NA) <= 18, "tile shape / vmcnt immediate");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const int tile = xcd_tile(p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- pixel operand: lane -> (row, 16-byte slot) of a DMA instruction, igemm's RB = 128 swizzle ----------------------------
    const int rsub = lane >> 3;
    const int dkey = ((wave & 1) << 2) | (rsub >> 1);                // key(row) for row = (wave + NWV * i) * 8 + rsub (NWV even)
    const int lslot = (lane & 7) ^ dkey;
    unsigned a_off[NA];
    int a_h0[NA], a_w0[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave + NWV * i) * RPI + rsub;
        const int m = m0 + row;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if constexpr (MODE == 1) {
            a_off[i] = a_ok[i] ? ((unsigned)mm * (unsigned)p.ldx + (unsigned)(lslot * VEC)) * E::BYTES : OOB;
        } else {
            const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, b = t / p.Ho;
            a_h0[i] = ho * p.sh - p.ph;
            a_w0[i] = wo * p.sw - p.pw;
            a_off[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES
                     + (unsigned)((a_h0[i] * p.W + a_w0[i]) * p.ldx + lslot * VEC) * E::BYTES;      // tap (0, 0) (may wrap below 0)
        }
    }
    int kc = 0, ky = 0, kx = 0;                                       // MODE 2: wave-uniform K position of the NEXT slice to issue
    int ichunk = 0;
    auto issue_a = [&](int stage, int part, int nparts) {             // DMA instructions i with i % nparts == part
        unsigned char* st = lds + stage * STAGE;
        unsigned tap_delta = 0;
        bool kvalid = true;
        if constexpr (MODE == 2) { kvalid = ky < p.kh; tap_delta = (unsigned)((ky * p.W + kx) * p.ldx + kc) * E::BYTES; }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i % nparts != part) continue;
            unsigned voff;
            if constexpr (MODE == 1) {
                voff = (a_off[i] != OOB && ichunk < p.nchunks) ? a_off[i] + (unsigned)ichunk * RB : OOB;
            } else {
                const int h = a_h0[i] + ky, w = a_w0[i] + kx;
                const bool ok = a_ok[i] && kvalid && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                voff = ok ? a_off[i] + tap_delta : OOB;
            }
            if constexpr (!(ICAF_WREG_ABL & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(st + (wave + NWV * i) * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto advance_a = [&]() {
        ++ichunk;
        if constexpr (MODE == 2) {
            kc += BK;
            if (kc >= p.Cin) { kc = 0; if (++kx == p.kw) { kx = 0; ++ky; } }
        }
    };

    // ---- weight operand: fragment-major [channel block of 32][MFMA step of 16 K][lane][8 elements]; this wave's channel block ----
    const int ksteps = p.Kp / 16;                                     // MFMA steps per channel block row
    const u32x4* __restrict__ wf = (const u32x4*)((const typename E::type*)wfrag + g * wf_gs)
                                   + ((long long)(n0 / 32 + wave * TN) * ksteps) * 64 + lane;      // the wave's first channel block (TN consecutive blocks)
    const int last_step = ksteps - 1;
    auto load_w = [&](u32x4 (&dst)[NSTEP][TN], int chunk) {           // unconditional, clamped: past the end the last fragments are re-read
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            int ks = chunk * NSTEP + s;
            ks = ks < last_step ? ks : last_step;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if constexpr (ICAF_WREG_ABL & 1) dst[s][t] = u32x4{(unsigned)ks, (unsigned)lane, 0x3f803f80u, 0x3f803f80u};
                else dst[s][t] = wf[((long long)t * ksteps + ks) * 64];
            }
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.0f;

    const int fkey = (l31 >> 1) & 7;
    int foff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);

    // ---- prologue: weight fragments of slices 0 (, 1) -> registers; pixel slices 0 .. NS - 2 -> ring ----------------------------
    u32x4 fw[NB][NSTEP][TN];
    load_w(fw[0], 0);
    if constexpr (NB == 3) load_w(fw[1], 1);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) {
        issue_a(s, 0, 1);
        advance_a();
    }
    wait_vmcnt<0>();                              // (the steps below then always find "everything older than two steps" complete)

    // One slice step.  P = c % NB selects the register buffer of the weight fragments of slice c (compile-time: the K loop is
    // unrolled by NB).  Issues, in this order, the weight loads of slice c + NB - 1 and the pixel DMA of slice c + NS - 1: exactly PER
    // vector-memory operations per wave — the counted wait below relies on that.
    auto step = [&](auto Ptag, int c) {
        constexpr int P = decltype(Ptag)::value;
        if constexpr (NB == 3) {
            // pixel slice c was issued in step c - 3 (or the prologue): complete once at most the 2 * PER operations of steps c - 2 and
            // c - 1 are outstanding (the weights of slice c were issued in step c - 2, before that step's pixel DMA)
            wait_vmcnt<2 * PER>();
        } else {
            // two weight buffers: the weights of slice c were the FIRST operations of step c - 1; everything up to them is complete
            // once at most that step's NA pixel-DMA instructions (issued after them) are outstanding — pixel slices c and c + 1 with it
            wait_vmcnt<NA>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // (a) slice c visible to every wave, (b) stage (c - 1) % NS is free
        const unsigned char* a_s = lds + (c & (NS - 1)) * STAGE;
        load_w(fw[(P + NB - 1) % NB], c + NB - 1);
        const int sfree = (c + NS - 1) & (NS - 1);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            u32x4 fp[TM];
#pragma unroll
            for (int b = 0; b < TM; ++b) {
                if constexpr (ICAF_WREG_ABL & 4) fp[b] = u32x4{(unsigned)(c + b), (unsigned)foff[s], 0x3f803f80u, 0x3f803f80u};
                else fp[b] = *(const u32x4*)(a_s + (b * 32) * RB + foff[s]);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int b = 0; b < TM; ++b) mma_step<DT>(acc[t][b], fw[P][s][t], fp[b]);
            issue_a(sfree, s, NSTEP);
        }
        advance_a();
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    int c = 0;
    if constexpr (NB == 3) {
        for (; c + 3 <= p.nchunks; c += 3) {
            step(P0{}, c);
            step(P1{}, c + 1);
            step(P2{}, c + 2);
        }
        if (c < p.nchunks) step(P0{}, c);
        if (c + 1 < p.nchunks) step(P1{}, c + 1);
    } else {
        for (; c + 2 <= p.nchunks; c += 2) {
            step(P0{}, c);
            step(P1{}, c + 1);
        }
        if (c < p.nchunks) step(P0{}, c);
    }
    wait_vmcnt<0>();                               // zero-fill slices issued past the end
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // (workgroup-uniform: most launches write whole tiles in vectors without a residual — the checked fast write-back of conv_common.h.  Carrying
    //  both loops costs the 128-pixel / 32-channel-per-wave tiles a workgroup per CU in registers, so only the 64-pixel and the 64-channel forms do)
    constexpr bool FAST_WB = (BM == 64 || TN == 2) && !ICAF_EPI_FAST;          // (only the ICAF_EPI_FAST=0 build: the shared epilogue has its own fast loops now)
    if (FAST_WB && !p.res && p.vec_y && p.Cout % BN == 0)
        epilogue<DT, DT, BM, BN, BM, 32 * TN, ACT, false, false, false, true>(acc, lds, p, g, [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; }, n0);
    else
        epilogue<DT, DT, BM, BN, BM, 32 * TN, ACT, false>(acc, lds, p, g, [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; }, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
const char* wreg_tag(int shape) {
    static const char* t[] = {"?", "128x128", "128x256", "128x512", "128x256w4", "64x256", "64x128"};
    return shape >= 1 && shape <= 6 ? t[shape] : "?";
}

int wreg_check(const icaf_conv_args* a, const ConvP& p, int shape) {
    if (shape < 1 || shape > 6) return fail(ICAF_ERR_ARG, "igemm_wreg: unknown shape %d", shape);
    if (!a->wf) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg: no fragment-major weights (icaf_conv_args.wf)");
    if (a->dtype == ICAF_F32 || a->out_dtype != a->dtype) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg: 16-bit types, out dtype == dtype");
    if ((a->Cin * 2) % 128) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg: Cin * 2 bytes must be a multiple of 128 (Cin = %d)", a->Cin);
    if (a->pre || a->w2) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg: no pre-activation term / chained layer");
    if (p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg: operand exceeds the 2 GiB buffer-descriptor range");
    if (a->Kp % 64) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg: Kp must be a multiple of 64");
    if (((uintptr_t)a->wf & 15) || (a->wf_gs * 2) % 16) return fail(ICAF_ERR_ARG, "igemm_wreg: wf must be 16-byte aligned");
    if ((shape == 2 || shape == 4 || shape == 5) && a->Cout <= 128) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg 128x256: Cout = %d <= 128 (use 128x128)", a->Cout);
    if (shape == 3 && a->act == ICAF_ACT_GELU) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg %s: built for SiLU / linear layers", wreg_tag(shape));
    if (shape == 3 && a->Cout <= 256) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg 128x512: Cout = %d <= 256 (use 128x256)", a->Cout);
    // the fragment-major copy covers Np = Cout rounded up to 128 channels: a wider tile must not reach beyond it
    const int bn = (shape == 1 || shape == 6) ? 128 : shape == 3 ? 512 : 256;
    if ((long long)((a->Cout + bn - 1) / bn) * bn > ((long long)a->Cout + 127) / 128 * 128)
        return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg %s: channel tiles reach beyond the packed weights (Cout = %d)", wreg_tag(shape), a->Cout);
    return ICAF_OK;
}

template <int DT, int NWV, int ACT, int TN = 1, int BM = 128>
static int launch_wreg_mode(const icaf_conv_args* a, const ConvP& p, int groups, hipStream_t s) {
    constexpr int BN = 32 * NWV * TN;
    constexpr int ring = 4 * BM * 128, stage_out = TileLds<DT, DT, BM, BN>::OUT_BYTES;
    constexpr int LDS = ring > stage_out ? ring : stage_out;
    ConvP q = p;
    q.mtiles = (p.M + BM - 1) / BM;
    q.ntiles = (p.Cout + BN - 1) / BN;
    q.nchunks = (p.K + 63) / 64;
    dim3 grid((unsigned)(q.mtiles * q.ntiles), 1, (unsigned)groups);
    const bool plain = q.kh == 1 && q.kw == 1 && q.sh == 1 && q.sw == 1 && q.ph == 0 && q.pw == 0;
    auto go = [&](auto kern) -> int {
        ICAF_LDS_OPTIN(kern, LDS);               // (one flag array per instantiation: the lambda is instantiated per kernel type)
        kern<<<grid, dim3(NWV * 64), LDS, s>>>(q, a->wf, a->wf_gs);
        ICAF_LAUNCH_CHECK();
        return ICAF_OK;
    };
    if (plain) return go(igemm_wreg_kernel<DT, NWV, ACT, 1, TN, BM>);
    return go(igemm_wreg_kernel<DT, NWV, ACT, 2, TN, BM>);
}

template <int DT, int NWV, int TN = 1, int BM = 128>
static int launch_wreg_act(const icaf_conv_args* a, const ConvP& p, int groups, hipStream_t s) {
    if (p.act == ICAF_ACT_SILU) return launch_wreg_mode<DT, NWV, ICAF_ACT_SILU, TN, BM>(a, p, groups, s);
    if constexpr (NWV == 4) { if (p.act == ICAF_ACT_GELU) return launch_wreg_mode<DT, NWV, ICAF_ACT_GELU, TN, BM>(a, p, groups, s); }
    else if (p.act == ICAF_ACT_GELU && TN == 2) return fail(ICAF_ERR_UNSUPPORTED, "igemm_wreg 128x512: built for SiLU / linear layers");
    if constexpr (NWV == 8 && TN == 1) { if (p.act == ICAF_ACT_GELU) return launch_wreg_mode<DT, NWV, ICAF_ACT_GELU, TN, BM>(a, p, groups, s); }
    return launch_wreg_mode<DT, NWV, ICAF_ACT_NONE, TN, BM>(a, p, groups, s);
}

template <int DT>
static int launch_wreg_shape(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    switch (shape) {
        case 1: return launch_wreg_act<DT, 4>(a, p, a->groups, s);
        case 2: return launch_wreg_act<DT, 8>(a, p, a->groups, s);
        case 3: return launch_wreg_act<DT, 8, 2>(a, p, a->groups, s);      // 128 x 512: eight waves x 64 channels
        case 4: return launch_wreg_act<DT, 4, 2>(a, p, a->groups, s);      // 128 x 256: four waves x 64 channels
        case 5: return launch_wreg_act<DT, 4, 2, 64>(a, p, a->groups, s);  //  64 x 256: four waves x 64 channels, half the pixels
        default: return launch_wreg_act<DT, 4, 1, 64>(a, p, a->groups, s); //  64 x 128: four waves x 32 channels, half the pixels
    }
}

int launch_wreg(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    int st = wreg_check(a, p, shape);
    if (st) return st;
    return a->dtype == ICAF_BF16 ? launch_wreg_shape<ICAF_BF16>(a, p, shape, s) : launch_wreg_shape<ICAF_F16>(a, p, shape, s);
}

}  // namespace icaf
