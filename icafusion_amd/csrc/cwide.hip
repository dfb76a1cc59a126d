// 3x3 convolutions (stride 1 or 2) of the 64- / 128-channel layers with 128 or more output channels on gfx950: halo patch RESIDENT
// in LDS, weights streamed PER WAVE from L2 straight into registers.  (C3 Bottlenecks of the P4 backbones, the head's 3x3 layers at
// 40 x 40, the down-sampling convolutions 160 -> 80 and 80 -> 40.)
//
// What bounds these layers in the implicit-GEMM kernel (igemm.hip) is not HBM and not the MFMA pipe but the feed: a 128 x 128 tile
// with K = 9 x 128 moves 590 KB through 576 LDS-DMA instructions (the pixel operand re-fetched once per filter tap — with stride 2
// neighbouring output pixels share nothing of a tap's row — and the weights through the same ring), every MFMA pair waits on two
// b128 fragment reads, and a barrier closes every 64-K slice: measured 24 % of the MFMA peak at batch 32.  Here a workgroup
// (4 wavefronts) owns an 8 x 16 or 8 x 8 patch of output pixels and 128 output channels:
//   * its input halo patch ((TH-1)S+3 x (TW-1)S+3 pixels) is fetched ONCE by LDS-DMA (out-of-image pixels zero-filled by the
//     descriptor range check), pixel-major, the 16-byte channel slots of a pixel XOR-swizzled by the entry index (ctile.hip's
//     layout); for stride 2 a row holds its even columns, then its odd columns, so a tap still reads consecutive entries for
//     consecutive output pixels;
//   * wave w owns output channels [32 w, 32 w + 32) of the block and ALL the 4 x 8 sub-tiles: one weight fragment feeds NSUB MFMAs,
//     and the fragments come from the fragment-major copy of the packed filter (icaf.h: icaf_conv_args.wf; one coalesced 16-byte
//     load per lane = one MFMA A operand), three 4-step slices ahead in registers: no weight ring, no barrier inside the K loop;
//   * the chained 1x1 (128 -> <= 128) takes its weights through the same per-wave stream;
//   * CHAIN = 2, the C3 TAIL (round 4): the block's LAST Bottleneck carries the C3's cv3 (models/common.py:226,
//     cv3(cat(m(cv1(x)), cv2(x)))): the completed m tile stays in the staging tile, the cv2 half of the same pixels (icaf.h:
//     icaf_conv_args.x2) is parked beside it, and the 1x1 over K = [m | cv2] = 256 -> 256 channels runs as two 128-channel passes
//     through the same weight ring — m is never written, cat(m, cv2) never read, one launch less per C3.
// K walks (ky, kx, cin) in igemm's order with igemm's MFMA step, and the epilogue repeats the shared epilogue's expressions (bias +
// SiLU on the fp32 accumulator, rounding to the storage type, residual added to the rounded value, chained 1x1 on the tile as
// stored): results are bit-identical to every other launch configuration of the layer (tests/test_gpu_fullsize.py).
#include "conv_common.h"

// Ablation switches for timing studies (tools/quick_variant.py -DICAF_CW_ABL=n; results are then meaningless):
//   1 = no halo-patch DMA, 2 = no weight loads inside the K loop, 4 = no LDS fragment reads, 8 = no MFMAs, 16 = no residual loads
#ifndef ICAF_CW_ABL
#define ICAF_CW_ABL 0
#endif
// 1 = the halo patch travels global -> registers -> LDS (buffer_load_dwordx4 + ds_write_b128) instead of by LDS-DMA (measured: equal or
// slower: 140 vs 131 us on the 160 -> 80 layer)
// Tiles of look-ahead of an L2 prefetch (0 = off, the default): a workgroup touches every 128-byte line of the halo patch of the tile
// CW_PF positions behind its own on the same XCD — whose workgroup starts a few microseconds later — so that tile's first touch of its
// patch, the longest wait of a one-shot workgroup (phase clocks: 9-11 k cycles), would find the lines in the XCD's L2.  Measured with
// 24 and 48 tiles of look-ahead at batch 32: no change on the 40 x 40 layers (43.7 / 44.1 vs 43.3 us), slower on the 160 -> 80 layer
// (139 / 133 vs 129 us) — the wait is not an L2 miss.
#ifndef ICAF_CW_PF
#define ICAF_CW_PF 0
#endif
#ifndef ICAF_CW_REGPATCH
#define ICAF_CW_REGPATCH 0
#endif

namespace icaf {

constexpr int CW_N = 128;                        // output channels per workgroup (4 waves x 32)
constexpr int CW_TH = 8;                         // output rows per workgroup
constexpr int CW_SO = CW_N * 2 + 16;             // staging row stride
constexpr int CW_SL = 4, CW_DEPTH = 3;           // K steps per weight slice; slices held in registers

// NSUB = 4: 8 x 16 output pixels per workgroup (four 4 x 8 sub-tiles per wave); NSUB = 2: 8 x 8 (two sub-tiles per wave: a 40 x 40
// map is covered without the half-empty right column of the 16-wide tiling and the smaller patch lets one more workgroup share the
// CU — at twice the weight traffic per pixel, which the L2 can afford here).
template <int CIN, int STR, int NSUB> struct CwTile {
    static constexpr int TW = NSUB == 4 ? 16 : 8;
    static constexpr int HH = (CW_TH - 1) * STR + 3, HWD = (TW - 1) * STR + 3;
    static constexpr int EH = STR == 2 ? TW + 1 : 0;                    // stride 2: entries of a row's even-column plane (odd: TW)
    // stride 1: rows of a 4 x 8 sub-tile 8 swizzle keys apart (256-byte entries: pitch = 8 mod 16) where the LDS allows it
    static constexpr int PITCH = STR == 2 ? ((HWD + 1) & ~1) : (NSUB == 4 ? 24 : HWD);
    static constexpr int PB = CIN * 2;                                // bytes per patch entry (pixel)
    static constexpr int LSP = CIN == 64 ? 3 : 4, SP = 1 << LSP, GSH = 4 - LSP;      // 16-byte slots per entry; swizzle key = (idx >> GSH) & (SP - 1)
    static constexpr int NENT = HH * PITCH;
    static constexpr int PATCH = (NENT * PB + 1023) / 1024 * 1024;
    static constexpr int NPX = CW_TH * TW;
    static constexpr int KSTEPS = 9 * CIN / 16, NSLICE = KSTEPS / CW_SL;
    static constexpr int LDS = PATCH > NPX * CW_SO ? PATCH : NPX * CW_SO;
    static constexpr int WG_PER_CU = (160 * 1024) / LDS >= 3 ? 3 : (160 * 1024) / LDS >= 2 ? 2 : 1;
    // C3 tail: the m tile and the cv2 tile side by side once the K loop is done
    static constexpr int LDS_TAIL = PATCH > 2 * NPX * CW_SO ? PATCH : 2 * NPX * CW_SO;
    static constexpr int WG_PER_CU_TAIL = (160 * 1024) / LDS_TAIL >= 3 ? 3 : (160 * 1024) / LDS_TAIL >= 2 ? 2 : 1;
    static_assert(CIN == 64 || CIN == 128, "entries of 128 or 256 bytes");
    static_assert(NSLICE % CW_DEPTH == 0, "the register ring rotates statically");
};

struct CwGeom { int tiles_x, tiles_y, ntile; };

// Phase clocks for timing studies (-DICAF_CW_DBG: workgroup 0 and the LAST workgroup of group 0 stamp s_memtime at their phase
// boundaries; icaf_cwide_debug_clocks copies the 2 x 16 stamps out; 8-10: inside the prologue).  Not compiled into the product library.
#ifdef ICAF_CW_DBG
__device__ long long icaf_cw_stamps[32];
#define CW_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) \
    icaf_cw_stamps[(blockIdx.x == 0 ? 0 : 16) + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define CW_STAMP(i) do {} while (0)
#endif

template <int DT, int CIN, int STR, int NSUB, int CHAIN>         // CHAIN: 0 = none, 1 = chained 1x1 over the tile, 2 = C3 tail (cv3 over [tile | x2])
__global__ __launch_bounds__(256, (CHAIN == 2 ? CwTile<CIN, STR, NSUB>::WG_PER_CU_TAIL : CwTile<CIN, STR, NSUB>::WG_PER_CU)) void cwide_kernel(const ConvP p, const CwGeom gm, const void* __restrict__ wfrag,
                                                                                    const long long wf_gs) {
    using E = Elem<DT>;
    using T = typename E::type;
    using G = CwTile<CIN, STR, NSUB>;
    constexpr int S = STR;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int VEC = E::VEC;
    constexpr int PITCH = G::PITCH, LSP = G::LSP, SP = G::SP, GSH = G::GSH, KSTEPS = G::KSTEPS, NSLICE = G::NSLICE;
    constexpr int KPT = CIN / 16;                                    // MFMA steps per filter tap
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* patch = lds;
    unsigned char* stg = lds;                                        // the output tile is staged over the patch once the K loop is done
    unsigned char* stgb = lds + G::NPX * CW_SO;                      // C3 tail: the cv2 half of cv3's input, then the second half of its output

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // = 32-channel group of the block
    const int g = blockIdx.z, n0 = blockIdx.y * CW_N;                // modality / first output channel of the block
    const int tile = xcd_tile(gm.ntile);
    const int per_img = gm.tiles_x * gm.tiles_y;
    const int b = tile / per_img, tr = tile - b * per_img, ty = tr / gm.tiles_x;
    const int y0 = ty * CW_TH, x0 = (tr - ty * gm.tiles_x) * G::TW;  // output-pixel origin

    CW_STAMP(0);
    // ---- the wave's weight stream: NSLICE slices of the 3x3 filter, then (CHAIN) 2 slices of the chained 1x1 ---------------------
    const u32x4* wf = (const u32x4*)((const T*)wfrag + g * wf_gs) + (long long)(blockIdx.y * 4 + wave) * KSTEPS * 64 + lane;
    u32x4 wq[CW_DEPTH][CW_SL];
#pragma unroll
    for (int u = 0; u < CW_DEPTH; ++u)
#pragma unroll
        for (int k = 0; k < CW_SL; ++k) wq[u][k] = wf[(u * CW_SL + k) * 64];

    CW_STAMP(8);
    // ---- halo patch -> LDS (each DMA instruction fills 64 consecutive 16-byte slots) ------------------------------------------------
    {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
        constexpr unsigned OOB = 0x80000000u;
        const unsigned img_off = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES;
        const int gy0 = y0 * S - 1, gx0 = x0 * S - 1;
        constexpr int NPI = (G::PATCH / 1024 + 3) / 4;
        u32x4 pv[ICAF_CW_REGPATCH ? NPI : 1];
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
            const int j = wave + 4 * i;
            if (j >= G::PATCH / 1024) break;
            const int L = (j << 6) + lane, idx = L >> LSP;
            const int cs = (L & (SP - 1)) ^ ((idx >> GSH) & (SP - 1));
            const int hy = idx / PITCH, rem = idx - hy * PITCH;
            int hx;
            if constexpr (S == 2) hx = rem < G::EH ? 2 * rem : 2 * (rem - G::EH) + 1;
            else hx = rem;
            const int gy = gy0 + hy, gx = gx0 + hx;
            const bool ok = hy < G::HH && hx < G::HWD && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned voff = ok ? img_off + (unsigned)((gy * p.W + gx) * p.ldx) * E::BYTES + (unsigned)(cs << 4) : OOB;
            if constexpr (ICAF_CW_REGPATCH) pv[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, voff, 0, 0);       // (out of range: zeros)
            else if constexpr (!(ICAF_CW_ABL & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(patch + (j << 10)), 16, voff, 0, 0, 0);
        }
        if constexpr (ICAF_CW_REGPATCH) {
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                const int j = wave + 4 * i;
                if (j >= G::PATCH / 1024) break;
                *(u32x4*)(patch + (j << 10) + (lane << 4)) = pv[i];
            }
        }
    }

    CW_STAMP(9);
    // per-lane constants
    T* __restrict__ yg = (T*)p.y + g * p.y_gs;
    const T* __restrict__ rg = p.res ? (const T*)p.res + g * p.res_gs : nullptr;
    const float alpha_acc = p.alpha_acc[g], alpha_res = p.alpha_res[g];
    // staging row r (sub-tile major: sub-tile = 4 rows x 8 columns) -> output pixel index, or -1 outside the tensor
    auto row_to_m = [&](int r) {
        const int st = r >> 5, q = r & 31;
        const int gy = y0 + (NSUB == 4 ? (st >> 1) : st) * 4 + (q >> 3), gx = x0 + (NSUB == 4 ? (st & 1) * 8 : 0) + (q & 7);
        return (gy < p.Ho && gx < p.Wo) ? (b * p.Ho + gy) * p.Wo + gx : -1;
    };
    CW_STAMP(10);
    // this lane's pixel of each sub-tile: row l31 >> 3, column l31 & 7; patch entry of tap (0, 0)
    int lbase[NSUB];
#pragma unroll
    for (int bb = 0; bb < NSUB; ++bb)
        lbase[bb] = ((NSUB == 4 ? (bb >> 1) : bb) * 4 + (l31 >> 3)) * S * PITCH + (NSUB == 4 ? (bb & 1) * 8 : 0) + (l31 & 7);

    // ---- L2 prefetch of a later tile's patch: one dword per 128-byte line, the values are never used (consumed by an empty asm at the
    //      end of the kernel); PFW loads per wave, the YOUNGEST operations of the prologue: the counted wait below leaves them in flight
    constexpr int PF_LR = (G::HWD * G::PB + 127) / 128, PF_LINES = G::HH * PF_LR, PFW = ICAF_CW_PF ? (PF_LINES + 255) / 256 : 0;
    unsigned pfsink = 0;
    if constexpr (ICAF_CW_PF > 0) {
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
        const int q8 = gm.ntile >> 3, r8 = gm.ntile & 7, xcd = blockIdx.x & 7, mine = q8 + (xcd < r8 ? 1 : 0);
        const bool pf_ok = (int)(blockIdx.x >> 3) + ICAF_CW_PF < mine;             // the later tile is one of this XCD's (xcd_tile's walk)
        const int tp = tile + ICAF_CW_PF;
        const int bp = tp / per_img, trp = tp - bp * per_img, typ = trp / gm.tiles_x;
        const int gy0 = typ * CW_TH * S - 1, gxb = ((trp - typ * gm.tiles_x) * G::TW * S - 1) * G::PB;      // first row / first byte within an image row
        const unsigned img_off = (unsigned)bp * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES;
        const int row_bytes = p.W * p.ldx * E::BYTES;
#pragma unroll
        for (int i = 0; i < PFW; ++i) {
            const int n = (wave + 4 * i) * 64 + lane, row = n / PF_LR, off = gxb + (n - row * PF_LR) * 128;
            const int gy = gy0 + row;
            const bool ok = pf_ok && n < PF_LINES && (unsigned)gy < (unsigned)p.H && off >= 0 && off < row_bytes;
            pfsink ^= __builtin_amdgcn_raw_buffer_load_b32(xr, ok ? img_off + (unsigned)(gy * row_bytes + off) : 0x80000000u, 0, 0);
        }
    }

    CW_STAMP(1);
    wait_vmcnt<PFW>();                             // patch (this wave's share), residual vectors, biases, first weight slices: all but the prefetch loads
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // (not __syncthreads(): it would drain the prefetch loads as well)
    CW_STAMP(2);
    // (the residual vectors and the biases are requested only now: in the prologue their 16 vector-memory instructions per wave sat in
    //  front of the barrier — phase clocks: 3.3-4.2 k cycles of issue — although nothing needs them before the epilogue)
    constexpr int NIT = G::NPX * (CW_N / VEC) / 256;                 // 16-byte vectors of the tile per thread (8 / 4)
    u32x4 rres[NIT];
    if (rg && !(ICAF_CW_ABL & 16)) {                                 // the residual vectors of this thread's flush positions: in flight during the K loop
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, row = idx >> 4, cv = idx & 15;
            const int m = row_to_m(row);
            rres[it] = *(const u32x4*)(rg + (long long)(m < 0 ? 0 : m) * p.ldr + n0 + cv * VEC);
        }
    }
    u32x4 xcat[CHAIN == 2 ? NIT : 1];                                // C3 tail: the cv2 vectors of the same positions, likewise in flight
    if constexpr (CHAIN == 2) {
        const T* __restrict__ x2g = (const T*)p.x2 + g * p.x2_gs;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, row = idx >> 4, cv = idx & 15;
            const int m = row_to_m(row);
            xcat[it] = *(const u32x4*)(x2g + (long long)(m < 0 ? 0 : m) * p.ldx2 + cv * VEC);
        }
    }
    f32x4 bq[4], bq2[4];
    const float* __restrict__ bias2 = (CHAIN && p.bias2) ? p.bias2 + g * p.bias2_gs : nullptr;
    {
        const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int n = wave * 32 + 8 * qd + 4 * hi;
            bq[qd] = bias ? *(const f32x4*)(bias + n0 + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (CHAIN != 2) bq2[qd] = bias2 ? *(const f32x4*)(bias2 + n) : f32x4{0.f, 0.f, 0.f, 0.f};      // (the tail asks for its 2 x 4 behind the K loop)
        }
    }

    // ---- K loop: KSTEPS MFMA steps x NSUB sub-tiles, weights from the register stream ----------------------------------------------
    f32x16 acc[NSUB];
#pragma unroll
    for (int bb = 0; bb < NSUB; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[bb][r] = 0.0f;
    const T* w2f = nullptr;
    if constexpr (CHAIN) w2f = (const T*)p.w2 + g * p.w2_gs;
    // slice c of the chained layer's weights — row-major packed [Np2][Kp2], lane (hi, r) of step ks2 reads w2[row + r][16 ks2 + 8 hi .. + 8].
    // CHAIN 1: 2 slices (K = 128); tail: 8 slices, slice c = steps 4 (c >> 1) .. + 4 of channel pass c & 1 (K = 256, two 128-channel passes)
    auto chain_slice = [&](int c, u32x4 (&dst)[CW_SL]) {
        const int row = (CHAIN == 2 ? (c & 1) * CW_N : 0) + wave * 32 + l31, k0 = (CHAIN == 2 ? (c >> 1) : c) * CW_SL;
#pragma unroll
        for (int k = 0; k < CW_SL; ++k) dst[k] = *(const u32x4*)(w2f + (long long)row * p.Kp2 + (k0 + k) * 16 + hi * 8);
    };
#pragma unroll
    for (int sl = 0; sl < NSLICE; ++sl) {
        const int u = sl % CW_DEPTH;
#pragma unroll
        for (int k = 0; k < CW_SL; ++k) {
            const int ks = sl * CW_SL + k, tap = ks / KPT, s = ks - tap * KPT, ky = tap / 3, kx = tap - 3 * ky;
            const int toff = ky * PITCH + (S == 2 ? (kx & 1) * G::EH + (kx >> 1) : kx);
            u32x4 fp[NSUB];
#pragma unroll
            for (int bb = 0; bb < NSUB; ++bb) {
                const int idx = lbase[bb] + toff;
                if constexpr (ICAF_CW_ABL & 4) fp[bb] = u32x4{(unsigned)idx, (unsigned)ks, 0x3f803f80u, 0x3f803f80u};
                else fp[bb] = *(const u32x4*)(patch + ((((idx << LSP) + (((2 * s + hi) ^ (idx >> GSH)) & (SP - 1)))) << 4));
            }
#pragma unroll
            for (int bb = 0; bb < NSUB; ++bb) {
                if constexpr (ICAF_CW_ABL & 8) acc[bb][0] += __uint_as_float(wq[u][k][0] ^ fp[bb][0]);
                else mma_step<DT>(acc[bb], wq[u][k], fp[bb]);
            }
        }
        // refill the slot: slice sl + 3 of the filter, or (CHAIN) the chained 1x1's fragments behind it — row-major packed
        // [Np2][Kp2], lane (hi, r) of step ks2 reads w2[32 wave + r][16 ks2 + 8 hi .. + 8]
        if (sl + CW_DEPTH < NSLICE && !(ICAF_CW_ABL & 2)) {
#pragma unroll
            for (int k = 0; k < CW_SL; ++k) wq[u][k] = wf[((sl + CW_DEPTH) * CW_SL + k) * 64];
        } else if constexpr (CHAIN != 0) {
            const int c2 = sl + CW_DEPTH - NSLICE;                   // 0 .. 2: the chained 1x1 has 8 steps = 2 slices, the C3 tail 8 slices
            if (c2 >= 0 && c2 < (CHAIN == 2 ? CW_DEPTH : 2)) chain_slice(c2, wq[u]);      // (NSLICE % CW_DEPTH == 0: chain slice c sits in slot c % CW_DEPTH)
        }
        __builtin_amdgcn_sched_barrier(0);         // (left alone the scheduler hoists the fragment reads of several slices: registers)
    }
    CW_STAMP(3);
    __syncthreads();                               // every wave has left the K loop: the patch may be overwritten by the staged tile
    CW_STAMP(4);

    auto stage = [&](unsigned char* dst, const f32x16 (&a)[NSUB], const f32x4 (&bv)[4], float scale) {
#pragma unroll
        for (int bb = 0; bb < NSUB; ++bb)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int nl = wave * 32 + 8 * qd + 4 * hi;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = a[bb][4 * qd + j] + bv[qd][j] + 0.0f;
                silu4_f(v, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= scale;
                u32x2 pk;
                if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                *(u32x2*)(dst + (bb * 32 + l31) * CW_SO + nl * E::BYTES) = pk;
            }
    };
    stage(stg, acc, bq, alpha_acc);
    if constexpr (CHAIN == 2) {                    // the cv2 half of cv3's pixel operand beside the tile
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, row = idx >> 4, cv = idx & 15;
            *(u32x4*)(stgb + row * CW_SO + cv * 16) = xcat[it];
        }
    }
    __syncthreads();
    CW_STAMP(5);
    if constexpr (CHAIN == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, row = idx >> 4, cv = idx & 15;
            const int m = row_to_m(row), n = n0 + cv * VEC;
            if (m >= 0 && n < p.Cout) {
                u32x4 sv = *(const u32x4*)(stg + row * CW_SO + cv * 16);
                if (rg) {                          // the shared epilogue's arithmetic: staged value + alpha_res * residual
                    float v[VEC], r[VEC];
                    unpack16<DT>(sv, v);
                    unpack16<DT>(rres[it], r);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                    sv = pack16<DT>(v);
                }
                *(u32x4*)(yg + (long long)m * p.ldy + n) = sv;
            }
        }
        CW_STAMP(6);
    } else {
        // chained layer (one channel block: n0 = 0): y is completed now — staged vector + alpha_res * residual, written to y when the
        // chain keeps it and BACK into the staging tile, which the chained 1x1 consumes as stored (igemm's CHAIN + WB)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256, row = idx >> 4, cv = idx & 15;
            const int m = row_to_m(row), n = cv * VEC;
            if (m >= 0 && n < p.Cout) {
                u32x4 sv = *(const u32x4*)(stg + row * CW_SO + cv * 16);
                if (rg) {
                    float v[VEC], r[VEC];
                    unpack16<DT>(sv, v);
                    unpack16<DT>(rres[it], r);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                    sv = pack16<DT>(v);
                    *(u32x4*)(stg + row * CW_SO + cv * 16) = sv;
                }
                if (p.keep1) *(u32x4*)(yg + (long long)m * p.ldy + n) = sv;
            }
        }
        __syncthreads();                           // the completed tile is visible
        T* __restrict__ y2g = (T*)p.y2 + g * p.y2_gs;
        if constexpr (CHAIN == 1) {
            f32x16 acc2[NSUB];
#pragma unroll
            for (int bb = 0; bb < NSUB; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[bb][r] = 0.0f;
#pragma unroll
            for (int ks2 = 0; ks2 < CW_N / 16; ++ks2) {                  // K = 128 channels of the tile: eight MFMA steps
                // (slice c2 = ks2 / 4 of the chained weights sits in ring slot (NSLICE + c2) % CW_DEPTH — the refill order above)
                const int u = (NSLICE + ks2 / CW_SL) % CW_DEPTH, k = ks2 % CW_SL;
#pragma unroll
                for (int bb = 0; bb < NSUB; ++bb) {
                    const u32x4 fp2 = *(const u32x4*)(stg + (bb * 32 + l31) * CW_SO + ((2 * ks2 + hi) << 4));
                    mma_step<DT>(acc2[bb], wq[u][k], fp2);
                }
            }
            __syncthreads();                           // the tile has been consumed
            stage(stg, acc2, bq2, 1.0f);
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * 256, row = idx >> 4, cv = idx & 15;
                const int m = row_to_m(row), n = cv * VEC;
                if (m >= 0 && n < p.Cout2) *(u32x4*)(y2g + (long long)m * p.ldy2 + n) = *(const u32x4*)(stg + row * CW_SO + cv * 16);
            }
        } else {
            // C3 tail: y2 = SiLU(W3 . [m | cv2] + bias3), K = 256 in cv3's own order (sixteen MFMA steps: eight over the tile, eight over
            // the cv2 half), two passes of 128 output channels with their own accumulators; the weights come through the ring as in the
            // K loop (slice c in slot c % 3, refilled three slices ahead), one slice = four steps of one pass
            f32x16 acc2[2][NSUB];
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
                for (int bb = 0; bb < NSUB; ++bb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[pz][bb][r] = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int u = c % CW_DEPTH, pz = c & 1;
#pragma unroll
                for (int k = 0; k < CW_SL; ++k) {
                    const int ks2 = (c >> 1) * CW_SL + k;
                    const unsigned char* src = ks2 < 8 ? stg : stgb;
                    u32x4 fp2[NSUB];
#pragma unroll
                    for (int bb = 0; bb < NSUB; ++bb) fp2[bb] = *(const u32x4*)(src + (bb * 32 + l31) * CW_SO + ((2 * (ks2 & 7) + hi) << 4));
#pragma unroll
                    for (int bb = 0; bb < NSUB; ++bb) mma_step<DT>(acc2[pz][bb], wq[u][k], fp2[bb]);
                }
                if (c + CW_DEPTH < 8) chain_slice(c + CW_DEPTH, wq[u]);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x4 bz[2][4];                            // (asked for only now: 32 registers the 8 x 8 form does not have during the GEMM)
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    bz[pz][qd] = bias2 ? *(const f32x4*)(bias2 + pz * CW_N + wave * 32 + 8 * qd + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();                           // both tiles have been consumed
            stage(stg, acc2[0], bz[0], 1.0f);
            stage(stgb, acc2[1], bz[1], 1.0f);
            __syncthreads();
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int idx = tid + it * 256, row = idx >> 4, cv = idx & 15;
                    const int m = row_to_m(row), n = pz * CW_N + cv * VEC;
                    if (m >= 0 && n < p.Cout2) *(u32x4*)(y2g + (long long)m * p.ldy2 + n) = *(const u32x4*)((pz ? stgb : stg) + row * CW_SO + cv * 16);
                }
        }
    }
    if constexpr (ICAF_CW_PF > 0) asm volatile("" ::"v"(pfsink));
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct CwShape { int cin, s, nsub; const char* tag; };
static const CwShape kCw[] = {{128, 1, 4, "8x16n128"}, {128, 1, 2, "8x8n128"}, {64, 2, 4, "8x16n128s2c64"}, {128, 2, 2, "8x8n128s2"}, {64, 2, 2, "8x8n128s2c64"}};
constexpr int CW_NSHAPES = 5;

const char* cwide_tag(int shape) { return (shape >= 1 && shape <= CW_NSHAPES) ? kCw[shape - 1].tag : "?"; }

int cwide_check(const icaf_conv_args* a, const ConvP& p, int shape) {
    if (shape < 1 || shape > CW_NSHAPES) return fail(ICAF_ERR_ARG, "cwide: unknown shape %d", shape);
    const CwShape& sh = kCw[shape - 1];
    if (a->dtype == ICAF_F32 || a->out_dtype != a->dtype) return fail(ICAF_ERR_UNSUPPORTED, "cwide: 16-bit types, out dtype == dtype");
    if (a->kh != 3 || a->kw != 3 || a->sh != sh.s || a->sw != sh.s || a->ph != 1 || a->pw != 1) return fail(ICAF_ERR_UNSUPPORTED, "cwide %s: 3x3 / stride %d / pad 1 layers", sh.tag, sh.s);
    if (a->Cin != sh.cin || a->Cout % CW_N || a->Kp != 9 * sh.cin) return fail(ICAF_ERR_UNSUPPORTED, "cwide %s: built for %d -> (multiples of 128) channels (Cin = %d, Cout = %d, Kp = %d)", sh.tag, sh.cin, a->Cin, a->Cout, a->Kp);
    if (a->act != ICAF_ACT_SILU || a->pre) return fail(ICAF_ERR_UNSUPPORTED, "cwide: SiLU layers without a pre-activation term");
    if (!a->wf) return fail(ICAF_ERR_UNSUPPORTED, "cwide: needs the fragment-major weight copy (icaf_conv_args.wf)");
    if (p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "cwide: operand exceeds the 2 GiB buffer-descriptor range");
    if (!p.vec_y || (a->res && !p.vec_r)) return fail(ICAF_ERR_UNSUPPORTED, "cwide: y / res must take 16-byte vectors");
    if (a->x2) {          // C3 tail: cv3 over [this layer's tile | x2]
        if (!a->w2) return fail(ICAF_ERR_ARG, "cwide: x2 without the chained layer's weights (w2)");
        if (sh.s != 1 || sh.cin != 128) return fail(ICAF_ERR_UNSUPPORTED, "cwide %s: the C3 tail is built for the stride-1 128 -> 128 shapes (tile ids 81 / 82)", sh.tag);
        if (a->Cout != CW_N || a->Cout2 != 2 * CW_N || a->Kp2 != 2 * CW_N || !p.vec_y2)
            return fail(ICAF_ERR_UNSUPPORTED, "cwide: the C3 tail is a 1x1 of [128 | 128] -> 256 channels with Kp2 = 256 (Cout = %d, Cout2 = %d, Kp2 = %d)", a->Cout, a->Cout2, a->Kp2);
        if (a->ldx2 < CW_N || a->ldx2 % 8 || ((uintptr_t)a->x2 & 15) || (a->x2_gs * 2) % 16) return fail(ICAF_ERR_ARG, "cwide: x2 must take 16-byte vectors (ldx2 = %d)", a->ldx2);
        if (a->chain_keep) return fail(ICAF_ERR_UNSUPPORTED, "cwide: the C3 tail does not keep the intermediate tensor");
    } else if (a->w2) {
        if (a->Cout != CW_N || a->Cout2 > CW_N || a->Cout2 % 32 || a->Kp2 != CW_N || !p.vec_y2) return fail(ICAF_ERR_UNSUPPORTED, "cwide: chained 1x1 of 128 -> (32, 64, 96 or 128) channels with Kp2 = 128");
        if (a->res && !a->chain_keep) return fail(ICAF_ERR_UNSUPPORTED, "cwide: a residual needs chain_keep");
        if (a->chain_keep && (a->alpha_acc[0] != 1.0f || a->alpha_acc[1] != 1.0f)) return fail(ICAF_ERR_UNSUPPORTED, "cwide: chain_keep with alpha_acc != 1");
    }
    return ICAF_OK;
}

template <int DT, int CIN, int STR, int NSUB, int CHAIN>
static int launch_cwide_cfg(const icaf_conv_args* a, const ConvP& p, hipStream_t s) {
    using G = CwTile<CIN, STR, NSUB>;
    CwGeom gm;
    gm.tiles_x = (p.Wo + G::TW - 1) / G::TW;
    gm.tiles_y = (p.Ho + CW_TH - 1) / CW_TH;
    gm.ntile = p.B * gm.tiles_x * gm.tiles_y;
    constexpr int lds = CHAIN == 2 ? G::LDS_TAIL : G::LDS;
    ICAF_LDS_OPTIN((cwide_kernel<DT, CIN, STR, NSUB, CHAIN>), lds);
    cwide_kernel<DT, CIN, STR, NSUB, CHAIN><<<dim3((unsigned)gm.ntile, (unsigned)(a->Cout / CW_N), (unsigned)a->groups), dim3(256), lds, s>>>(p, gm, a->wf, a->wf_gs);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <int DT, int CIN, int STR, int NSUB>
static int launch_cwide_ch(const icaf_conv_args* a, const ConvP& p, hipStream_t s) {
    if constexpr (CIN == 128 && STR == 1) { if (a->x2) return launch_cwide_cfg<DT, CIN, STR, NSUB, 2>(a, p, s); }      // (cwide_check let x2 through for these shapes only)
    return a->w2 ? launch_cwide_cfg<DT, CIN, STR, NSUB, 1>(a, p, s) : launch_cwide_cfg<DT, CIN, STR, NSUB, 0>(a, p, s);
}

template <int DT>
static int launch_cwide_dt(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    switch (shape) {
        case 1: return launch_cwide_ch<DT, 128, 1, 4>(a, p, s);
        case 2: return launch_cwide_ch<DT, 128, 1, 2>(a, p, s);
        case 3: return launch_cwide_ch<DT, 64, 2, 4>(a, p, s);
        case 4: return launch_cwide_ch<DT, 128, 2, 2>(a, p, s);
        default: return launch_cwide_ch<DT, 64, 2, 2>(a, p, s);
    }
}

// shapes: see kCw (tile id 80 + shape)
int launch_cwide(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    int st = cwide_check(a, p, shape);
    if (st) return st;
    return a->dtype == ICAF_BF16 ? launch_cwide_dt<ICAF_BF16>(a, p, shape, s) : launch_cwide_dt<ICAF_F16>(a, p, shape, s);
}

#ifdef ICAF_CW_DBG
}  // namespace icaf
extern "C" __attribute__((visibility("default"))) int icaf_cwide_debug_clocks(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(icaf::icaf_cw_stamps), 32 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
namespace icaf {
#endif
}  // namespace icaf
