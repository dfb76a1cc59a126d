// Persistent form of cwide.hip (3x3 convolutions, stride 1 or 2, halo patch resident in LDS, weights streamed per wave into
// registers) for gfx950 — one workgroup of EIGHT wavefronts per CU walks its XCD's share of the (image, tile) list.
//
// cwide.hip's workgroups are one-shot: each waits for its halo patch (an HBM round trip under load), computes for a few
// microseconds and stores; with two or three of them per CU the MFMA pipe idles through most of those waits (ablation at batch 32:
// removing ALL the MFMAs of a 128-channel 40 x 40 layer took 44 -> 36 us, removing the patch DMA 44 -> 40: neither the math nor any
// single feed is the bound, the unoverlapped prologue / epilogue chain is).  Here:
//   * TWO patch buffers: the DMA of tile t + 1 is issued before the K loop of tile t and lands during it; a tile's output is staged
//     over its own (consumed) patch, and its stores are in flight during the next tile — nothing of a tile is waited for except by
//     one counted `s_waitcnt vmcnt` that leaves the youngest operations (those stores, the weight slices already requested for the
//     next tile) outstanding;
//   * a wave = (32-channel group, pixel half): 4 groups x 2 halves (128 output channels per workgroup) or 8 x 1 (256: the 128 -> 256
//     down-sampling layer then fetches its patch once, not once per channel block);
//   * the weight stream of a wave is CYCLIC — the same fragments for every tile — so the register ring simply keeps rolling from
//     one tile into the next; the chained 1x1's eight fragments and the biases stay in registers for the workgroup's life.
// Patch layout, K order, MFMA step and epilogue expressions are cwide.hip's (= igemm's): bit-identical results (tested).
#include "conv_common.h"

namespace icaf {

constexpr int CP_TH = 8;                         // output rows per tile
constexpr int CP_SL = 4, CP_DEPTH = 3;           // K steps per weight slice; slices held in registers

// TWS = sub-tile columns (tile = 8 x 8 TWS output pixels, 2 TWS sub-tiles of 4 x 8), NCG = 32-channel groups per workgroup,
// NW = wavefronts per workgroup: 8 (one workgroup per CU) or 4 (small tiles: two or three workgroups per CU, each with its own pair
// of patch buffers — their K loops, epilogues and waits interleave, which a single workgroup's barriers rule out)
template <int CIN, int STR, int TWS, int NCG, int NW> struct CpTile {
    static constexpr int THREADS = 64 * NW;
    static constexpr int TW = 8 * TWS, NST = 2 * TWS, NPH = NW / NCG, NSUB = NST / NPH;
    static constexpr int N = 32 * NCG;                                // output channels per workgroup
    static constexpr int HH = (CP_TH - 1) * STR + 3, HWD = (TW - 1) * STR + 3;
    static constexpr int EH = STR == 2 ? TW + 1 : 0;
    static constexpr int PITCH = STR == 2 ? ((HWD + 1) & ~1) : (TWS == 2 ? 24 : HWD);
    static constexpr int PB = CIN * 2;
    static constexpr int LSP = CIN == 64 ? 3 : 4, SP = 1 << LSP, GSH = 4 - LSP;
    static constexpr int NENT = HH * PITCH;
    static constexpr int PATCH = (NENT * PB + 1023) / 1024 * 1024;
    static constexpr int NPX = CP_TH * TW;
    static constexpr int SO = N * 2 + 16;                             // staging row stride
    static constexpr int KSTEPS = 9 * CIN / 16, NSLICE = KSTEPS / CP_SL;
    static constexpr int BUF = PATCH > NPX * SO ? PATCH : NPX * SO;   // one buffer: patch, later the staged tile
    static constexpr bool W2LDS = CIN == 128 && STR == 1 && NW == 8;  // chained 1x1's fragments in LDS (where it has room) instead of 32 registers
    static constexpr int LDS = 2 * BUF + N * 4;                       // + the chained layer's bias table (+ N * N * 2 with W2LDS and a chain)
    static_assert(CIN == 64 || CIN == 128, "entries of 128 or 256 bytes");
    static_assert(NSLICE % CP_DEPTH == 0, "the register ring rotates statically and rolls over from tile to tile");
    static constexpr int WGPC = (160 * 1024) / (LDS + 1024) >= 2 ? 2 : 1;     // workgroups per CU (NW = 4): two — a third would fit the LDS of the 8 x 8 tiles but not the registers
    static_assert(NSUB >= 1 && NPH * NSUB == NST && LDS <= 160 * 1024, "tile shape");
};

struct CpGeom { int tiles_x, tiles_y, ntile; };

template <int DT, int CIN, int STR, int TWS, int NCG, int NW, bool CHAIN>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : CpTile<CIN, STR, TWS, NCG, NW>::WGPC)) void cwpers_kernel(const ConvP p, const CpGeom gm, const void* __restrict__ wfrag, const long long wf_gs) {
    using E = Elem<DT>;
    using T = typename E::type;
    using G = CpTile<CIN, STR, TWS, NCG, NW>;
    constexpr int CP_THREADS = G::THREADS;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int S = STR, VEC = E::VEC, NSUB = G::NSUB, N = G::N, SO = G::SO;
    constexpr int PITCH = G::PITCH, LSP = G::LSP, SP = G::SP, GSH = G::GSH, KSTEPS = G::KSTEPS, NSLICE = G::NSLICE;
    constexpr int KPT = CIN / 16;                                    // MFMA steps per filter tap
    constexpr int VPR = N / VEC;                                     // 16-byte vectors per staged row
    constexpr int NIT = G::NPX * VPR / CP_THREADS;                   // vectors of the tile per thread
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % NCG, ph = wave / NCG;                      // 32-channel group / pixel half
    const int g = blockIdx.z, n0 = blockIdx.y * N;

    // ---- the wave's weight stream (cyclic over the tiles), the chained 1x1's fragments, the biases ------------------------------
    const u32x4* wf = (const u32x4*)((const T*)wfrag + g * wf_gs) + (long long)(blockIdx.y * NCG + cg) * KSTEPS * 64 + lane;
    u32x4 wq[CP_DEPTH][CP_SL];
#pragma unroll
    for (int u = 0; u < CP_DEPTH; ++u)
#pragma unroll
        for (int k = 0; k < CP_SL; ++k) wq[u][k] = wf[(u * CP_SL + k) * 64];
    constexpr bool W2L = CHAIN && G::W2LDS;
    // The chained 1x1's eight fragments per wave (row-major packed [Np2][Kp2]: lane (hi, r) of step ks2 reads
    // w2[32 cg + r][16 ks2 + 8 hi .. + 8]): in LDS for the workgroup's life where it has room (W2L), otherwise fetched again behind
    // every tile's K loop — held through it they cost 32 registers and a spill.
    unsigned char* w2l = lds + 2 * G::BUF + N * 4;                   // W2L: [N / 16 steps][NCG groups][64 lanes][16 bytes]
    const T* w2lane = CHAIN ? (const T*)p.w2 + g * p.w2_gs + (long long)(cg * 32 + l31) * p.Kp2 + hi * 8 : nullptr;
    if constexpr (W2L) {
#pragma unroll
        for (int ks2 = 0; ks2 < N / 16; ++ks2)
            *(u32x4*)(w2l + ((ks2 * NCG + cg) * 64 + lane) * 16) = *(const u32x4*)(w2lane + ks2 * 16);      // (both pixel halves write the same bytes)
    }
    f32x4 bq[4];
    float* b2s = (float*)(lds + 2 * G::BUF);                         // chained layer's bias: LDS (16 registers fewer through the K loop)
    {
        const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int n = cg * 32 + 8 * qd + 4 * hi;
            bq[qd] = bias ? *(const f32x4*)(bias + n0 + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (CHAIN) {
            if (tid < N) b2s[tid] = p.bias2 ? (p.bias2 + g * p.bias2_gs)[tid] : 0.0f;      // (visible after the first tile's barriers)
        }
    }

    // ---- tile walk: XCD x owns the x-th contiguous eighth of the (image, tile row, tile column) list ----------------------------
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, wgx = gridDim.x >> 3;
    const int per_xcd = (gm.ntile + 7) >> 3, t_lo = xcd * per_xcd, t_hi = min(t_lo + per_xcd, gm.ntile);
    const int per_img = gm.tiles_x * gm.tiles_y;
    auto decode = [&](int t, int& b, int& y0, int& x0) {
        b = t / per_img;
        const int r = t - b * per_img, ty = r / gm.tiles_x;
        y0 = ty * CP_TH;
        x0 = (r - ty * gm.tiles_x) * G::TW;
    };
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
    auto issue_patch = [&](int t, unsigned char* dst) {
        constexpr unsigned OOB = 0x80000000u;
        int b, y0, x0;
        decode(t, b, y0, x0);
        const unsigned img_off = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES;
        const int gy0 = y0 * S - 1, gx0 = x0 * S - 1;
        int ln = lane;
        asm volatile("" : "+v"(ln));               // (keeps the per-lane entry decomposition of every DMA instruction from being hoisted out of the tile loop)
        for (int j = wave; j < G::PATCH / 1024; j += CP_THREADS / 64) {
            const int L = (j << 6) + ln, idx = L >> LSP;
            const int cs = (L & (SP - 1)) ^ ((idx >> GSH) & (SP - 1));
            const int hy = idx / PITCH, rem = idx - hy * PITCH;
            int hx;
            if constexpr (S == 2) hx = rem < G::EH ? 2 * rem : 2 * (rem - G::EH) + 1;
            else hx = rem;
            const int gy = gy0 + hy, gx = gx0 + hx;
            const bool ok = hy < G::HH && hx < G::HWD && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned voff = ok ? img_off + (unsigned)((gy * p.W + gx) * p.ldx) * E::BYTES + (unsigned)(cs << 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(dst + (j << 10)), 16, voff, 0, 0, 0);
        }
    };

    T* __restrict__ yg = (T*)p.y + g * p.y_gs;
    const T* __restrict__ rg = p.res ? (const T*)p.res + g * p.res_gs : nullptr;
    T* __restrict__ y2g = CHAIN ? (T*)p.y2 + g * p.y2_gs : nullptr;
    const float alpha_acc = p.alpha_acc[g], alpha_res = p.alpha_res[g];
    // staging row r (sub-tile major; sub-tile st = 4 rows x 8 columns at (st / TWS, st % TWS)) -> output pixel index, or -1
    auto row_to_m = [&](int r, int b, int y0, int x0) {
        const int st = r >> 5, q = r & 31;
        const int gy = y0 + (st / TWS) * 4 + (q >> 3), gx = x0 + (st % TWS) * 8 + (q & 7);
        return (gy < p.Ho && gx < p.Wo) ? (b * p.Ho + gy) * p.Wo + gx : -1;
    };
    // this lane's pixel of each of its sub-tiles: row l31 >> 3, column l31 & 7; patch entry of tap (0, 0)
    int lbase[NSUB];
#pragma unroll
    for (int bb = 0; bb < NSUB; ++bb) {
        const int st = ph * NSUB + bb;
        lbase[bb] = ((st / TWS) * 4 + (l31 >> 3)) * S * PITCH + (st % TWS) * 8 + (l31 & 7);
    }

    int t = t_lo + lb;
    if (t >= t_hi) return;                                           // (workgroup-uniform)
    issue_patch(t, lds);
    wait_vmcnt<0>();                                                 // first patch, weights, biases: the only full drain
    int cur = 0;
    while (true) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const int tn = t + wgx;
        unsigned char* patch = lds + cur * G::BUF;
        unsigned char* stg = patch;
        // Patch t was requested a whole tile ago; every vector-memory operation issued after it except the last 12 — the previous
        // tile's stores and up to two of the three weight slices requested for this tile — must have completed before it counts as
        // landed (operations complete in order).  The weight registers have their own, compiler-placed waits.
        wait_vmcnt<12>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // patch t complete; every wave is done with the other buffer (previous tile's flush)
        if (tn < t_hi) issue_patch(tn, lds + (cur ^ 1) * G::BUF);    // the next patch travels during everything below
        u32x4 rres[NIT];
        if (rg) {                                  // the residual vectors of this thread's flush positions: in flight during the K loop
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * CP_THREADS, row = idx / VPR, cv = idx - row * VPR;
                const int m = row_to_m(row, b, y0, x0);
                rres[it] = *(const u32x4*)(rg + (long long)(m < 0 ? 0 : m) * p.ldr + n0 + cv * VEC);
            }
        }
        // (the fragment addresses depend on lane constants only: left visible, the compiler hoists all KSTEPS x NSUB of them out of the
        //  tile loop and spills them — 180 to 956 bytes of scratch per lane)
        int lbv[NSUB];
#pragma unroll
        for (int bb = 0; bb < NSUB; ++bb) { lbv[bb] = lbase[bb]; asm volatile("" : "+v"(lbv[bb])); }
        const u32x4* wfv = wf;                     // (likewise the 64-bit addresses of the weight slices beyond the immediate-offset range)
        asm volatile("" : "+v"(wfv));
        // ---- K loop: KSTEPS MFMA steps x NSUB sub-tiles, weights from the rolling register stream -----------------------------------
        f32x16 acc[NSUB];
#pragma unroll
        for (int bb = 0; bb < NSUB; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bb][r] = 0.0f;
#pragma unroll
        for (int sl = 0; sl < NSLICE; ++sl) {
            const int u = sl % CP_DEPTH;
#pragma unroll
            for (int k = 0; k < CP_SL; ++k) {
                const int ks = sl * CP_SL + k, tap = ks / KPT, s = ks - tap * KPT, ky = tap / 3, kx = tap - 3 * ky;
                const int toff = ky * PITCH + (S == 2 ? (kx & 1) * G::EH + (kx >> 1) : kx);
                u32x4 fp[NSUB];
#pragma unroll
                for (int bb = 0; bb < NSUB; ++bb) {
                    const int idx = lbv[bb] + toff;
                    fp[bb] = *(const u32x4*)(patch + ((((idx << LSP) + (((2 * s + hi) ^ (idx >> GSH)) & (SP - 1)))) << 4));
                }
#pragma unroll
                for (int bb = 0; bb < NSUB; ++bb) mma_step<DT>(acc[bb], wq[u][k], fp[bb]);
            }
            {                                      // refill the slot three slices ahead — rolling over into the next tile's first slices
                const int nx = (sl + CP_DEPTH) % NSLICE;
#pragma unroll
                for (int k = 0; k < CP_SL; ++k) wq[u][k] = wfv[(nx * CP_SL + k) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);     // (left alone the scheduler hoists the fragment reads of several slices: registers)
        }
        u32x4 w2r[(CHAIN && !W2L) ? N / 16 : 1];
        if constexpr (CHAIN && !W2L) {
#pragma unroll
            for (int ks2 = 0; ks2 < N / 16; ++ks2) w2r[ks2] = *(const u32x4*)(w2lane + ks2 * 16);
        }
        lds_barrier();                             // every wave has left the K loop: the patch may be overwritten by the staged tile

        auto stage = [&](const f32x16 (&a)[NSUB], const f32x4* bv, const float* btab, float scale) {
#pragma unroll
            for (int bb = 0; bb < NSUB; ++bb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int nl = cg * 32 + 8 * qd + 4 * hi;
                    const f32x4 bb4 = btab ? *(const f32x4*)(btab + nl) : bv[qd];
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = a[bb][4 * qd + j] + bb4[j] + 0.0f;
                    silu4_f(v, v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= scale;
                    u32x2 pk;
                    if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                    else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                    *(u32x2*)(stg + ((ph * NSUB + bb) * 32 + l31) * SO + nl * E::BYTES) = pk;
                }
        };
        stage(acc, bq, nullptr, alpha_acc);
        lds_barrier();
        if constexpr (!CHAIN) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * CP_THREADS, row = idx / VPR, cv = idx - row * VPR;
                const int m = row_to_m(row, b, y0, x0), n = n0 + cv * VEC;
                if (m >= 0 && n < p.Cout) {
                    u32x4 sv = *(const u32x4*)(stg + row * SO + cv * 16);
                    if (rg) {                      // the shared epilogue's arithmetic: staged value + alpha_res * residual
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
        } else {
            // chained layer (one channel block: n0 = 0): y is completed now — staged vector + alpha_res * residual, written to y when the
            // chain keeps it and BACK into the staging tile, which the chained 1x1 consumes as stored (igemm's CHAIN + WB)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * CP_THREADS, row = idx / VPR, cv = idx - row * VPR;
                const int m = row_to_m(row, b, y0, x0), n = cv * VEC;
                if (m >= 0 && n < p.Cout) {
                    u32x4 sv = *(const u32x4*)(stg + row * SO + cv * 16);
                    if (rg) {
                        float v[VEC], r[VEC];
                        unpack16<DT>(sv, v);
                        unpack16<DT>(rres[it], r);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                        sv = pack16<DT>(v);
                        *(u32x4*)(stg + row * SO + cv * 16) = sv;
                    }
                    if (p.keep1) *(u32x4*)(yg + (long long)m * p.ldy + n) = sv;
                }
            }
            lds_barrier();                         // the completed tile is visible
            f32x16 acc2[NSUB];
#pragma unroll
            for (int bb = 0; bb < NSUB; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[bb][r] = 0.0f;
#pragma unroll
            for (int ks2 = 0; ks2 < N / 16; ++ks2)                   // K = the tile's channels
#pragma unroll
                for (int bb = 0; bb < NSUB; ++bb) {
                    const u32x4 fp2 = *(const u32x4*)(stg + ((ph * NSUB + bb) * 32 + l31) * SO + ((2 * ks2 + hi) << 4));
                    u32x4 fw2;
                    if constexpr (W2L) fw2 = *(const u32x4*)(w2l + ((ks2 * NCG + cg) * 64 + lane) * 16);
                    else fw2 = w2r[ks2];
                    mma_step<DT>(acc2[bb], fw2, fp2);
                }
            lds_barrier();                         // the tile has been consumed
            stage(acc2, nullptr, b2s, 1.0f);
            lds_barrier();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * CP_THREADS, row = idx / VPR, cv = idx - row * VPR;
                const int m = row_to_m(row, b, y0, x0), n = cv * VEC;
                if (m >= 0 && n < p.Cout2) *(u32x4*)(y2g + (long long)m * p.ldy2 + n) = *(const u32x4*)(stg + row * SO + cv * 16);
            }
        }
        if (tn >= t_hi) break;
        t = tn;
        cur ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct CpShape { int cin, s, tws, ncg; const char* tag; };
static const CpShape kCp[] = {{128, 1, 2, 4, "8x16n128"}, {64, 2, 2, 4, "8x16n128s2c64"}, {128, 2, 1, 8, "8x8n256s2"}, {128, 2, 1, 4, "8x8n128s2"},
                              {128, 1, 1, 4, "8x8n128w4"}, {64, 2, 1, 4, "8x8n128s2c64w4"}};
constexpr int CP_NSHAPES = 6;

const char* cwpers_tag(int shape) { return (shape >= 1 && shape <= CP_NSHAPES) ? kCp[shape - 1].tag : "?"; }

int cwpers_check(const icaf_conv_args* a, const ConvP& p, int shape) {
    if (shape < 1 || shape > CP_NSHAPES) return fail(ICAF_ERR_ARG, "cwpers: unknown shape %d", shape);
    const CpShape& sh = kCp[shape - 1];
    const int nwg = 32 * sh.ncg;
    if (a->dtype == ICAF_F32 || a->out_dtype != a->dtype) return fail(ICAF_ERR_UNSUPPORTED, "cwpers: 16-bit types, out dtype == dtype");
    if (a->kh != 3 || a->kw != 3 || a->sh != sh.s || a->sw != sh.s || a->ph != 1 || a->pw != 1) return fail(ICAF_ERR_UNSUPPORTED, "cwpers %s: 3x3 / stride %d / pad 1 layers", sh.tag, sh.s);
    if (a->Cin != sh.cin || a->Cout % nwg || a->Kp != 9 * sh.cin) return fail(ICAF_ERR_UNSUPPORTED, "cwpers %s: built for %d -> (multiples of %d) channels (Cin = %d, Cout = %d, Kp = %d)", sh.tag, sh.cin, nwg, a->Cin, a->Cout, a->Kp);
    if (a->act != ICAF_ACT_SILU || a->pre) return fail(ICAF_ERR_UNSUPPORTED, "cwpers: SiLU layers without a pre-activation term");
    if (!a->wf) return fail(ICAF_ERR_UNSUPPORTED, "cwpers: needs the fragment-major weight copy (icaf_conv_args.wf)");
    if (p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "cwpers: operand exceeds the 2 GiB buffer-descriptor range");
    if (!p.vec_y || (a->res && !p.vec_r)) return fail(ICAF_ERR_UNSUPPORTED, "cwpers: y / res must take 16-byte vectors");
    if (a->w2) {
        if (shape == 5) return fail(ICAF_ERR_UNSUPPORTED, "cwpers %s: no chained 1x1 (register budget)", sh.tag);
        if (sh.ncg != 4 || a->Cout != 128 || a->Cout2 > 128 || a->Cout2 % 32 || a->Kp2 != 128 || !p.vec_y2) return fail(ICAF_ERR_UNSUPPORTED, "cwpers: chained 1x1 of 128 -> (32, 64, 96 or 128) channels with Kp2 = 128");
        if (a->res && !a->chain_keep) return fail(ICAF_ERR_UNSUPPORTED, "cwpers: a residual needs chain_keep");
        if (a->chain_keep && (a->alpha_acc[0] != 1.0f || a->alpha_acc[1] != 1.0f)) return fail(ICAF_ERR_UNSUPPORTED, "cwpers: chain_keep with alpha_acc != 1");
    }
    return ICAF_OK;
}

template <int DT, int CIN, int STR, int TWS, int NCG, int NW, bool CHAIN>
static int launch_cwpers_cfg(const icaf_conv_args* a, const ConvP& p, hipStream_t s) {
    using G = CpTile<CIN, STR, TWS, NCG, NW>;
    CpGeom gm;
    gm.tiles_x = (p.Wo + G::TW - 1) / G::TW;
    gm.tiles_y = (p.Ho + CP_TH - 1) / CP_TH;
    gm.ntile = p.B * gm.tiles_x * gm.tiles_y;
    const int nblk = a->Cout / G::N;
    constexpr int lds_bytes = G::LDS + ((CHAIN && G::W2LDS) ? G::N * G::N * 2 : 0);
    static_assert(lds_bytes <= 160 * 1024, "LDS");
    int dev = 0, cus = 256;
    ICAF_HIP(hipGetDevice(&dev));
    ICAF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int grid = (cus * (NW == 8 ? 1 : G::WGPC) / (a->groups * nblk)) & ~7;    // the groups and channel blocks side by side; 8 XCDs
    if (grid < 8) grid = 8;
    ICAF_LDS_OPTIN((cwpers_kernel<DT, CIN, STR, TWS, NCG, NW, CHAIN>), lds_bytes);
    cwpers_kernel<DT, CIN, STR, TWS, NCG, NW, CHAIN><<<dim3((unsigned)grid, (unsigned)nblk, (unsigned)a->groups), dim3(G::THREADS), lds_bytes, s>>>(p, gm, a->wf, a->wf_gs);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <int DT, int CIN, int STR, int TWS, int NCG, int NW>
static int launch_cwpers_ch(const icaf_conv_args* a, const ConvP& p, hipStream_t s) {
    if constexpr (NCG == 4) { if (a->w2) return launch_cwpers_cfg<DT, CIN, STR, TWS, NCG, NW, true>(a, p, s); }
    return launch_cwpers_cfg<DT, CIN, STR, TWS, NCG, NW, false>(a, p, s);
}

template <int DT>
static int launch_cwpers_dt(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    switch (shape) {
        case 1: return launch_cwpers_ch<DT, 128, 1, 2, 4, 8>(a, p, s);
        case 2: return launch_cwpers_ch<DT, 64, 2, 2, 4, 8>(a, p, s);
        case 3: return launch_cwpers_ch<DT, 128, 2, 1, 8, 8>(a, p, s);
        case 4: return launch_cwpers_ch<DT, 128, 2, 1, 4, 8>(a, p, s);
        case 5: return launch_cwpers_cfg<DT, 128, 1, 1, 4, 4, false>(a, p, s);
        default: return launch_cwpers_ch<DT, 64, 2, 1, 4, 4>(a, p, s);
    }
}

// shapes: see kCp (tile id 90 + shape)
int launch_cwpers(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    int st = cwpers_check(a, p, shape);
    if (st) return st;
    return a->dtype == ICAF_BF16 ? launch_cwpers_dt<ICAF_BF16>(a, p, shape, s) : launch_cwpers_dt<ICAF_F16>(a, p, shape, s);
}

}  // namespace icaf
