// Persistent streaming GEMM for the 1x1 / stride-1 layers on gfx950 (MI355X) — reference models/common.py:48-60 with k = 1
// (C3's cv1 | cv2 / cv3, the Bottleneck's cv1, SPPF's cv1 / cv2, the PANet 1x1s), 16-bit types.
//
//   y[m][n] = alpha_acc * act( sum_k x[m][k] * Wp[n][k] + bias[n] )          m = pixel (NHWC row), k = input channel
//
// A 1x1 layer is a plain [M x K] . [K x N] GEMM with K = 64 .. 1024: its K loop is 1-8 slices of 128 bytes, so an ordinary
// one-tile-per-workgroup launch (igemm.hip) spends most of a tile's life in its prologue (first slice: a full HBM round trip with
// nothing else in flight for this workgroup) and its epilogue (no loads in flight at all).  These layers are HBM-bound streaming
// jobs — 30 % of the forward's kernel time — and ran at 1.1-3.5 TB/s.  Here
//   * a workgroup is PERSISTENT: it keeps one channel tile (BN = 128 / 64 channels) and walks pixel tiles of 128 rows;
//   * the K slices of ALL its tiles form ONE continuous stream through an NS-stage LDS ring fed by `buffer_load ... lds`:
//     the slices of tile t + 1 are already in flight while tile t finishes its MFMAs, runs its epilogue and stores — the
//     bytes in flight per CU never drop to zero at a tile boundary;
//   * the epilogue has its OWN staging buffer (the ring is busy) and uses LDS-only barriers: `__syncthreads()` would drain
//     vmcnt, i.e. wait for the prefetched slices; the bias vector lives in registers for the workgroup's whole life;
//   * counted `s_waitcnt vmcnt`: every slice is exactly NA + NB DMA instructions per wave (zero-fill slices are issued past
//     the end of the stream to keep that count), and the epilogue's global stores — younger than the next tile's first slices,
//     older than the ones issued after them — never invalidate the count: "at most one slice's worth of operations outstanding"
//     always implies that the slice about to be consumed has landed (its younger sibling slice alone accounts for that many);
//   * arithmetic = igemm's: same K order, same MFMA step, same bias / activation / rounding expressions => bit-identical
//     results (tested: every launch configuration of a layer must give the same bits).
// XCD-aware walk: XCD x owns the x-th contiguous eighth of the pixel tiles; the workgroups that compute the different channel
// tiles of one pixel tile run side by side on that XCD, so the pixels are fetched from HBM once and re-read out of ONE L2.
#pragma once
#include "conv_common.h"

namespace icaf {

// EPI = epilogue policy: what happens to a finished [128 x BN] accumulator tile.
//   EPI::SO                        staging row stride in bytes (the workgroup's staging buffer is 128 rows)
//   epi.stage(acc, bq, stg, row0, col0, l31, hi, m0, n0)   this lane's accumulator quads (+ bias bq) -> staging rows row0 + b * 32 + l31
//                                  (m0 / n0: first pixel / channel of the tile, for policies whose arithmetic depends on the position)
//   epi.flush(stg, m0, n0, tid)                    staging -> global memory, after an LDS-only barrier
// MODE 1: 1x1 / stride 1 / pad 0 — the pixel operand is a plain row-major matrix.  MODE 2: any filter with Cin * bytes a multiple of
// 128 (a K slice lies inside ONE tap): igemm.hip's implicit-GEMM gather with a wave-uniform tap walk, restarted at every tile.
template <int DT, int BN, int MODE, class EPI>
__device__ __forceinline__ void stream_gemm(const ConvP& p, const EPI& epi) {
    using E = Elem<DT>;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int BM = 128, RB = 128, NS = 3, NW = 8;
    constexpr int WN = 32, WM = BN == 128 ? 64 : 32;                 // 8 waves: 2 x 4 (BN = 128) or 4 x 2 (BN = 64)
    constexpr int TM = WM / 32, TN = 1;
    constexpr int WAVES_M = BM / WM;
    constexpr int VEC = E::VEC;                                      // 8
    constexpr int RPI = 8;                                           // LDS rows per wave-wide DMA instruction (1024 / RB)
    constexpr int NA = BM / RPI / NW, NB = BN / RPI / NW;            // DMA instructions per slice and wave: 2 + 2 (BN = 128), 2 + 1
    constexpr int PER = NA + NB;
    constexpr int STAGE = (BM + BN) * RB;
    constexpr int NSTEP = RB / 32;                                   // 4 MFMA steps per slice
    constexpr int RING = NS * STAGE;
    static_assert(NA >= 1 && NB >= 1 && (BN == 128 || BN == 64), "tile shape");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* stg = lds + RING;                                 // epilogue staging: BM rows x EPI::SO bytes

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.w + g * p.w_gs), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // lane -> (row, 16-byte slot) of a DMA instruction, igemm's RB = 128 swizzle: key(row) = (row >> 1) & 7; a wave's
    // instructions are j = wave + NW * i, so row = j * 8 + rsub has key ((wave & 1) << 2) | (rsub >> 1)
    const int rsub = lane >> 3;
    const int dkey = ((wave & 1) << 2) | (rsub >> 1);
    const int lslot = (lane & 7) ^ dkey;

    // ---- tile walk.  A workgroup keeps ONE channel tile nt for its whole life (bias in registers, no per-tile decode) and walks pixel
    //      tiles: XCD x owns the x-th contiguous eighth of the pixel tiles; its wgx workgroups are nt = lb % ntiles, pixel slot
    //      lb / ntiles — the ntiles workgroups of one pixel slot read the same pixels at the same time, out of the same L2.
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, wgx = gridDim.x >> 3;
    const int per_xcd = (p.mtiles + 7) >> 3, m_lo = xcd * per_xcd, m_hi = min(m_lo + per_xcd, p.mtiles);
    const int nt = lb % p.ntiles, mstride = wgx / p.ntiles, m_first = m_lo + lb / p.ntiles;
    const int n0 = nt * BN;
    const int nch = p.nchunks;

    // issue cursor (runs NS - 1 .. NS slices ahead of the consume cursor)
    int it = m_first, ic = 0;
    int kc = 0, ky = 0, kx = 0;                   // MODE 2: K position of the cursor's slice (wave-uniform): tap (ky, kx), channel kc
    unsigned ia_off[NA], iw_off[NB];
    int ia_h0[NA], ia_w0[NA];
#pragma unroll
    for (int i = 0; i < NB; ++i)
        iw_off[i] = ((unsigned)(n0 + (wave + NW * i) * RPI + rsub) * (unsigned)p.Kp + (unsigned)(lslot * VEC)) * E::BYTES;
    auto issue_setup = [&]() {                    // per-tile lane offsets of the issue cursor's pixel tile (OOB beyond the tensor / the stream)
        const bool live = it < m_hi;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = it * BM + (wave + NW * i) * RPI + rsub;
            const bool ok = live && m < p.M;
            if constexpr (MODE == 1) {
                ia_off[i] = ok ? ((unsigned)m * (unsigned)p.ldx + (unsigned)(lslot * VEC)) * E::BYTES : OOB;
            } else {
                const int mm = ok ? m : 0;
                const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, b = t / p.Ho;
                ia_h0[i] = ok ? ho * p.sh - p.ph : -0x10000;                          // (dead rows: every tap fails the bounds test)
                ia_w0[i] = wo * p.sw - p.pw;
                ia_off[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES
                          + (unsigned)((ia_h0[i] * p.W + ia_w0[i]) * p.ldx + lslot * VEC) * E::BYTES;   // tap (0, 0) (may wrap below 0)
            }
        }
    };
    // DMA instructions of the cursor's slice, portion `part` of NSTEP, into ring stage `stage`
    auto issue_part = [&](int stage, int part) {
        unsigned char* st = lds + stage * STAGE;
        const unsigned koff = (unsigned)ic * RB;
        unsigned tap_delta = 0;
        if constexpr (MODE == 2) tap_delta = (unsigned)((ky * p.W + kx) * p.ldx + kc) * E::BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i % NSTEP != part) continue;
            unsigned voff;
            if constexpr (MODE == 1) {
                voff = ia_off[i] == OOB ? OOB : ia_off[i] + koff;
            } else {
                const int h = ia_h0[i] + ky, w = ia_w0[i] + kx;
                const bool ok = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                voff = ok ? ia_off[i] + tap_delta : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(st + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if ((NA + i) % NSTEP != part) continue;
            const unsigned voff = it < m_hi ? iw_off[i] + koff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(st + BM * RB + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto issue_advance = [&]() {                  // (wave-uniform)
        if constexpr (MODE == 2) {
            kc += RB / E::BYTES;
            if (kc >= p.Cin) { kc = 0; if (++kx == p.kw) { kx = 0; ++ky; } }
        }
        if (++ic == nch) {
            ic = 0; kc = 0; ky = 0; kx = 0;
            it += mstride;
            issue_setup();
        }
    };

    // bias of this lane's channels, in registers for the workgroup's whole life; consumed right here: otherwise the compiler's
    // wait-count bookkeeping treats it as possibly pending at every use and drains the ring with it
    const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
    f32x4 bq[4];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const int n = n0 + wn * WN + 8 * qd + 4 * hi;
        const bool okn = bias && n < p.Cout;                       // (the packed bias is padded to a multiple of 128 >= Cout only)
        const float* bp = okn ? bias + n : (const float*)p.w;      // (any mapped address: the value is discarded)
        const f32x4 t = *(const f32x4*)bp;
        bq[qd] = okn ? t : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) asm volatile("" : "+v"(bq[qd]));

    if (m_first >= m_hi) return;                                     // (workgroup-uniform; before any barrier)

    // fragment reads: row * RB + ((2 * step + hi) ^ key(row)) * 16 with key(row) = key(l31) (fragment base rows are multiples of 32)
    const int fkey = (l31 >> 1) & 7;
    int foff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);

    // ---- prologue: every stage is empty, NS slices go out at once ----------------------------------------------------------
    issue_setup();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int part = 0; part < NSTEP; ++part) issue_part(s, part);
        issue_advance();
    }

    int q = 0;                                                       // slices consumed so far: slice q lives in stage q % NS
    for (int mt = m_first; mt < m_hi; mt += mstride) {
        const int m0 = mt * BM;
        f32x16 acc[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
        for (int c = 0; c < nch; ++c, ++q) {
            // slice q has landed once at most ONE younger slice's worth of this wave's operations is outstanding: its younger sibling
            // q + 1 alone is PER operations, and whatever else is outstanding (slice q + 2 right after the prologue, a previous tile's
            // stores) is younger still (header comment)
            wait_vmcnt<PER>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // (a) slice q visible to every wave, (b) stage (q - 1) % NS is free
            const int sq = q % NS;
            const unsigned char* a_s = lds + sq * STAGE;
            const unsigned char* b_s = a_s + BM * RB;
            u32x4 fp[NSTEP][TM], fw[NSTEP];
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
                for (int b = 0; b < TM; ++b) fp[s][b] = *(const u32x4*)(a_s + (wm * WM + b * 32) * RB + foff[s]);
                fw[s] = *(const u32x4*)(b_s + (wn * WN) * RB + foff[s]);
            }
            const int sfree = (q + NS - 1) % NS;   // = (q - 1) % NS
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
                for (int b = 0; b < TM; ++b) mma_step<DT>(acc[b], fw[s], fp[s][b]);
                if (q > 0) issue_part(sfree, s);   // (slice NS - 1 left with the prologue)
            }
            if (q > 0) issue_advance();
        }

        // ---- epilogue: the policy stages the tile through the workgroup's own LDS buffer and writes it out ------------------------
        epi.stage(acc, bq, stg, wm * WM, wn * WN, l31, hi, m0, n0);
        lds_barrier();                             // staged tile visible (LDS-only: the prefetched slices stay in flight)
        epi.flush(stg, m0, n0, tid);
        // (the ring barrier of the next slice separates these staging reads from the next tile's staging writes)
    }
    wait_vmcnt<0>();                               // zero-fill slices issued past the end of the stream
}


}  // namespace icaf
