// Implicit-GEMM convolution / linear layer for gfx950 (MI355X).
//
//   y[m][n] = alpha_res * res[m][n] + alpha_acc * act( sum_k A[m][k] * Wp[n][k] + bias[n] )
//
// m = output pixel (b, ho, wo) of an NHWC tensor, k = (kh, kw, cin) gathered on the fly from the NHWC input (no
// im2col buffer), Wp = packed K-major weights.  Replaces reference models/common.py:48-60 (Conv), :184-194
// (Bottleneck shortcut), nn.Linear layers of :607-618 / :704-709 and models/yolo_test.py:50 (Detect convs).
//
// Common structure (256-thread workgroup = 4 wavefronts of 64, BM x BN output tile, K walked in 64-byte slices):
//   * v_mfma_f32_32x32x16_{bf16,f16} (or v_mfma_f32_32x32x2_f32 x4 in the fp32 parity build): the weight rows are
//     the MFMA A operand and the pixel rows the B operand, so each lane's accumulator registers hold 4 consecutive
//     output channels of ONE pixel — bias/activation are applied in registers, the tile is staged through LDS
//     (reusing the operand buffers) and written back as whole 16-byte channel vectors, fully coalesced, with the
//     residual read the same way;
//   * blockIdx -> tile mapping is XCD-aware: tiles that share an input tile (same pixels, different channel
//     block) and neighbouring pixel tiles (3x3 halos) are placed on the same XCD so re-reads hit its private L2.
//
// Two operand-staging pipelines share that structure:
//   igemm_dma_kernel  (default) LDS-DMA: every K slice (64 or 128 bytes per row) is fetched with
//     `buffer_load_dwordx4 ... lds` straight into a 2/3-stage LDS ring (no VGPR round trip, no ds_write), slices stay
//     in flight across the single s_barrier per slice (counted s_waitcnt vmcnt), halo / tail / out-of-range lanes
//     are zero-filled by the buffer descriptor's range check, and the LDS image is XOR-swizzled through the per-lane
//     SOURCE address (the DMA destination is lane-linear) so that the ds_read_b128 fragment reads are bank-conflict
//     free without padding.
//   igemm_kernel      register-staged double buffer with padded LDS rows; used when an operand exceeds the 2 GiB
//     buffer-descriptor range and kept selectable for A/B measurements (configuration ids 11..14).
#include "conv_common.h"

static_assert(sizeof(icaf_conv_args) == 312, "icaf_conv_args layout is mirrored by ctypes in icafusion_amd/_lib.py");
static_assert(sizeof(icaf_bneck_args) == 376, "icaf_bneck_args layout is mirrored by ctypes in icafusion_amd/_lib.py");

namespace icaf {

// RB = bytes of K per LDS row per slice (64 or 128), NS = ring depth.  With RB = 128 every DMA lane group fetches a
// whole 128-byte cache line of one pixel / weight row: the LDS-DMA feed rate from L2 measured on MI355X is 14-21 TB/s
// for full lines against 8 TB/s for 64-byte half lines (lab/probes/dma_bw_probe.hip), and that feed rate — not the
// MFMA pipe — is what bounds these small-tile GEMMs.
// MODE selects the pixel-operand address generator (the K order, hence the result, is the same for all three):
//   0  generic: every lane walks its own (ky, kx, cin) position — needed when a K slice straddles filter taps
//      (Cin * bytes not a multiple of RB);
//   1  1x1 / stride 1 / pad 0 with Cin * bytes a multiple of RB: the pixel operand is a plain row-major matrix, one
//      VALU add per DMA instruction and slice;
//   2  any filter with Cin * bytes a multiple of RB: a slice lies inside ONE tap, so the tap walk is wave-uniform
//      (scalar unit) and a row costs two bounds checks and one add.
// Loop shape: after the single barrier of a slice, ALL its fragments are read into registers (LDS latency overlaps the
// address arithmetic), then the next slice's DMA instructions are issued in NSTEP portions between the MFMA steps, so
// their VALU work hides under the matrix pipe instead of preceding it.
// CHAIN = true appends a second, 1x1 GEMM to the tile before anything is stored (icaf.h: w2 / bias2 / y2): the SiLU'd
// output tile is staged in LDS as bf16 / f16 exactly as it would be written to HBM, re-read as the pixel operand of
//   y2 = SiLU(W2 . tile + bias2)     (K = this layer's channels, all inside the one N tile; W2 resident in LDS, fetched by
// LDS-DMA while the main loop runs), and only y2 is written.  Same rounding points and K order as two launches.
template <int DT, int ODT, int BM, int BN, int WM, int WN, int ACT, int RB, int NS, int MODE, bool PRE, bool CHAIN = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void igemm_dma_kernel(const ConvP p) {
    using E = Elem<DT>;
    using L = TileLds<DT, ODT, BM, BN>;
    constexpr int VEC = E::VEC;
    constexpr int BK = RB / E::BYTES;              // K elements per slice
    constexpr int SPR = RB / 16;                   // 16-byte slots per row (4 / 8) = lanes per row of a DMA instruction
    constexpr int RPI = 1024 / RB;                 // LDS rows written by one wave-wide DMA instruction (16 / 8)
    constexpr int AI = BM / RPI, BI = BN / RPI;    // DMA instructions per slice for the pixel / weight tile
    constexpr int NW = (BM / WM) * (BN / WN);      // wavefronts per workgroup (4, or 8 for the 256-row tiles)
    constexpr int NA = AI / NW;                    // ... per wave (pixel tile)
    constexpr int NBF = BI / NW, NBR = BI % NW;    // weight tile: NBF per wave, waves < NBR one more
    constexpr int NBMAX = NBF + (NBR ? 1 : 0);
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_M = BM / WM;
    constexpr int STAGE = (BM + BN) * RB;
    constexpr int NSTEP = RB / 32;                 // MFMA steps per slice
    static_assert(AI % NW == 0 && (NW == 4 || NW == 8), "tile shape");
    // (the launch allocates max(ring, epilogue staging) bytes of LDS)
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const int tile = xcd_tile(p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.w + g * p.w_gs), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;          // beyond any legal range (operands are < 2 GiB): reads as zero

    // Lane -> (row, slot) of one DMA instruction: SPR lanes per row; the LDS slot is lane-linear, the SOURCE slot is
    // XOR-swizzled with key(row).  RB = 64: key = (row >> 2) & 3;  RB = 128: key = (row >> 1) & 7.  A wave's
    // instructions are j = wave + NW*i (NW even), so row = j*RPI + rsub has a key that depends on (wave & 1, rsub) only.
    const int rsub = lane / SPR;
    const int dkey = RB == 64 ? ((rsub >> 2) & 3) : (((wave & 1) << 2) | (rsub >> 1));
    const int lslot = (lane % SPR) ^ dkey;                         // logical 16-byte slot fetched by this lane
    unsigned a_off[NA];                            // MODE 0: image base; MODE 1: running row offset; MODE 2: tap-0 offset
    int a_h0[NA], a_w0[NA];
    bool a_ok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave + NW * i) * RPI + rsub;
        const int m = m0 + row;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        if constexpr (MODE == 1) {
            a_off[i] = a_ok[i] ? ((unsigned)mm * (unsigned)p.ldx + (unsigned)(lslot * VEC)) * E::BYTES : OOB;
        } else {
            const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, b = t / p.Ho;
            a_h0[i] = ho * p.sh - p.ph;
            a_w0[i] = wo * p.sw - p.pw;
            a_off[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES;
            if constexpr (MODE == 2)               // offset of tap (0, 0), channel slot of this lane (may wrap below 0)
                a_off[i] += (unsigned)((a_h0[i] * p.W + a_w0[i]) * p.ldx + lslot * VEC) * E::BYTES;
        }
    }
    // MODE 0: per-lane position; MODE 2: wave-uniform position of the slice (kc = channel offset inside the tap)
    int kc = MODE == 0 ? lslot * VEC : 0, ky = 0, kx = 0;
    if constexpr (MODE == 0) { while (kc >= p.Cin) { kc -= p.Cin; if (++kx == p.kw) { kx = 0; ++ky; } } }
    unsigned w_row[NBMAX];
#pragma unroll
    for (int i = 0; i < NBMAX; ++i)
        w_row[i] = ((unsigned)(n0 + (wave + NW * i) * RPI + rsub) * (unsigned)p.Kp + (unsigned)(lslot * VEC)) * E::BYTES;
    const int nb_mine = NBF + (wave < NBR ? 1 : 0);

    // DMA instructions of one slice, portion `part` of NSTEP (instruction j of the wave belongs to portion j % NSTEP)
    auto issue_part = [&](int chunk, int stage, int part) {
        unsigned char* st = lds + stage * STAGE;
        unsigned tap_delta = 0;
        bool kvalid = true;
        if constexpr (MODE == 0) kvalid = ky < p.kh;
        if constexpr (MODE == 2) { kvalid = ky < p.kh; tap_delta = (unsigned)((ky * p.W + kx) * p.ldx + kc) * E::BYTES; }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (i % NSTEP != part) continue;
            unsigned voff;
            if constexpr (MODE == 1) {
                voff = a_off[i] + (unsigned)chunk * RB;
            } else {
                const int h = a_h0[i] + ky, w = a_w0[i] + kx;
                const bool ok = a_ok[i] && kvalid && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
                if constexpr (MODE == 2) voff = ok ? a_off[i] + tap_delta : OOB;
                else voff = ok ? a_off[i] + (unsigned)((h * p.W + w) * p.ldx + kc) * E::BYTES : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(st + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
        }
        const bool cvalid = chunk < p.nchunks;
#pragma unroll
        for (int i = 0; i < NBMAX; ++i) {
            if ((NA + i) % NSTEP != part) continue;
            if (i < nb_mine) {
                const unsigned voff = cvalid ? w_row[i] + (unsigned)chunk * RB : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(st + BM * RB + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
            }
        }
    };
    auto advance = [&]() {                         // move the K position by one slice
        if constexpr (MODE == 0) {
            kc += BK;
            while (kc >= p.Cin) { kc -= p.Cin; if (++kx == p.kw) { kx = 0; ++ky; } }
        } else if constexpr (MODE == 2) {
            kc += BK;
            if (kc >= p.Cin) { kc = 0; if (++kx == p.kw) { kx = 0; ++ky; } }
        }
    };

    // chained 1x1: its weights (BN rows x BN channels, 128-byte slices, igemm's swizzle) go to LDS behind the ring / staging
    constexpr int RING_BYTES = NS * STAGE > L::OUT_BYTES ? NS * STAGE : L::OUT_BYTES;
    constexpr int W2_OFF = (RING_BYTES + 1023) / 1024 * 1024;
    constexpr int W2_SLICES = BN / 64;             // K2 = BN channels of a 16-bit type = BN/64 slices of 128 bytes
    if constexpr (CHAIN) {
        const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const typename E::type*)p.w2 + g * p.w2_gs), 0, p.w2_bytes, 0x00020000);
        const int rs8 = lane >> 3;
        auto issue_w2 = [&]() {
#pragma unroll
            for (int c2 = 0; c2 < W2_SLICES; ++c2)
#pragma unroll
                for (int i = 0; i < BN / 8 / NW; ++i) {
                    const int j = wave + NW * i;                       // 8 weight rows per instruction
                    const int sl = (lane & 7) ^ (((j & 1) << 2) | (rs8 >> 1));
                    const unsigned voff = ((unsigned)(j * 8 + rs8) * (unsigned)p.Kp2 + (unsigned)(c2 * 64 + sl * 8)) * 2u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w2r, (lds_ptr_t)(lds + W2_OFF + c2 * BN * 128 + j * 1024), 16, voff, 0, 0, 0);
                }
        };
        issue_w2();
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    // fragment reads: row*RB + ((2*step + hi) ^ key(row)) * 16; every fragment base row is a multiple of 32, so
    // key(row) = key(l31)
    const int fkey = RB == 64 ? ((l31 >> 2) & 3) : ((l31 >> 1) & 7);
    int foff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);

    // Prologue: ALL NS stages are empty, so the first NS slices go out at once (the steady-state loop below keeps NS - 1 in flight:
    // it can only refill the stage consumed one barrier earlier).  A 1x1 layer's K loop is 2-4 slices long; with the first two
    // round trips to HBM overlapped instead of chained, a K = 128 tile waits for memory once, not twice.
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int part = 0; part < NSTEP; ++part) issue_part(s, s, part);
        advance();
    }

    for (int c = 0; c < p.nchunks; ++c) {
        // slice c has landed once at most the (NS-2) younger slices of this wave are still outstanding (c = 0: NS - 1 younger slices
        // are outstanding; the same count then also waits for slice 1, which costs nothing extra: both were issued together)
        if (nb_mine == NBF) wait_vmcnt<(NS - 2) * (NA + NBF)>();
        else wait_vmcnt<(NS - 2) * (NA + NBF + 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // (a) slice c visible to every wave, (b) stage (c-1)%NS is free
        const unsigned char* a_s = lds + (c % NS) * STAGE;
        const unsigned char* b_s = a_s + BM * RB;
        u32x4 fp[NSTEP][TM], fw[NSTEP][TN];
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
            for (int b = 0; b < TM; ++b) fp[s][b] = *(const u32x4*)(a_s + (wm * WM + b * 32) * RB + foff[s]);
#pragma unroll
            for (int a = 0; a < TN; ++a) fw[s][a] = *(const u32x4*)(b_s + (wn * WN + a * 32) * RB + foff[s]);
        }
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) mma_step<DT>(acc[a][b], fw[s][a], fp[s][b]);
            if (c > 0) issue_part(c + NS - 1, (c + NS - 1) % NS, s);          // (slice NS - 1 left with the prologue)
        }
        if (c > 0) advance();
    }
    wait_vmcnt<0>();                               // drain the zero-fill slices issued past the end
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (CHAIN) {
        static_assert(DT != ICAF_F32 && ODT == DT && ACT == ICAF_ACT_SILU && !PRE, "chained 1x1: 16-bit SiLU layers");
        constexpr int SO = BN * E::BYTES + 16;     // staging row stride (as the epilogue's)
        const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
        // (a) this layer's output tile -> LDS, rounded to the storage type (and, with keep1, written to y by the ordinary
        //     epilogue, whose staged tile is the same thing)
        if (p.keep1) {            // (a residual — the Bottleneck shortcut — is added by the epilogue and written back to the tile)
            epilogue<DT, ODT, BM, BN, WM, WN, ICAF_ACT_SILU, false, false, true>(acc, lds, p, g, [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; }, 0);
        } else {
            f32x4 bvr[TN][4];                      // all bias quads in one batch of loads (see conv_common.h: epilogue)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = wn * WN + a * 32 + 8 * q + 4 * hi;
                    const bool okn = nl < p.Cout;
                    bvr[a][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (bias) {
                        const f32x4 t = *(const f32x4*)(bias + (okn ? nl : 0));
                        bvr[a][q] = okn ? t : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = wn * WN + a * 32 + 8 * q + 4 * hi;
                    const f32x4 bv = bvr[a][q];
#pragma unroll
                    for (int b = 0; b < TM; ++b) {
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = acc[a][b][4 * q + j] + bv[j];
                        silu4_f(v, v);
                        u32x2 pk;
                        if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                        else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                        *(u32x2*)(lds + (wm * WM + b * 32 + l31) * SO + nl * E::BYTES) = pk;
                    }
                }
        }
        __syncthreads();
        // (b) y2 tile = W2 . tile: pixels are the rows of the staged tile, K = its BN channels
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
        const int fk2 = (l31 >> 1) & 7;
#pragma unroll
        for (int s2 = 0; s2 < BN / 16; ++s2) {
            u32x4 fp2[TM], fw2[TN];
#pragma unroll
            for (int b = 0; b < TM; ++b) fp2[b] = *(const u32x4*)(lds + (wm * WM + b * 32 + l31) * SO + ((2 * s2 + hi) << 4));
#pragma unroll
            for (int a = 0; a < TN; ++a)
                fw2[a] = *(const u32x4*)(lds + W2_OFF + (s2 >> 2) * BN * 128 + (wn * WN + a * 32 + l31) * 128 + (((2 * (s2 & 3) + hi) ^ fk2) << 4));
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) mma_step<DT>(acc[a][b], fw2[a], fp2[b]);
        }
        __syncthreads();                           // the staged tile has been consumed: the epilogue may overwrite it
        // (c) the ordinary epilogue, writing the second layer's fields of p (SECOND)
        epilogue<DT, ODT, BM, BN, WM, WN, ICAF_ACT_SILU, false, true>(acc, lds, p, g, [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; }, 0);
    } else {
        epilogue<DT, ODT, BM, BN, WM, WN, ACT, PRE>(acc, lds, p, g, [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; }, n0);
    }
}

// ===============================================================================================================
// register-staged pipeline (fallback / A-B reference)
// ===============================================================================================================
template <int DT, int ODT, int BM, int BN, int WM, int WN, int ACT>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(const ConvP p) {
    using E = Elem<DT>;
    using L = TileLds<DT, ODT, BM, BN>;
    constexpr int VEC = E::VEC;              // elements per 16-byte vector
    constexpr int BK = ROWB / E::BYTES;      // K elements per chunk
    constexpr int NA = BM * 4 / NTHREADS;    // 16-byte vectors of the pixel tile per thread
    constexpr int NBv = (BN * 4 + NTHREADS - 1) / NTHREADS;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_M = BM / WM;
    constexpr int LDS_BYTES = L::REG_BYTES > L::OUT_BYTES ? L::REG_BYTES : L::OUT_BYTES;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    static_assert(LDS_BYTES <= 65536, "static LDS limit");

    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int g = blockIdx.z;
    const int tile = xcd_tile(p.mtiles * p.ntiles);
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const typename E::type* __restrict__ xg = (const typename E::type*)p.x + g * p.x_gs;
    const typename E::type* __restrict__ wg = (const typename E::type*)p.w + g * p.w_gs;

    long long a_base[NA];
    int a_h0[NA], a_w0[NA];
    bool a_ok[NA];
    const int kv = tid & 3;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (tid >> 2) + i * (NTHREADS / 4);
        const int m = m0 + row;
        a_ok[i] = m < p.M;
        const int mm = a_ok[i] ? m : 0;
        const int wo = mm % p.Wo, t = mm / p.Wo, ho = t % p.Ho, b = t / p.Ho;
        a_h0[i] = ho * p.sh - p.ph;
        a_w0[i] = wo * p.sw - p.pw;
        a_base[i] = (long long)b * p.H * p.W * p.ldx;
    }
    int kc = kv * VEC, ky = 0, kx = 0;
    while (kc >= p.Cin) { kc -= p.Cin; if (++kx == p.kw) { kx = 0; ++ky; } }

    const long long w_row = (long long)(n0 + (tid >> 2)) * p.Kp + kv * VEC;

    u32x4 ra[NA], rb[NBv];
    auto load_tiles = [&](int chunk) {
        const bool kvalid = ky < p.kh;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int h = a_h0[i] + ky, w = a_w0[i] + kx;
            const bool ok = a_ok[i] && kvalid && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *(const u32x4*)(xg + a_base[i] + ((long long)h * p.W + w) * p.ldx + kc);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NBv; ++i) {
            const int row = (tid >> 2) + i * (NTHREADS / 4);
            if (row < BN) rb[i] = *(const u32x4*)(wg + w_row + (long long)i * (NTHREADS / 4) * p.Kp + (long long)chunk * BK);
        }
        kc += BK;
        while (kc >= p.Cin) { kc -= p.Cin; if (++kx == p.kw) { kx = 0; ++ky; } }
    };
    auto store_tiles = [&](int buf) {
        unsigned char* a_s = lds + buf * (BM + BN) * ROWS;
        unsigned char* b_s = a_s + BM * ROWS;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = (tid >> 2) + i * (NTHREADS / 4);
            *(u32x4*)(a_s + row * ROWS + kv * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NBv; ++i) {
            const int row = (tid >> 2) + i * (NTHREADS / 4);
            if (row < BN) *(u32x4*)(b_s + row * ROWS + kv * 16) = rb[i];
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int c = 0; c < p.nchunks; ++c) {
        const bool more = c + 1 < p.nchunks;
        if (more) load_tiles(c + 1);
        const unsigned char* a_s = lds + (c & 1) * (BM + BN) * ROWS;
        const unsigned char* b_s = a_s + BM * ROWS;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 fp[TM], fw[TN];
#pragma unroll
            for (int b = 0; b < TM; ++b)
                fp[b] = *(const u32x4*)(a_s + (wm * WM + b * 32 + l31) * ROWS + s * 32 + hi * 16);
#pragma unroll
            for (int a = 0; a < TN; ++a)
                fw[a] = *(const u32x4*)(b_s + (wn * WN + a * 32 + l31) * ROWS + s * 32 + hi * 16);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) mma_step<DT>(acc[a][b], fw[a], fp[b]);
        }
        if (more) store_tiles((c + 1) & 1);
        __syncthreads();
    }
    epilogue<DT, ODT, BM, BN, WM, WN, ACT, false>(acc, lds, p, g, [&](int row) { const int m = m0 + row; return m < p.M ? m : -1; }, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct TileCfg { int id, bm, bn; const char* tag; };
static const TileCfg kTiles[] = {{1, 128, 128, "128x128"}, {2, 128, 64, "128x64"}, {3, 256, 32, "256x32"}, {4, 64, 64, "64x64"},
                                 {5, 256, 128, "256x128"}, {6, 256, 256, "256x256"}, {7, 0, 0, "-"}, {8, 128, 128, "128x128w8"},
                                 {9, 128, 64, "128x64w8"}};

// Launch configuration id = tile (1..4) + 10 * pipeline:
//   pipeline 0: LDS-DMA, 64-byte slices, 3-stage ring      pipeline 1: register-staged (fallback)
//   pipeline 2: LDS-DMA, 128-byte slices, 2-stage ring     pipeline 3: LDS-DMA, 128-byte slices, 3-stage ring
// Tiles 8 (128x128) and 9 (128x64) are the 128-row tiles with 8 wavefronts (the tuner uses 8: +1.3 % on the forward; 9 wins
// isolated timings but loses in the graph, where it competes with the DMFF branches for wave slots, so it is not a candidate).
// Tiles 5 (256x128) and 6 (256x256) are 8-wavefront workgroups (one per CU, 96 / 128 KB ring) that exist only on
// pipeline 2 for the 16-bit types: they halve the L2 -> LDS bytes per FLOP of the 128x128 tile, which is what bounds
// the deep layers (the LDS-DMA feed tops out near 20 bytes / clock / CU).
// ctile.hip
int ctile_check(const icaf_conv_args* a, const ConvP& p, int shape);
int launch_ctile(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s);
int launch_bneck(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s);
const char* ctile_tag(int shape);
// igemm_stream.hip
int stream_check(const icaf_conv_args* a, const ConvP& p, int shape);
int launch_stream(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s);
const char* stream_tag(int shape);
// cstream.hip
int cstream_check(const icaf_conv_args* a, const ConvP& p);
int launch_cstream(const icaf_conv_args* a, const ConvP& p, hipStream_t s);
// cwide.hip
int cwide_check(const icaf_conv_args* a, const ConvP& p, int shape);
int launch_cwide(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s);
const char* cwide_tag(int shape);
// igemm_wreg.hip
int wreg_check(const icaf_conv_args* a, const ConvP& p, int shape);
int launch_wreg(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s);
const char* wreg_tag(int shape);

//   40 + shape: 3x3 direct convolution from an LDS halo tile (ctile.hip); 50 + shape: persistent streaming GEMM for 1x1 layers
//   (igemm_stream.hip); 60 + shape: weight operand fed from registers (igemm_wreg.hip); an explicit request that the layer cannot
//   satisfy is an error (the autotuner skips it), it is never chosen silently.  71: persistent 3x3 with a resident filter (cstream.hip);
//   80 + shape: 3x3 (stride 1 / 2) from a resident halo patch with the weights streamed per wave into registers (cwide.hip).
static int pick_tile(const icaf_conv_args* a, const ConvP& p) {
    const bool dma_ok = p.x_bytes != 0;
    if (a->tile > 40 && a->tile < 90) return a->tile;
    if (a->tile >= 1 && a->tile <= 34 && a->tile % 10 >= 1 && a->tile % 10 <= 4) {
        const int pipe = a->tile / 10;
        return (pipe != 1 && !dma_ok) ? a->tile % 10 + 10 : a->tile;
    }
    if (a->tile == 25 || a->tile == 26 || a->tile == 28 || a->tile == 29) return a->tile;          // validated in launch_tile
    if (a->x2) return 81;                           // C3 tail: cwide.hip only (its check names what the layer must look like)
    const bool f32 = a->dtype == ICAF_F32 || a->out_dtype == ICAF_F32;
    const int N = a->Cout;
    const long long M = p.M;
    auto blocks = [&](int t) { return ((M + kTiles[t - 1].bm - 1) / kTiles[t - 1].bm) * ((N + kTiles[t - 1].bn - 1) / kTiles[t - 1].bn) * a->groups; };
    int t;
    if (N > 64 && !f32) t = 1;
    else if (N > 32) t = 2;
    else t = 3;
    // small problems: prefer more, smaller workgroups so every CU gets several
    if (t == 1 && blocks(1) < 512 && blocks(2) > blocks(1)) t = 2;
    if (blocks(t) < 512 && blocks(4) > blocks(t)) t = 4;
    if (!dma_ok) return t + 10;
    const int eb = a->dtype == ICAF_F32 ? 4 : 2;
    if (a->w2) {                                    // chained 1x1: the N tile must hold every channel of both layers
        const int c = a->Cout > a->Cout2 ? a->Cout : a->Cout2;
        const bool w128 = ((long long)a->Cin * eb) % 128 == 0;
        if (c <= 64) return w128 ? 22 : 2;
        return w128 ? 21 : 1;                       // (c > 128 is rejected at launch: W2 would not fit beside the ring)
    }
    if (a->pre) {                                   // only tiles 1 / 2 on pipelines 0 / 2 carry the pre term
        if (t > 2) t = 2;
        return ((long long)a->Cin * eb) % 128 == 0 ? t + 20 : t;
    }
    // full 128-byte lines whenever a pixel's tap (or two adjacent taps) provides them
    return (long long)p.K * eb >= 256 ? t + 20 : t;
}

template <int DT, int ODT, int BM, int BN, int WM, int WN, int ACT, int RB, int NS, int MODE, bool PRE = false, bool CHAIN = false>
static int launch_dma_mode(const ConvP& q, dim3 grid, hipStream_t s) {
    constexpr int ring_only = NS * (BM + BN) * RB, stage_out = PRE ? TileLds<DT, ODT, BM, BN>::PRE_BYTES : TileLds<DT, ODT, BM, BN>::OUT_BYTES;
    constexpr int ring0 = ring_only > stage_out ? ring_only : stage_out;
    constexpr int ring = CHAIN ? (ring0 + 1023) / 1024 * 1024 + BN * BN * 2 : ring0;      // + the chained layer's weights
    static_assert(ring <= 160 * 1024, "LDS capacity");
    ICAF_LDS_OPTIN((igemm_dma_kernel<DT, ODT, BM, BN, WM, WN, ACT, RB, NS, MODE, PRE, CHAIN>), ring);
    igemm_dma_kernel<DT, ODT, BM, BN, WM, WN, ACT, RB, NS, MODE, PRE, CHAIN><<<grid, dim3((BM / WM) * (BN / WN) * 64), ring, s>>>(q);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <int DT, int ODT, int BM, int BN, int WM, int WN, int ACT, int RB, int NS>
static int launch_dma(const ConvP& q, dim3 grid, hipStream_t s) {
    const int eb = DT == ICAF_F32 ? 4 : 2;
    const bool whole_taps = (q.Cin * eb) % RB == 0;          // a K slice never straddles two filter taps
    if (q.w2) {       // chained 1x1 (icaf.h): one N tile of 64 / 128 / 256 channels, tap-uniform K slices, pipelines 0 and 2
        if constexpr (ACT == ICAF_ACT_SILU && ODT == DT && DT != ICAF_F32 && BN >= 64 && NS * RB != 384 &&
                      ((BM / WM) * (BN / WN) == 4 ? BM == 128 : BN == 256)) {
            constexpr int lds_need = (NS * (BM + BN) * RB > TileLds<DT, ODT, BM, BN>::OUT_BYTES ? NS * (BM + BN) * RB : TileLds<DT, ODT, BM, BN>::OUT_BYTES) + 1024 + BN * BN * 2;
            if constexpr (lds_need <= 160 * 1024) {
                if (whole_taps && q.Cout <= BN && q.Cout2 <= BN && !q.pre && (!q.res || q.keep1) &&
                    (!q.keep1 || (q.alpha_acc[0] == 1.0f && q.alpha_acc[1] == 1.0f)))
                    return launch_dma_mode<DT, ODT, BM, BN, WM, WN, ACT, RB, NS, 2, false, true>(q, grid, s);
            }
        }
        return fail(ICAF_ERR_UNSUPPORTED, "icaf_conv2d: the chained 1x1 needs a 16-bit SiLU layer with Cout, Cout2 <= the N tile (64 / 128 / 256), "
                                          "Cin*bytes %% %d == 0, no pre term, a residual only together with chain_keep, on pipelines 0 / 2", RB);
    }
    if (q.pre) {      // pre-activation bilinear term (DMFF fused tail): 1x1 + SiLU on the 128-row tiles only
        if constexpr (ACT == ICAF_ACT_SILU && ODT == DT && BM == 128 && NS * RB != 384) {      // 4- and 8-wavefront 128-row tiles
            if (whole_taps && q.kh == 1 && q.kw == 1 && q.sh == 1 && q.sw == 1 && q.ph == 0 && q.pw == 0)
                return launch_dma_mode<DT, ODT, BM, BN, WM, WN, ACT, RB, NS, 1, true>(q, grid, s);
        }
        return fail(ICAF_ERR_UNSUPPORTED, "icaf_conv2d: `pre` is built for 1x1 / stride 1 SiLU layers with Cin*bytes %% %d == 0 on tiles 128x128 / 128x64 "
                                          "(pipelines 0 and 2; 8-wavefront 128x128 on pipeline 2)", RB);
    }
    if (whole_taps && q.kh == 1 && q.kw == 1 && q.sh == 1 && q.sw == 1 && q.ph == 0 && q.pw == 0)
        return launch_dma_mode<DT, ODT, BM, BN, WM, WN, ACT, RB, NS, 1>(q, grid, s);
    if (whole_taps) return launch_dma_mode<DT, ODT, BM, BN, WM, WN, ACT, RB, NS, 2>(q, grid, s);
    return launch_dma_mode<DT, ODT, BM, BN, WM, WN, ACT, RB, NS, 0>(q, grid, s);
}

template <int DT, int ODT, int BM, int BN, int WM, int WN, int ACT>
static int launch_act(const ConvP& q, dim3 grid, int pipe, hipStream_t s) {
    switch (pipe) {
        case 0: return launch_dma<DT, ODT, BM, BN, WM, WN, ACT, 64, 3>(q, grid, s);
        case 2: return launch_dma<DT, ODT, BM, BN, WM, WN, ACT, 128, 2>(q, grid, s);
        case 3: return launch_dma<DT, ODT, BM, BN, WM, WN, ACT, 128, 3>(q, grid, s);
        default:
            if (q.pre || q.w2) return fail(ICAF_ERR_UNSUPPORTED, "icaf_conv2d: `pre` / chained 1x1 are not built for the register-staged pipeline");
            igemm_kernel<DT, ODT, BM, BN, WM, WN, ACT><<<grid, dim3(NTHREADS), 0, s>>>(q);
            ICAF_LAUNCH_CHECK();
            return ICAF_OK;
    }
}

template <int DT, int ODT, int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvP& p, int groups, int pipe, hipStream_t s) {
    ConvP q = p;
    q.mtiles = (p.M + BM - 1) / BM;
    q.ntiles = (p.Cout + BN - 1) / BN;
    if (pipe != 1) {       // the DMA pipelines walk K in RB-byte slices
        const int eb = DT == ICAF_F32 ? 4 : 2, bk = (pipe == 0 ? 64 : 128) / eb;
        q.nchunks = (p.K + bk - 1) / bk;
    }
    dim3 grid((unsigned)(q.mtiles * q.ntiles), 1, (unsigned)groups);
    if (p.act == ICAF_ACT_SILU) return launch_act<DT, ODT, BM, BN, WM, WN, ICAF_ACT_SILU>(q, grid, pipe, s);
    if (p.act == ICAF_ACT_GELU) return launch_act<DT, ODT, BM, BN, WM, WN, ICAF_ACT_GELU>(q, grid, pipe, s);
    return launch_act<DT, ODT, BM, BN, WM, WN, ICAF_ACT_NONE>(q, grid, pipe, s);
}

// 8-wavefront tiles: pipeline 2 only
template <int DT, int ODT, int BM, int BN, int WM, int WN>
static int launch_big(const ConvP& p, int groups, hipStream_t s) {
    ConvP q = p;
    q.mtiles = (p.M + BM - 1) / BM;
    q.ntiles = (p.Cout + BN - 1) / BN;
    q.nchunks = (p.K + 63) / 64;                  // 128-byte slices of a 16-bit type
    dim3 grid((unsigned)(q.mtiles * q.ntiles), 1, (unsigned)groups);
    if (p.act == ICAF_ACT_SILU) return launch_dma<DT, ODT, BM, BN, WM, WN, ICAF_ACT_SILU, 128, 2>(q, grid, s);
    if (p.act == ICAF_ACT_GELU) return launch_dma<DT, ODT, BM, BN, WM, WN, ICAF_ACT_GELU, 128, 2>(q, grid, s);
    return launch_dma<DT, ODT, BM, BN, WM, WN, ICAF_ACT_NONE, 128, 2>(q, grid, s);
}

template <int DT, int ODT>
static int launch_tile(const ConvP& p, int groups, int cfg, hipStream_t s) {
    const int pipe = cfg / 10;
    switch (cfg % 10) {
        case 1:
            if constexpr (ODT == ICAF_F32) return fail(ICAF_ERR_UNSUPPORTED, "tile 128x128 has no fp32-output build");
            else return launch_cfg<DT, ODT, 128, 128, 64, 64>(p, groups, pipe, s);
        case 2: return launch_cfg<DT, ODT, 128, 64, 64, 32>(p, groups, pipe, s);
        case 3: return launch_cfg<DT, ODT, 256, 32, 64, 32>(p, groups, pipe, s);
        case 4: return launch_cfg<DT, ODT, 64, 64, 32, 32>(p, groups, pipe, s);
        case 8:                                       // 128x128 / 128x64 tiles with 8 wavefronts (64x32 / 32x32 each): more waves
        case 9:                                       // per SIMD to hide latency, same LDS footprint as the 4-wave tiles
            if constexpr (DT == ICAF_F32 || ODT == ICAF_F32) return fail(ICAF_ERR_UNSUPPORTED, "8-wavefront 128-row tiles exist for 16-bit types only");
            else {
                if (pipe != 2 || p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "8-wavefront 128-row tiles run on pipeline 2 only (ids 28 / 29)");
                return cfg % 10 == 8 ? launch_big<DT, ODT, 128, 128, 64, 32>(p, groups, s) : launch_big<DT, ODT, 128, 64, 32, 32>(p, groups, s);
            }
        case 5:
        case 6:
            if constexpr (DT == ICAF_F32 || ODT == ICAF_F32) return fail(ICAF_ERR_UNSUPPORTED, "tiles 256x128 / 256x256 exist for 16-bit types only");
            else {
                if (pipe != 2 || p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "tiles 256x128 / 256x256 run on the 128-byte LDS-DMA pipeline only (id 25 / 26)");
                return cfg % 10 == 5 ? launch_big<DT, ODT, 256, 128, 64, 64>(p, groups, s) : launch_big<DT, ODT, 256, 256, 128, 64>(p, groups, s);
            }
        default: return fail(ICAF_ERR_ARG, "unknown tile id %d", cfg);
    }
}

static int validate(const icaf_conv_args* a) {
    if (!a || !a->x || !a->w || !a->y) return fail(ICAF_ERR_ARG, "icaf_conv2d: null pointer");
    if (a->groups < 1 || a->groups > 2) return fail(ICAF_ERR_ARG, "icaf_conv2d: groups must be 1 or 2");
    if (a->tile < 0 || a->tile >= 90) return fail(ICAF_ERR_ARG, "icaf_conv2d: unknown launch configuration %d", a->tile);
    const int vec = a->dtype == ICAF_F32 ? 4 : 8;
    if (a->dtype < 0 || a->dtype > 2) return fail(ICAF_ERR_ARG, "icaf_conv2d: bad dtype %d", a->dtype);
    if (a->out_dtype != a->dtype && a->out_dtype != ICAF_F32) return fail(ICAF_ERR_ARG, "icaf_conv2d: out_dtype must equal dtype or be fp32");
    if (a->Cin % vec || a->ldx % vec) return fail(ICAF_ERR_ARG, "icaf_conv2d: Cin (%d) and ldx (%d) must be multiples of %d", a->Cin, a->ldx, vec);
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->w & 15)) return fail(ICAF_ERR_ARG, "icaf_conv2d: x / w must be 16-byte aligned");
    if (a->B < 1 || a->Ho < 1 || a->Wo < 1 || a->Cout < 1 || a->kh < 1 || a->kw < 1 || a->sh < 1 || a->sw < 1)
        return fail(ICAF_ERR_ARG, "icaf_conv2d: bad geometry");
    if (a->Ho != (a->H + 2 * a->ph - a->kh) / a->sh + 1 || a->Wo != (a->W + 2 * a->pw - a->kw) / a->sw + 1)
        return fail(ICAF_ERR_ARG, "icaf_conv2d: output size does not match input/kernel/stride/padding");
    const int ka = a->dtype == ICAF_F32 ? 32 : 64;
    if (a->Kp % ka || a->Kp < a->kh * a->kw * a->Cin) return fail(ICAF_ERR_ARG, "icaf_conv2d: Kp=%d must be a multiple of %d covering K=%d", a->Kp, ka, a->kh * a->kw * a->Cin);
    if (a->act < 0 || a->act > 2) return fail(ICAF_ERR_ARG, "icaf_conv2d: bad activation code %d", a->act);
    if (a->ldy < a->Cout) return fail(ICAF_ERR_ARG, "icaf_conv2d: ldy < Cout");
    if (a->res && a->ldr < a->Cout) return fail(ICAF_ERR_ARG, "icaf_conv2d: ldr < Cout");
    if ((long long)a->B * a->Ho * a->Wo > 0x7fffffffLL) return fail(ICAF_ERR_ARG, "icaf_conv2d: too many output pixels");
    if (a->w2 && (!a->y2 || a->Cout2 < 1 || a->ldy2 < a->Cout2 || a->Kp2 < a->Cout || (a->Kp2 * (a->dtype == ICAF_F32 ? 4 : 2)) % 128 || ((uintptr_t)a->w2 & 15)))
        return fail(ICAF_ERR_ARG, "icaf_conv2d: chained 1x1 needs y2, ldy2 >= Cout2 >= 1, Kp2 >= Cout in whole 128-byte slices, aligned w2");
    if (a->x2 && (!a->w2 || a->dtype == ICAF_F32 || a->Kp2 < 2 * a->Cout || a->ldx2 < a->Cout))
        return fail(ICAF_ERR_ARG, "icaf_conv2d: x2 (the second half of the chained 1x1's input) needs w2, a 16-bit type, Kp2 >= 2 Cout and ldx2 >= Cout");
    if (a->pre && (a->pre_mode < 0 || a->pre_mode > 1)) return fail(ICAF_ERR_ARG, "icaf_conv2d: pre_mode must be 0 (bilinear) or 1 (nearest)");
    if (a->pre && (a->pre_h < 1 || a->pre_w < 1 || a->ldpre < a->Cout || (a->ldpre & 3) || ((uintptr_t)a->pre & 15) || a->groups != 1))
        return fail(ICAF_ERR_ARG, "icaf_conv2d: pre needs pre_h, pre_w >= 1, ldpre >= Cout and a multiple of 4, 16-byte alignment, groups == 1");
    return ICAF_OK;
}

static void fill(const icaf_conv_args* a, ConvP& p) {
    p.x = a->x; p.w = a->w; p.bias = a->bias; p.y = a->y; p.res = a->res;
    p.x_gs = a->x_gs; p.w_gs = a->w_gs; p.bias_gs = a->bias_gs; p.y_gs = a->y_gs; p.res_gs = a->res_gs;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.ldx = a->ldx; p.Ho = a->Ho; p.Wo = a->Wo; p.Cout = a->Cout;
    p.ldy = a->ldy; p.kh = a->kh; p.kw = a->kw; p.sh = a->sh; p.sw = a->sw; p.ph = a->ph; p.pw = a->pw; p.ldr = a->ldr;
    p.Kp = a->Kp; p.act = a->act;
    p.x2 = a->x2; p.x2_gs = a->x2_gs; p.ldx2 = a->ldx2;      // (C3 tail, cwide.hip; icaf_bottleneck overwrites them with its own)
    p.M = a->B * a->Ho * a->Wo;
    p.K = a->kh * a->kw * a->Cin;
    const int bk = a->dtype == ICAF_F32 ? 16 : 32;
    p.nchunks = (p.K + bk - 1) / bk;                  // 64-byte slices (launch_cfg recomputes it for 128-byte slices)
    p.mtiles = p.ntiles = 0;
    const int vo = a->out_dtype == ICAF_F32 ? 4 : 8, vi = a->dtype == ICAF_F32 ? 4 : 8;
    const int yb = a->out_dtype == ICAF_F32 ? 4 : 2, eb = a->dtype == ICAF_F32 ? 4 : 2;
    p.vec_y = (a->ldy % vo == 0) && (((uintptr_t)a->y & 15) == 0) && ((a->y_gs * yb) % 16 == 0);
    p.vec_r = a->res && (a->ldr % vi == 0) && (((uintptr_t)a->res & 15) == 0) && ((a->res_gs * eb) % 16 == 0);
    // buffer-descriptor ranges for the DMA pipeline: the x view spans ((pixels-1)*ldx + Cin) elements, the packed
    // weights Np x Kp; both must stay below 2 GiB so that offset 0x80000000 is always out of range (reads as zero)
    const long long xb = (((long long)a->B * a->H * a->W - 1) * a->ldx + a->Cin) * eb;
    const long long np = ((long long)a->Cout + 127) / 128 * 128;
    const long long wb = np * a->Kp * eb;
    if (xb < 0x7fffff00LL && wb < 0x7fffff00LL) { p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb; }
    else { p.x_bytes = 0; p.w_bytes = 0; }
    for (int i = 0; i < 2; ++i) { p.alpha_acc[i] = a->alpha_acc[i]; p.alpha_res[i] = a->alpha_res[i]; }
    p.pre = a->pre; p.pre_h = a->pre_h; p.pre_w = a->pre_w; p.ldpre = a->ldpre; p.pre_mode = a->pre_mode;
    p.w1 = nullptr; p.bias1 = nullptr; p.w1_gs = p.bias1_gs = 0; p.Kp1 = 0; p.w1_bytes = 0;
    p.w2 = a->w2; p.bias2 = a->bias2; p.y2 = a->y2; p.w2_gs = a->w2_gs; p.bias2_gs = a->bias2_gs; p.y2_gs = a->y2_gs;
    p.Kp2 = a->Kp2; p.Cout2 = a->Cout2; p.ldy2 = a->ldy2; p.keep1 = a->chain_keep;
    p.vec_y2 = a->w2 && (a->ldy2 % vo == 0) && (((uintptr_t)a->y2 & 15) == 0) && ((a->y2_gs * yb) % 16 == 0);
    p.w2_bytes = a->w2 ? (unsigned)((((long long)a->Cout2 + 127) / 128 * 128) * a->Kp2 * eb) : 0;
}

// validate + fill for the other translation units that run a convolution through their own kernel (detect.hip)
int conv_prepare(const icaf_conv_args* a, ConvP& p) {
    int st = validate(a);
    if (st) return st;
    fill(a, p);
    return ICAF_OK;
}

}  // namespace icaf

using namespace icaf;

extern "C" int icaf_conv2d(const icaf_conv_args* a, icaf_stream_t s) {
    int st = validate(a);
    if (st) return st;
    ConvP p;
    fill(a, p);
    const int tile = pick_tile(a, p);
    hipStream_t hs = S(s);
    if (a->x2 && tile != 81 && tile != 82) return fail(ICAF_ERR_UNSUPPORTED, "icaf_conv2d: x2 (C3 tail) is built for launch configurations 81 / 82 only (tile %d)", tile);
    if (tile == 71) return launch_cstream(a, p, hs);        // persistent 3x3 with the filter resident in LDS (64 -> 64 channels)
    if (tile > 80) return launch_cwide(a, p, tile - 80, hs);  // 3x3 (stride 1 / 2) from a resident halo patch, weights streamed into registers
    if (tile > 70) return fail(ICAF_ERR_ARG, "unknown tile id %d", tile);
    if (tile > 60) return launch_wreg(a, p, tile - 60, hs);
    if (tile > 50) return launch_stream(a, p, tile - 50, hs);
    if (tile > 40) return launch_ctile(a, p, tile - 40, hs);
    if (a->dtype == ICAF_BF16)
        return a->out_dtype == ICAF_F32 ? launch_tile<ICAF_BF16, ICAF_F32>(p, a->groups, tile, hs)
                                        : launch_tile<ICAF_BF16, ICAF_BF16>(p, a->groups, tile, hs);
    if (a->dtype == ICAF_F16)
        return a->out_dtype == ICAF_F32 ? launch_tile<ICAF_F16, ICAF_F32>(p, a->groups, tile, hs)
                                        : launch_tile<ICAF_F16, ICAF_F16>(p, a->groups, tile, hs);
    return launch_tile<ICAF_F32, ICAF_F32>(p, a->groups, tile, hs);
}

extern "C" int icaf_bottleneck(const icaf_bneck_args* b, icaf_stream_t s) {
    if (!b || !b->w1) return fail(ICAF_ERR_ARG, "icaf_bottleneck: null pointer");
    const icaf_conv_args* a = &b->conv;
    int st = validate(a);
    if (st) return st;
    const int eb = a->dtype == ICAF_F32 ? 4 : 2;
    if (b->Kp1 * eb != 128) return fail(ICAF_ERR_ARG, "icaf_bottleneck: the packed 1x1 weights must have 128-byte rows (Kp1 = %d)", b->Kp1);
    if (a->pre) return fail(ICAF_ERR_ARG, "icaf_bottleneck: no pre-activation term");
    if (a->x2) return fail(ICAF_ERR_ARG, "icaf_bottleneck: conv.x2 must be NULL (the block's cv2 half is icaf_bneck_args.x2)");
    if (a->x == a->y) return fail(ICAF_ERR_ARG, "icaf_bottleneck: in-place operation is not possible (neighbouring patches read x)");
    ConvP p;
    fill(a, p);
    p.w1 = b->w1; p.bias1 = b->bias1; p.w1_gs = b->w1_gs; p.bias1_gs = b->bias1_gs; p.Kp1 = b->Kp1;
    p.w1_bytes = (unsigned)((((long long)a->Cout + 127) / 128 * 128) * b->Kp1 * eb);
    p.x2 = b->x2; p.x2_gs = b->x2_gs; p.ldx2 = b->ldx2;
    return launch_bneck(a, p, b->shape, S(s));
}

extern "C" int icaf_conv2d_kernel_name(const icaf_conv_args* a, char* buf, int buf_len) {
    int st = validate(a);
    if (st) return st;
    ConvP p;
    fill(a, p);
    const int tile = pick_tile(a, p);
    if (a->x2 && tile != 81 && tile != 82) return fail(ICAF_ERR_UNSUPPORTED, "icaf_conv2d: x2 (C3 tail) is built for launch configurations 81 / 82 only (tile %d)", tile);
    static const char* dn[] = {"f32", "bf16", "f16"};
    if (tile == 71) {
        st = cstream_check(a, p);
        if (st) return st;
        snprintf(buf, buf_len, "cstream_%s_8x16n64", dn[a->dtype]);
        return ICAF_OK;
    }
    if (tile > 80) {
        st = cwide_check(a, p, tile - 80);
        if (st) return st;
        snprintf(buf, buf_len, "cwide_%s_%s", dn[a->dtype], cwide_tag(tile - 80));
        return ICAF_OK;
    }
    if (tile > 70) return fail(ICAF_ERR_ARG, "unknown tile id %d", tile);
    if (tile > 60) {
        st = wreg_check(a, p, tile - 60);
        if (st) return st;
        snprintf(buf, buf_len, "igemm_wreg_%s_%s", dn[a->dtype], wreg_tag(tile - 60));
        return ICAF_OK;
    }
    if (tile > 50) {
        st = stream_check(a, p, tile - 50);
        if (st) return st;
        snprintf(buf, buf_len, "igemm_stream_%s_%s", dn[a->dtype], stream_tag(tile - 50));
        return ICAF_OK;
    }
    if (tile > 40) {
        st = ctile_check(a, p, tile - 40);
        if (st) return st;
        snprintf(buf, buf_len, "ctile_%s_%s", dn[a->dtype], ctile_tag(tile - 40));
        return ICAF_OK;
    }
    static const char* pn[] = {"_dma64x3", "_reg", "_dma128x2", "_dma128x3"};
    snprintf(buf, buf_len, "igemm%s_%s_%s_%s", pn[tile / 10], dn[a->dtype], dn[a->out_dtype], kTiles[tile % 10 - 1].tag);
    return ICAF_OK;
}
