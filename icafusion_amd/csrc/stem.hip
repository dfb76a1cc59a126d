// Stem of both backbones for gfx950 (MI355X): input staging + the 6x6 / stride 2 / pad 2 convolution (+BN+SiLU) of yaml
// rows 0 and 10 in ONE persistent kernel (reference: detect_twostream.py:70-80 / test.py:116-123 for the staging,
// models/common.py:48-60 for the layer).
//
// The two-kernel form (icaf_preprocess_* then a 3x3 convolution over the space-to-depth tensor) writes and re-reads a
// 16-channel bf16 copy of the images (2 x 105 MB at batch 32) and spends most of its time in workgroup prologues: the
// layer has K = 144 and 25600 output patches.  Here
//   * a workgroup is PERSISTENT: it owns the folded weights (Cout x 144, resident in LDS for its whole life) and walks
//     patches p = blockIdx.x, += gridDim.x of 8 x 32 output pixels;
//   * the 10 x 34 space-to-depth halo patch of patch p+1 is fetched straight from the NCHW image (fp32, or the
//     dataloader's uint8 6-channel batch with the /255 fused) into REGISTERS while patch p is being multiplied, then
//     converted and written to the other LDS patch buffer — the image is read once, nothing intermediate reaches HBM;
//   * the arithmetic is ctile.hip's: same swizzled patch layout, same (tap, channel) K order and MFMA step, so the
//     result is bit-identical to the two-kernel form (tested);
//   * the epilogue is the shared one (bias + SiLU in registers, LDS-staged 16-byte NHWC stores).
#include "conv_common.h"
#include <cstring>
#ifdef ICAF_S2_CLK
#include <cstdlib>          // (probe builds only: lab/probes/stem2_phases.py)
#endif

namespace icaf {

constexpr int ST_TH = 8, ST_TW = 32, ST_HH = ST_TH + 2, ST_HW = ST_TW + 2, ST_NIDX = ST_HH * ST_HW;      // 340 entries
constexpr int ST_PATCH_BYTES = ((ST_NIDX * 2 + 63) / 64) * 1024;                                        // 11 KiB

struct StemP {
    const void* img;            // fp32 [nstreams*B][3][H][W]   or   uint8 [B][ctot][H][W]
    int ctot;                   // uint8: channels per image (6); stream s reads channels [3s, 3s+3)
    int B, H, W, nstreams;
    int tiles_x, tiles_y, npatch;
    ConvP c;                    // y, bias, w, strides, Ho, Wo, Cout, ldy, Kp, alphas (the epilogue's view of the layer)
};

// One space-to-depth entry = 2x2 pixels x 3 channels.  The six 2-pixel loads are issued UNCONDITIONALLY from
// coordinates clamped into the image and kept raw: no control flow or conversion depends on them, so they stay in
// flight until finish_entry() — called one tile later — turns them into the 12 values (zero outside the image).
template <bool U8> struct RawEntry;
template <> struct RawEntry<false> { float2 r[6]; bool in; };
template <> struct RawEntry<true> { unsigned short r[6]; bool in; };

template <bool U8, class P>
__device__ __forceinline__ void load_entry(const P& q, int stream, int b, int gy, int gx, RawEntry<U8>& e) {
    const int hh = q.H >> 1, hw = q.W >> 1;
    e.in = (unsigned)gy < (unsigned)hh && (unsigned)gx < (unsigned)hw;
    const int cy = min(max(gy, 0), hh - 1), cx = min(max(gx, 0), hw - 1);
    const long long plane = (long long)q.H * q.W;
    // wave-uniform plane base (scalar registers) + a 32-bit lane offset (the host checks that a plane is < 2 GiB)
    const unsigned off = (unsigned)(2 * cy) * (unsigned)q.W + (unsigned)(2 * cx);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if constexpr (U8) {                                                    // W even, 2*cx even: 2-byte aligned
            const unsigned char* base = (const unsigned char*)q.img + ((long long)b * q.ctot + 3 * stream + c) * plane;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) e.r[c * 2 + dy] = *(const unsigned short*)(base + off + (unsigned)(dy * q.W));
        } else {                                                               // 8-byte aligned
            const float* base = (const float*)q.img + ((long long)(stream * q.B + b) * 3 + c) * plane;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) e.r[c * 2 + dy] = *(const float2*)(base + off + (unsigned)(dy * q.W));
        }
    }
}

// Walk of a persistent workgroup over tiles p, p + stride, ... of a [image][tile row][tile column] grid without a
// division per tile: position and stride are kept as mixed-radix digits and added with carries (all wave-uniform).
struct TileWalk {
    int img, ty, tx, s_img, s_ty, s_tx, tiles_x, tiles_y;
    __device__ __forceinline__ void init(int pt, int stride, int tiles_x_, int tiles_y_) {
        tiles_x = tiles_x_; tiles_y = tiles_y_;
        const int per_img = tiles_x * tiles_y;
        img = pt / per_img;
        int r = pt - img * per_img;
        ty = r / tiles_x;
        tx = r - ty * tiles_x;
        s_img = stride / per_img;
        r = stride - s_img * per_img;
        s_ty = r / tiles_x;
        s_tx = r - s_ty * tiles_x;
    }
    __device__ __forceinline__ void next() {
        tx += s_tx;
        if (tx >= tiles_x) { tx -= tiles_x; ++ty; }
        ty += s_ty;
        if (ty >= tiles_y) { ty -= tiles_y; ++img; }
        img += s_img;
    }
};

template <bool U8>
__device__ __forceinline__ void finish_entry(const RawEntry<U8>& e, float (&v)[12]) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            float a0, a1;
            if constexpr (U8) {
                const unsigned short two = e.r[c * 2 + dy];
                a0 = (float)(two & 0xff) / 255.0f;
                a1 = (float)(two >> 8) / 255.0f;
            } else {
                a0 = e.r[c * 2 + dy].x;
                a1 = e.r[c * 2 + dy].y;
            }
            v[(dy * 2 + 0) * 3 + c] = e.in ? a0 : 0.0f;                        // channel order (dy, dx, c) of ops.s2d_conv_weight
            v[(dy * 2 + 1) * 3 + c] = e.in ? a1 : 0.0f;
        }
}

template <int DT, bool U8>
__device__ __forceinline__ void store_entry(unsigned char* patch, int idx, const RawEntry<U8>& e) {
    float v[12], lo[8], hi4[8];
    finish_entry<U8>(e, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { lo[i] = v[i]; hi4[i] = i < 4 ? v[8 + i] : 0.0f; }
    const int key = (idx >> 3) & 1;                                            // ctile's swizzle for 2 slots per pixel
    *(u32x4*)(patch + (((idx << 1) + (0 ^ key)) << 4)) = pack16<DT>(lo);
    *(u32x4*)(patch + (((idx << 1) + (1 ^ key)) << 4)) = pack16<DT>(hi4);
}

template <int DT, int BN, bool U8>
__global__ __launch_bounds__(NTHREADS) void stem_kernel(const StemP q) {
    using E = Elem<DT>;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int RB = 128, TM = 2, TN = BN / 32, BM = ST_TH * ST_TW, WM = BM / 4;
    constexpr int NBW = BN / 32;
    constexpr int NSLICE = 3;                                                  // Kp = 192 elements = 3 slices of 128 bytes
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* patch0 = lds;
    unsigned char* patch1 = lds + ST_PATCH_BYTES;
    unsigned char* wbuf = lds + 2 * ST_PATCH_BYTES;                            // NSLICE x BN rows x 128 bytes
    unsigned char* stage = wbuf + NSLICE * BN * RB;                            // epilogue staging (aliasing it with the idle
    // patch buffers would allow 4 instead of 2 workgroups per CU but costs two barriers per patch: measured 288 vs 270 us)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ConvP& p = q.c;

    auto decode = [&](const TileWalk& t, int& stream, int& b, int& y0, int& x0) {
        stream = t.img >= q.B;                                                 // (at most two streams)
        b = t.img - (stream ? q.B : 0);
        y0 = t.ty * ST_TH;
        x0 = t.tx * ST_TW;
    };
    // each thread stages entries tid and tid + 256 of a patch
    const int hy0 = tid / ST_HW, hx0 = tid - hy0 * ST_HW;
    const int i1 = min(tid + NTHREADS, ST_NIDX - 1);                           // (threads past the end re-load the last entry)
    const int hy1 = i1 / ST_HW, hx1 = i1 - hy1 * ST_HW;
    auto fetch = [&](const TileWalk& t, RawEntry<U8>& v0, RawEntry<U8>& v1) {
        int stream, b, y0, x0;
        decode(t, stream, b, y0, x0);
        load_entry<U8>(q, stream, b, y0 - 1 + hy0, x0 - 1 + hx0, v0);
        load_entry<U8>(q, stream, b, y0 - 1 + hy1, x0 - 1 + hx1, v1);
    };
    auto commit = [&](unsigned char* patch, const RawEntry<U8>& v0, const RawEntry<U8>& v1) {
        store_entry<DT, U8>(patch, tid, v0);
        if (tid + NTHREADS < ST_NIDX) store_entry<DT, U8>(patch, tid + NTHREADS, v1);
    };

    auto load_weights = [&](int stream) {
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const typename E::type*)p.w + stream * p.w_gs), 0, p.w_bytes, 0x00020000);
        const int rsub = lane >> 3, dkey = ((wave & 1) << 2) | (rsub >> 1), lslot = (lane & 7) ^ dkey;
        const unsigned w_off0 = ((unsigned)(wave * 8 + rsub) * (unsigned)p.Kp + (unsigned)(lslot * E::VEC)) * E::BYTES;
#pragma unroll
        for (int c = 0; c < NSLICE; ++c)
#pragma unroll
            for (int i = 0; i < NBW; ++i) {
                // (offset in a variable: a template-dependent expression passed inline as the builtin's offset operand
                // makes clang's host pass silently drop the kernel's stub)
                const unsigned voff = w_off0 + (unsigned)c * RB + (unsigned)(32 * i) * (unsigned)p.Kp * E::BYTES;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(wbuf + c * BN * RB + (wave + 4 * i) * 1024), 16, voff, 0, 0, 0);
            }
        wait_vmcnt<0>();
    };

    // XCD-aware patch walk: workgroups are dispatched round-robin over the 8 XCDs, each with a private L2.  XCD x walks
    // the x-th contiguous eighth of the patches, so patches that share halo rows / columns (and the 128-byte lines
    // their 8-byte-misaligned rows straddle) meet in ONE L2 instead of being fetched from HBM once per XCD.
    const int xcd = blockIdx.x & 7, per_xcd = (q.npatch + 7) >> 3, pstride = gridDim.x >> 3;
    const int pend = min((xcd + 1) * per_xcd, q.npatch);
    int pt = xcd * per_xcd + (blockIdx.x >> 3);
    if (pt >= pend) return;
    TileWalk cur_t, nxt_t;
    cur_t.init(pt, pstride, q.tiles_x, q.tiles_y);
    RawEntry<U8> v0, v1;
    fetch(cur_t, v0, v1);

    // fragment / lane constants (as ctile.hip, TW = 32, stride 1, 2 slots per pixel)
    int lbase[TM];
#pragma unroll
    for (int bb = 0; bb < TM; ++bb) lbase[bb] = (wave * TM + bb) * ST_HW + l31;
    const int fkey = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);

    int cur = 0, wstream = -1;
    commit(patch0, v0, v1);
    while (true) {
        int stream, b, y0, x0;
        decode(cur_t, stream, b, y0, x0);
        if (stream != wstream) {                   // (re)load this stream's weights (at most twice in a workgroup's life)
            __syncthreads();                       // nobody still reads the previous weights
            load_weights(stream);
            wstream = stream;
        }
        const int pn = pt + pstride;
        const bool more = pn < pend;
        nxt_t = cur_t;
        nxt_t.next();
        if (more) fetch(nxt_t, v0, v1);            // next patch's image reads stay in flight during the MFMAs below
        lds_barrier();                             // current patch (and weights) visible
        const unsigned char* patch = cur ? patch1 : patch0;
        f32x16 acc[TN][TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int bb = 0; bb < TM; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {        // K = (tap, 16 channels): one MFMA step per tap
            const int ky = tap / 3, kx = tap - 3 * ky, toff = ky * ST_HW + kx;
            u32x4 fp[TM], fw[TN];
#pragma unroll
            for (int bb = 0; bb < TM; ++bb) {
                const int idx = lbase[bb] + toff;
                fp[bb] = *(const u32x4*)(patch + (((idx << 1) + (hi ^ ((idx >> 3) & 1))) << 4));
            }
#pragma unroll
            for (int a = 0; a < TN; ++a) fw[a] = *(const u32x4*)(wbuf + (tap >> 2) * BN * RB + (a * 32) * RB + foff[tap & 3]);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int bb = 0; bb < TM; ++bb) mma_step<DT>(acc[a][bb], fw[a], fp[bb]);
        }
        epilogue<DT, DT, BM, BN, WM, BN, ICAF_ACT_SILU, false>(acc, stage, p, stream, [&](int row) {
            const int gy = y0 + (row >> 5), gx = x0 + (row & 31);
            return (gy < p.Ho && gx < p.Wo) ? (b * p.Ho + gy) * p.Wo + gx : -1;
        }, 0);
        if (!more) break;
        commit(cur ? patch0 : patch1, v0, v1);     // (every wave passed the epilogue's barrier: nobody reads that buffer)
        cur ^= 1;
        pt = pn;
        cur_t = nxt_t;
    }
}


// ===============================================================================================================
// stem2: yaml rows 0, 1 and the cv1 | cv2 GEMM of row 2 (10, 11, 12) in one persistent kernel (icaf.h: icaf_stem2)
// ===============================================================================================================
// At batch 32 the stem's output (320 x 320 x 32 channels per image and stream) is the largest tensor of the network:
// writing it and reading it back for the next layer is 840 MB of the ~1.5 GB the first two launches move.  Here a
// workgroup of 8 wavefronts owns a 4 x 32 tile of the SECOND convolution's output pixels and never lets t0 / t1 leave LDS:
//   stage 0  the 11 x 67 space-to-depth halo patch of the tile is fetched from the NCHW image into registers one tile
//            ahead (as the stem kernel does), then written to LDS as even | odd column planes: the stem taps of
//            consecutive lanes (stem pixels two image-columns apart) then read consecutive entries;
//   stage 1  stem GEMM over the 9 x 65 stem pixels the tile needs (K = 9 taps x 16, the stem kernel's arithmetic),
//            bias + SiLU, zero outside the stem's output (= the next layer's zero padding), rounded to the storage type
//            and written as ctile.hip's stride-2 halo patch (64 bytes per pixel, swizzled);
//   stage 2  ctile's 3x3 / stride 2 loop over that patch (K = 288, weights resident in LDS), bias + SiLU, the rounded
//            tile staged in LDS exactly as igemm's CHAIN stages it;
//   stage 3  the chained 1x1 (K = 64) and the shared epilogue writing only y.
// Every stage repeats the K order, MFMA step and rounding points of the kernel it replaces, so y is bit-identical to
// icaf_stem -> icaf_conv2d(chained) (tested).
//
// Round 6, what bounds it (lab/probes/valu_rate_probe.hip, stem2_phases.py, -DICAF_S2_FAKEJOBS): instruction ISSUE.  One wave issues
// a plain VALU instruction every 5.4 cycles and a transcendental every 8.75; two waves per SIMD (all this kernel's 247 registers allow) reach 2.7 / 6.5 per
// SIMD, four would reach 1.8 / 4.8.  The launch issues ~88 k instructions per SIMD (57.5 M VALU of which 14.3 M v_exp / v_rcp — SiLU's two
// transcendentals per value — 13.8 M SALU, 9.1 M LDS, 4.4 M MFMA per launch): ~320 k cycles at those rates against 470 k measured.  A producer /
// consumer split of the waves (stage 1 on waves 0-3, stages 2 + 3 on waves 4-7, two halo patches, two barriers per tile instead of five) was built,
// bit-identical on first run, and measured SLOWER (352 vs 330 us alone): with 19 stage-1 jobs on four waves a producer wave issues five jobs back to
// back at ~1,900 cycles each and the tile waits for it — the work per SIMD is the same ~9 SiLU units either way, and the two-barrier schedule only
// exchanged barrier waits for a longer critical wave.  (commit f26bb5f: "stem2, producer / consumer form".)  LDS: 28 KiB + 45 KiB patches + 60 KiB weights = 133 KiB, one workgroup
// per CU; the patch walk is XCD-aware like the stem's.
constexpr int S2_TH = 4, S2_TW = 32;                                   // tile of the 3x3/s2 layer's output pixels
constexpr int S2_HH = 2 * S2_TH + 1, S2_HWD = 2 * S2_TW + 1;           // stem-output halo patch: 9 x 65 pixels
// Round 4 (late): both patches are laid out so that the tap offsets of the two MFMA loops are multiples of 16 entries: the XOR-swizzle keys
// ((entry >> 3) & 1 for the 32-byte entries, (entry >> 2) & 3 for the 64-byte ones) then do not change from tap to tap, a lane's fragment
// addresses are a handful of tile-invariant registers + compile-time immediates (precomputed once per workgroup), and the K loops carry no
// address arithmetic.  SQ counters had put this kernel at 0.54 VALU issue against 0.21 MFMA busy: 758 vector instructions per wave and tile,
// of which only ~350 are the three SiLU layers — the rest was index decode (divisions by the pitch) and swizzle arithmetic per fragment read.
constexpr int S2_LHALF = (S2_HWD + 1) / 2, S2_LPITCH = 2 * S2_LHALF;   // LOGICAL enumeration of the stem pixels of a tile: 33 even | 33 odd columns per row
constexpr int S2_NL = S2_HH * S2_LPITCH;                               // 594 logical entries = 19 sub-tile jobs of 32
constexpr int S2_HALF = 48, S2_PITCH = 80;                             // PHYSICAL: even columns at [0, 33), odd columns at [48, 80) of an 80-entry row
constexpr int S2_NH = S2_HH * S2_PITCH;                                // 720 entries of 64 bytes
constexpr int S2_HALO_BYTES = ((S2_NH * 4 + 63) / 64) * 1024;          // 45 KiB
constexpr int S2_SR = S2_HH + 2, S2_SC = S2_HWD + 2;                   // space-to-depth patch under it: 11 x 67
constexpr int S2_SPAIRS = (S2_SC + 1) / 2;                             // 34 column pairs per row (the PAIR staging's thread map)
constexpr int S2_SHALF = 40, S2_SPITCH = 80;                           // PHYSICAL: even columns at [0, 34), odd at [40, 74) of an 80-entry row
constexpr int S2_NS = S2_SR * S2_SPITCH;                               // 880 entries of 32 bytes
constexpr int S2_S2D_BYTES = ((S2_NS * 2 + 63) / 64) * 1024;           // 28 KiB
static_assert(S2_PITCH % 16 == 0 && S2_HALF % 16 == 0 && S2_SPITCH % 16 == 0, "tap offsets must keep the swizzle keys");
static_assert(S2_LHALF <= S2_HALF && S2_HALF + S2_HWD / 2 - 1 < S2_PITCH && S2_SPAIRS <= S2_SHALF && S2_SHALF + S2_SPAIRS <= S2_SPITCH, "planes fit their rows");
static_assert(S2_NS <= 2 * 512, "two space-to-depth entries per thread");
constexpr int S2_THREADS = 512;
constexpr int S2_C0 = 32, S2_C1 = 64, S2_C2 = 64, S2_W1_SLICES = 5;    // K1 = 288 -> 5 slices of 128 bytes
constexpr int S2_LDS = S2_S2D_BYTES + S2_HALO_BYTES + (3 * S2_C0 + S2_W1_SLICES * S2_C1 + S2_C2) * 128 + (S2_C1 + S2_C2) * 4;   // + the bias vectors of stages 2 / 3

struct Stem2P {
    const void* img;            // as StemP
    int ctot;
    int B, H, W, nstreams;
    int tiles_x, tiles_y, npatch;
    int Hs, Ws;                 // the stem's output size (H/2, W/2)
#ifdef ICAF_S2_CLK
    unsigned long long* clk;    // [wave 0..7][tile 0..S2_CLK_TILES)[S2_CLK_N] s_memtime stamps of workgroup 0
#endif
    const void* w0; const float* bias0; long long w0_gs, bias0_gs; int Kp0;
    ConvP c;                    // w / bias / Kp: the 3x3 layer;  w2 / bias2 / y2 / ldy2 / Cout2: the 1x1;  Ho, Wo: output
};

// PAIR (fp32 images, W % 4 == 0): a thread stages TWO horizontally adjacent space-to-depth entries — 4 consecutive image pixels = one
// 16-byte load per (channel, row) instead of two 8-byte loads: 6 vector-memory instructions for 374 threads instead of 12 for 512.
// (Phase clocks of the 3x3 kernels put an issued vector-memory instruction at ~200 cycles of a wave's time beside MFMA work; the
// prefetch + commit of a tile was 19 % of this kernel.)
#ifdef ICAF_S2_CLK
constexpr int S2_CLK_TILES = 16, S2_CLK_N = 12;
#define S2_STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && clk_tile < S2_CLK_TILES) q.clk[(wave * S2_CLK_TILES + clk_tile) * S2_CLK_N + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S2_STAMP(i) do { } while (0)
#endif
template <int DT, bool U8, bool PAIR>
__global__ __launch_bounds__(S2_THREADS) void stem2_kernel(const Stem2P q) {
    using E = Elem<DT>;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int RB = 128, C0 = S2_C0, C1 = S2_C1, C2 = S2_C2;
    constexpr int SO = C1 * E::BYTES + 16;                                  // staged-tile row stride (the epilogue's)
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* s2d = lds;                                              // stage 0/1 patch; later the staged t1 tile
    unsigned char* halo = lds + S2_S2D_BYTES;                              // stage 1/2 patch; later the output staging
    unsigned char* w0b = halo + S2_HALO_BYTES;                             // 3 slices x C0 rows x 128 bytes
    unsigned char* w1b = w0b + 3 * C0 * RB;                                // 5 slices x C1 rows
    unsigned char* w2b = w1b + S2_W1_SLICES * C1 * RB;                     // 1 slice x C2 rows
    float* bl1 = (float*)(w2b + C2 * RB);                                  // bias of the 3x3 layer (C1 floats), then of the 1x1 (C2 floats): read back
    float* bl2 = bl1 + C1;                                                 // per tile (32 registers fewer than holding them; the stem's stays in registers)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;                               // output row of the tile / channel half
    const ConvP& p = q.c;

    auto decode = [&](const TileWalk& t, int& stream, int& b, int& y0, int& x0) {
        stream = t.img >= q.B;                                                 // (at most two streams)
        b = t.img - (stream ? q.B : 0);
        y0 = t.ty * S2_TH;
        x0 = t.tx * S2_TW;
    };
    // space-to-depth entry e of the patch -> (row, column) in the patch; columns are stored even plane | odd plane
    auto entry_rc = [&](int e, int& sr, int& sc) {
        sr = e / S2_SPITCH;
        const int rem = e - sr * S2_SPITCH, pl = rem >= S2_SHALF;
        sc = 2 * (rem - pl * S2_SHALF) + pl;
    };
    int sr0, sc0, sr1, sc1;                                                // each thread stages entries tid and tid + 512
    entry_rc(tid, sr0, sc0);
    entry_rc(min(tid + S2_THREADS, S2_NS - 1), sr1, sc1);
    if (sc0 >= S2_SC) sr0 = -0x10000;                                      // padding entries: outside any image -> zero
    if (sc1 >= S2_SC) sr1 = -0x10000;
    auto fetch = [&](const TileWalk& t, RawEntry<U8>& v0, RawEntry<U8>& v1) {
        int stream, b, y0, x0;
        decode(t, stream, b, y0, x0);
        load_entry<U8>(q, stream, b, 2 * y0 - 2 + sr0, 2 * x0 - 2 + sc0, v0);
        load_entry<U8>(q, stream, b, 2 * y0 - 2 + sr1, 2 * x0 - 2 + sc1, v1);
    };
    auto commit = [&](const RawEntry<U8>& v0, const RawEntry<U8>& v1) {
        store_entry<DT, U8>(s2d, tid, v0);
        if (tid + S2_THREADS < S2_NS) store_entry<DT, U8>(s2d, tid + S2_THREADS, v1);
    };
    // PAIR: thread -> (patch row psr, column pair pk): entries (psr, 2 pk) of the even plane and (psr, 2 pk + 1) of the odd plane
    constexpr int S2_NPAIR = S2_SR * S2_SPAIRS;                            // 374
    const int psr = tid / S2_SPAIRS, pk = tid - psr * S2_SPAIRS;
    float4 pr[PAIR ? 6 : 1];
    bool pin0 = false, pin1 = false;
    auto fetch_pair = [&](const TileWalk& t) {
        if (tid >= S2_NPAIR) return;
        int stream, b, y0, x0;
        decode(t, stream, b, y0, x0);
        const int hh = q.H >> 1, hw = q.W >> 1;                               // (hw even: the host selects PAIR for W % 4 == 0 only)
        const int gy = 2 * y0 - 2 + psr, gx = 2 * x0 - 2 + 2 * pk;            // gx even: the four pixels never straddle a 16-byte boundary
        const bool iny = (unsigned)gy < (unsigned)hh;
        pin0 = iny && (unsigned)gx < (unsigned)hw && 2 * pk < S2_SC;
        pin1 = iny && (unsigned)(gx + 1) < (unsigned)hw && 2 * pk + 1 < S2_SC;
        const int cy = min(max(gy, 0), hh - 1), cx = min(max(gx, 0), hw - 2);
        const long long plane = (long long)q.H * q.W;
        const unsigned off = (unsigned)(2 * cy) * (unsigned)q.W + (unsigned)(2 * cx);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* base = (const float*)q.img + ((long long)(stream * q.B + b) * 3 + c) * plane;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) pr[PAIR ? c * 2 + dy : 0] = *(const float4*)(base + off + (unsigned)(dy * q.W));
        }
    };
    auto commit_pair = [&]() {
        if (tid >= S2_NPAIR) return;
        if constexpr (!U8) {
            RawEntry<false> e0, e1;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4 v = pr[PAIR ? i : 0];
                e0.r[i] = float2{v.x, v.y};
                e1.r[i] = float2{v.z, v.w};
            }
            e0.in = pin0; e1.in = pin1;
            store_entry<DT, false>(s2d, psr * S2_SPITCH + pk, e0);
            store_entry<DT, false>(s2d, psr * S2_SPITCH + S2_SHALF + pk, e1);
        }
    };

    // weight matrix -> LDS: `nsl` slices of `rows` rows x 128 bytes in igemm's swizzle (8 rows per DMA instruction)
    auto dma_rows = [&](const void* base, int rows, int kp, int nsl, unsigned char* dst) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(rows * kp * 2), 0x00020000);
        const int rs8 = lane >> 3, per = rows >> 3;
        for (int t = wave; t < nsl * per; t += S2_THREADS / 64) {
            const int c = t / per, j = t - c * per;
            const int sl = (lane & 7) ^ (((j & 1) << 2) | (rs8 >> 1));
            const unsigned voff = ((unsigned)(j * 8 + rs8) * (unsigned)kp + (unsigned)(c * 64 + sl * 8)) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(dst + (c * rows + j * 8) * RB), 16, voff, 0, 0, 0);
        }
    };
    auto load_weights = [&](int stream) {
        dma_rows((const typename E::type*)q.w0 + stream * q.w0_gs, C0, q.Kp0, 3, w0b);
        dma_rows((const typename E::type*)p.w + stream * p.w_gs, C1, p.Kp, S2_W1_SLICES, w1b);
        dma_rows((const typename E::type*)p.w2 + stream * p.w2_gs, C2, p.Kp2, 1, w2b);
        if (tid < C1) bl1[tid] = p.bias[stream * p.bias_gs + tid];
        else if (tid < C1 + C2) bl2[tid - C1] = tid - C1 < p.Cout2 ? p.bias2[stream * p.bias2_gs + tid - C1] : 0.0f;
        wait_vmcnt<0>();
    };

    const int xcd = blockIdx.x & 7, per_xcd = (q.npatch + 7) >> 3, pstride = gridDim.x >> 3;
    const int pend = min((xcd + 1) * per_xcd, q.npatch);
    int pt = xcd * per_xcd + (blockIdx.x >> 3);
    if (pt >= pend) return;
    TileWalk cur_t, nxt_t;
    cur_t.init(pt, pstride, q.tiles_x, q.tiles_y);
    RawEntry<U8> v0, v1;
    if constexpr (PAIR) fetch_pair(cur_t); else fetch(cur_t, v0, v1);

    const int fkey = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);
    // ---- tile-invariant fragment addresses (see the layout note above) ---------------------------------------------------------
    // stage 1: this wave's up to three sub-tile jobs (19 jobs over 8 waves), dealt from wave 0 up: waves w and w + 4 share a SIMD and VALU issue is
    // arbitrated by age, so a third job costs least on the OLDER wave of a pair (round 6, same box: 319 -> 314 us; dealing from the last wave
    // down — the waves that stage no image data — left the younger wave of three SIMDs alone with its third job).  Results do not depend on the dealing.
#ifdef ICAF_S2_FAKEJOBS            // timing probe only (results are garbage): pretend stage 1 has this many jobs — 16 = two per wave, 24 = three per wave
    constexpr int S2_NJOBS = ICAF_S2_FAKEJOBS, S2_JPW = (S2_NJOBS + S2_THREADS / 64 - 1) / (S2_THREADS / 64);
#else
    constexpr int S2_NJOBS = (S2_NL + 31) / 32, S2_JPW = (S2_NJOBS + S2_THREADS / 64 - 1) / (S2_THREADS / 64);
#endif
    const int job0 = wave;
    int s1_rd[S2_JPW][3], s1_wr[S2_JPW], s1_key[S2_JPW], s1_hy[S2_JPW], s1_hx[S2_JPW];
#pragma unroll
    for (int jj = 0; jj < S2_JPW; ++jj) {
        const int lidx = ((job0 + jj * (S2_THREADS / 64)) << 5) + l31;         // logical entry: (row, plane, column index)
        const int lc = lidx < S2_NL ? lidx : 0;
        const int hy = lc / S2_LPITCH, rem = lc - hy * S2_LPITCH, pl = rem >= S2_LHALF, i = rem - pl * S2_LHALF;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int sc = 2 * i + pl + kx;
            const int e = hy * S2_SPITCH + (sc & 1) * S2_SHALF + (sc >> 1);     // space-to-depth entry of tap (0, kx); tap row ky: + ky * S2_SPITCH
            s1_rd[jj][kx] = ((e << 1) + (hi ^ ((e >> 3) & 1))) << 4;
        }
        const int idx = hy * S2_PITCH + pl * S2_HALF + i;                      // physical entry of the stem pixel in the halo patch
        // (the logical enumeration has S2_LHALF = 33 odd slots per row but only 32 odd columns exist: the phantom slot (hx = 65) would land on the next
        //  row's even column 0 — the same address another lane of the same ds_write targets — and, in the last row, one entry past the patch: never stored)
        s1_wr[jj] = (lidx < S2_NL && 2 * i + pl < S2_HWD) ? (idx << 6) + (hi << 3) : -1;      // byte address of channel quad 0's slot 0 (+ 8 bytes for the upper lane half)
        s1_key[jj] = (idx >> 2) & 3;
        s1_hy[jj] = hy;
        s1_hx[jj] = 2 * i + pl;
    }
    // stage 2: the lane's output pixel (row wm, column l31): entry of tap (0, 0) and of tap (0, 2) (one entry further), each with the two
    // 16-byte channel slots a K step of 16 reads; taps (ky, kx & 1) are immediates
    int s2_rd[2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int idx = wm * 2 * S2_PITCH + l31 + d;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) s2_rd[d][h2] = ((idx << 2) + (((h2 * 2 + hi) ^ (idx >> 2)) & 3)) << 4;
    }

    // Per-stream constants held in registers: the stem's weight fragments and the three bias vectors of this lane's
    // channels.  (Loading a bias inside the tile loop is a dependent L2 round trip per use — with one workgroup per CU
    // nothing else would cover it.)
    u32x4 fw0[9];
    f32x4 b0r[4];
    int wstream = -1;
    if constexpr (PAIR) commit_pair(); else commit(v0, v1);
#ifdef ICAF_S2_CLK
    int clk_tile = -1;
#endif
    while (true) {
        int stream, b, y0, x0;
        decode(cur_t, stream, b, y0, x0);
#ifdef ICAF_S2_CLK
        ++clk_tile;
#endif
        S2_STAMP(0);
        if (stream != wstream) {                   // (re)load this stream's weights (at most twice in a workgroup's life)
            lds_barrier();
            load_weights(stream);
            lds_barrier();
            wstream = stream;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) fw0[tap] = *(const u32x4*)(w0b + (tap >> 2) * C0 * RB + foff[tap & 3]);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) b0r[qd] = *(const f32x4*)(q.bias0 + stream * q.bias0_gs + 8 * qd + 4 * hi);
            // consume them here: otherwise the compiler's wait-count bookkeeping treats them as possibly pending at every
            // use inside the tile loop and drains the NEXT tile's prefetch with them
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) asm volatile("" : "+v"(b0r[qd]));
        }
        const int pn = pt + pstride;
        const bool more = pn < pend;
        nxt_t = cur_t;
        nxt_t.next();
        if (more) { if constexpr (PAIR) fetch_pair(nxt_t); else fetch(nxt_t, v0, v1); }     // next tile's image reads stay in flight during everything below
        S2_STAMP(1);
        lds_barrier();                             // space-to-depth patch visible
        S2_STAMP(2);

        // ---- stage 1: stem over the halo patch ------------------------------------------------------------------
        {
            const int sy0 = 2 * y0 - 1, sx0 = 2 * x0 - 1;
            // The nine fragment reads of a job are issued as one burst and pinned there (sched_barrier): left alone the scheduler sinks every
            // read to just above its MFMA and waits lgkmcnt(0) per step — a full LDS round trip (>= 64 cycles) per 32-cycle MFMA, two waves per
            // SIMD cannot hide that.  The NEXT job's burst goes out right behind this job's MFMAs, i.e. it lands under the SiLU epilogue.
            u32x4 fp[2][9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) fp[0][tap] = *(const u32x4*)(s2d + s1_rd[0][tap % 3] + (tap / 3) * (S2_SPITCH * 32));
#ifdef ICAF_S2_PRIO
            if (job0 + (S2_JPW - 1) * (S2_THREADS / 64) < S2_NJOBS) __builtin_amdgcn_s_setprio(ICAF_S2_PRIO);      // the waves with a third job
#endif
#pragma unroll
            for (int jj = 0; jj < S2_JPW; ++jj) {
                if (job0 + jj * (S2_THREADS / 64) >= S2_NJOBS) break;             // (wave-uniform)
                f32x16 a0;
#pragma unroll
                for (int r = 0; r < 16; ++r) a0[r] = 0.0f;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) mma_step<DT>(a0, fw0[tap], fp[jj & 1][tap]);
                if (jj + 1 < S2_JPW && job0 + (jj + 1) * (S2_THREADS / 64) < S2_NJOBS) {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap)
                        fp[(jj + 1) & 1][tap] = *(const u32x4*)(s2d + s1_rd[jj + 1 < S2_JPW ? jj + 1 : 0][tap % 3] + (tap / 3) * (S2_SPITCH * 32));
                }
                __builtin_amdgcn_sched_barrier(0);
                const bool inside = s1_hx[jj] < S2_HWD && (unsigned)(sy0 + s1_hy[jj]) < (unsigned)q.Hs && (unsigned)(sx0 + s1_hx[jj]) < (unsigned)q.Ws;
                if (s1_wr[jj] >= 0) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        float v[4] = {0.f, 0.f, 0.f, 0.f};
                        if (inside) {
                            const float x[4] = {a0[4 * qd] + b0r[qd][0], a0[4 * qd + 1] + b0r[qd][1], a0[4 * qd + 2] + b0r[qd][2], a0[4 * qd + 3] + b0r[qd][3]};
                            silu4_f(x, v);
                        }
                        u32x2 pk;
                        if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                        else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                        *(u32x2*)(halo + s1_wr[jj] + ((qd ^ s1_key[jj]) << 4)) = pk;      // channel quad qd lives in slot qd ^ key of the entry
                    }
                }
            }
        }
        S2_STAMP(3);
#ifdef ICAF_S2_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        lds_barrier();                             // halo patch visible; the space-to-depth patch is free
        S2_STAMP(4);

        // ---- stage 2: 3x3 / stride 2 over the halo patch (ctile.hip's loop, weights resident) ---------------------
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
        f32x4 b1r[4];
        {                                          // K = (tap, channel): 16 per MFMA step; fragment reads S2_PFD steps ahead of their MFMA, pinned (see stage 1)
            constexpr int NK = 9 * C0 / 16, S2_PFD = 4;
            u32x4 fpq[NK], fwq[NK];
            auto rd2 = [&](int k) {
                const int tap = k >> 1, ky = tap / 3, kx = tap - 3 * ky;
                fpq[k] = *(const u32x4*)(halo + s2_rd[kx >> 1][k & 1] + (ky * S2_PITCH + (kx & 1) * S2_HALF) * 64);
                fwq[k] = *(const u32x4*)(w1b + (k >> 2) * C1 * RB + (wn * 32) * RB + foff[k & 3]);
            };
#pragma unroll
            for (int k = 0; k < S2_PFD; ++k) rd2(k);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                if (k + S2_PFD < NK) rd2(k + S2_PFD);
                if (k + S2_PFD == NK) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) b1r[qd] = *(const f32x4*)(bl1 + wn * 32 + 8 * qd + 4 * hi);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma_step<DT>(acc[0][0], fwq[k], fpq[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        S2_STAMP(5);
        {                                          // t1 tile -> LDS, rounded to the storage type (igemm CHAIN, step a)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int nl = wn * 32 + 8 * qd + 4 * hi;
                float v[4];
                const float x[4] = {acc[0][0][4 * qd] + b1r[qd][0], acc[0][0][4 * qd + 1] + b1r[qd][1], acc[0][0][4 * qd + 2] + b1r[qd][2], acc[0][0][4 * qd + 3] + b1r[qd][3]};
                silu4_f(x, v);
                u32x2 pk;
                if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                *(u32x2*)(s2d + (wm * 32 + l31) * SO + nl * E::BYTES) = pk;
            }
        }
        S2_STAMP(6);
        lds_barrier();
        S2_STAMP(7);

        // ---- stage 3: chained 1x1 on the staged tile, then the epilogue (conv_common.h's, with the bias in registers
        //      and LDS-only barriers so that the prefetch stays in flight) -----------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
        f32x4 b2r[4];
        {
            u32x4 fp2[C1 / 16], fw2[C1 / 16];
#pragma unroll
            for (int s2 = 0; s2 < C1 / 16; ++s2) {
                fp2[s2] = *(const u32x4*)(s2d + (wm * 32 + l31) * SO + ((2 * s2 + hi) << 4));
                fw2[s2] = *(const u32x4*)(w2b + (wn * 32) * RB + foff[s2]);
            }
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) b2r[qd] = *(const f32x4*)(bl2 + wn * 32 + 8 * qd + 4 * hi);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < C1 / 16; ++s2) mma_step<DT>(acc[0][0], fw2[s2], fp2[s2]);
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {           // (every wave finished stage 2 before the barrier above: halo is free)
            const int nl = wn * 32 + 8 * qd + 4 * hi;
            float v[4];
            const float x[4] = {acc[0][0][4 * qd] + b2r[qd][0], acc[0][0][4 * qd + 1] + b2r[qd][1], acc[0][0][4 * qd + 2] + b2r[qd][2], acc[0][0][4 * qd + 3] + b2r[qd][3]};
            silu4_f(x, v);
            u32x2 pk;
            if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
            else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
            *(u32x2*)(halo + (wm * 32 + l31) * SO + nl * E::BYTES) = pk;
        }
        S2_STAMP(8);
        lds_barrier();
        S2_STAMP(9);
        {
            // staged vectors -> registers, then the next tile's patch is committed BEFORE the global stores are issued:
            // the wait for the prefetch (vmcnt counts in order) then never includes this tile's stores
            typename E::type* __restrict__ yg = (typename E::type*)p.y2 + stream * p.y2_gs;
            constexpr int NIT = S2_TH * S2_TW * (C2 / 8) / S2_THREADS;
            u32x4 sv[NIT];
            long long yoff[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * S2_THREADS, row = idx >> 3, cv = idx & 7;
                const int gy = y0 + (row >> 5), gx = x0 + (row & 31);
                sv[it] = *(const u32x4*)(halo + row * SO + cv * 16);
                yoff[it] = (gy < p.Ho && gx < p.Wo && cv * 8 < p.Cout2) ? (long long)((b * p.Ho + gy) * p.Wo + gx) * p.ldy2 + cv * 8 : -1;
            }
            if (more) { if constexpr (PAIR) commit_pair(); else commit(v0, v1); }     // (every wave passed the barrier above: the staged t1 tile is consumed)
            S2_STAMP(10);
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if (yoff[it] >= 0) *(u32x4*)(yg + yoff[it]) = sv[it];
            S2_STAMP(11);
        }
        if (!more) break;
        pt = pn;
        cur_t = nxt_t;
    }
}


}  // namespace icaf

using namespace icaf;

template <int DT, int BN, bool U8>
static int launch_stem(const StemP& q, hipStream_t s) {
    constexpr int EB = Elem<DT>::BYTES;
    const int lds = 2 * ST_PATCH_BYTES + 3 * BN * 128 + ST_TH * ST_TW * (BN * EB + 16);
    int dev = 0, cus = 256;
    ICAF_HIP(hipGetDevice(&dev));
    ICAF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int per_cu = (160 * 1024) / lds < 4 ? (160 * 1024) / lds : 4;
    int grid = cus * (per_cu < 1 ? 1 : per_cu);
    if (grid > q.npatch) grid = q.npatch;
    grid = (grid + 7) & ~7;                         // the patch walk is per XCD (8 of them)
    ICAF_LDS_OPTIN((stem_kernel<DT, BN, U8>), lds);
    stem_kernel<DT, BN, U8><<<dim3((unsigned)grid), dim3(NTHREADS), lds, s>>>(q);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

extern "C" int icaf_stem(const void* img, int img_u8, int ctot, const void* w, const float* bias, void* y, int ldy, int dtype, int nstreams,
                         int B, int H, int W, int Cout, int Kp, long long w_gs, long long bias_gs, long long y_gs, icaf_stream_t s) {
    if (!img || !w || !y) return fail(ICAF_ERR_ARG, "icaf_stem: null pointer");
    if (dtype != ICAF_BF16 && dtype != ICAF_F16) return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem: 16-bit types only");
    if ((H | W) & 1) return fail(ICAF_ERR_ARG, "icaf_stem: H and W must be even");
    if ((long long)H * W >= (1LL << 29)) return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem: image plane of 2^29 or more pixels");
    if (Cout != 32 && Cout != 64) return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem: built for 32 or 64 output channels (got %d)", Cout);
    if (Kp != 192) return fail(ICAF_ERR_ARG, "icaf_stem: the packed space-to-depth weights must have Kp = 192 (K = 9 x 16), got %d", Kp);
    if (nstreams < 1 || nstreams > 2 || (img_u8 && ctot < 3 * nstreams)) return fail(ICAF_ERR_ARG, "icaf_stem: bad nstreams / ctot");
    if (ldy % 8 || ((uintptr_t)y & 15) || (y_gs * 2) % 16) return fail(ICAF_ERR_ARG, "icaf_stem: y must be 16-byte aligned with ldy %% 8 == 0");
    StemP q;
    memset(&q, 0, sizeof(q));
    q.img = img; q.ctot = ctot; q.B = B; q.H = H; q.W = W; q.nstreams = nstreams;
    ConvP& p = q.c;
    p.w = w; p.bias = bias; p.y = y; p.res = nullptr;
    p.w_gs = w_gs; p.bias_gs = bias_gs; p.y_gs = y_gs;
    p.B = B; p.H = H / 2; p.W = W / 2; p.Ho = H / 2; p.Wo = W / 2; p.Cin = 16; p.Cout = Cout; p.ldy = ldy; p.Kp = Kp;
    p.M = B * p.Ho * p.Wo; p.K = 144; p.act = ICAF_ACT_SILU;
    p.vec_y = 1; p.vec_r = 0;
    p.w_bytes = (unsigned)(128LL * Kp * 2);
    p.alpha_acc[0] = p.alpha_acc[1] = 1.0f; p.alpha_res[0] = p.alpha_res[1] = 0.0f;
    q.tiles_x = (p.Wo + ST_TW - 1) / ST_TW; q.tiles_y = (p.Ho + ST_TH - 1) / ST_TH;
    q.npatch = nstreams * B * q.tiles_x * q.tiles_y;
    hipStream_t hs = S(s);
    if (dtype == ICAF_BF16) {
        if (Cout == 32) return img_u8 ? launch_stem<ICAF_BF16, 32, true>(q, hs) : launch_stem<ICAF_BF16, 32, false>(q, hs);
        return img_u8 ? launch_stem<ICAF_BF16, 64, true>(q, hs) : launch_stem<ICAF_BF16, 64, false>(q, hs);
    }
    if (Cout == 32) return img_u8 ? launch_stem<ICAF_F16, 32, true>(q, hs) : launch_stem<ICAF_F16, 32, false>(q, hs);
    return img_u8 ? launch_stem<ICAF_F16, 64, true>(q, hs) : launch_stem<ICAF_F16, 64, false>(q, hs);
}

template <int DT, bool U8, bool PAIR>
static int launch_stem2(const Stem2P& q, hipStream_t s) {
    int dev = 0, cus = 256;
    ICAF_HIP(hipGetDevice(&dev));
    ICAF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int grid = cus < q.npatch ? cus : q.npatch;     // 122 KiB of LDS: one workgroup per CU
    grid = (grid + 7) & ~7;                         // the tile walk is per XCD (8 of them)
    ICAF_LDS_OPTIN((stem2_kernel<DT, U8, PAIR>), S2_LDS);
    stem2_kernel<DT, U8, PAIR><<<dim3((unsigned)grid), dim3(S2_THREADS), S2_LDS, s>>>(q);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

extern "C" int icaf_stem2(const icaf_stem2_args* a, icaf_stream_t s) {
    if (!a || !a->img || !a->w0 || !a->w1 || !a->w2 || !a->bias0 || !a->bias1 || !a->bias2 || !a->y)
        return fail(ICAF_ERR_ARG, "icaf_stem2: null pointer");
    if (a->dtype != ICAF_BF16 && a->dtype != ICAF_F16) return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem2: 16-bit types only");
    if ((a->H | a->W) & 1 || a->H < 4 || a->W < 4) return fail(ICAF_ERR_ARG, "icaf_stem2: H and W must be even and >= 4");
    if ((long long)a->H * a->W >= (1LL << 29)) return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem2: image plane of 2^29 or more pixels");
    if (a->C0 != S2_C0 || a->C1 != S2_C1 || a->C2 < 8 || a->C2 > S2_C2 || a->C2 % 8)
        return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem2: built for 32 -> 64 -> (<= 64) channels, got %d -> %d -> %d", a->C0, a->C1, a->C2);
    if (a->Kp0 != 192 || a->Kp1 != 64 * S2_W1_SLICES || a->Kp2 != 64)
        return fail(ICAF_ERR_ARG, "icaf_stem2: packed weights must have Kp = 192 / %d / 64, got %d / %d / %d", 64 * S2_W1_SLICES, a->Kp0, a->Kp1, a->Kp2);
    if (a->nstreams < 1 || a->nstreams > 2 || (a->img_u8 && a->ctot < 3 * a->nstreams)) return fail(ICAF_ERR_ARG, "icaf_stem2: bad nstreams / ctot");
    if (a->ldy % 8 || ((uintptr_t)a->y & 15) || (a->y_gs * 2) % 16) return fail(ICAF_ERR_ARG, "icaf_stem2: y must be 16-byte aligned with ldy %% 8 == 0");
    Stem2P q;
    memset(&q, 0, sizeof(q));
    q.img = a->img; q.ctot = a->ctot; q.B = a->B; q.H = a->H; q.W = a->W; q.nstreams = a->nstreams;
    q.Hs = a->H / 2; q.Ws = a->W / 2;
#ifdef ICAF_S2_CLK
    q.clk = (unsigned long long*)strtoull(getenv("ICAF_S2_CLK_PTR") ? getenv("ICAF_S2_CLK_PTR") : "0", nullptr, 0);
    if (!q.clk) return fail(ICAF_ERR_ARG, "icaf_stem2 probe build: ICAF_S2_CLK_PTR is not set");
#endif
    q.w0 = a->w0; q.bias0 = a->bias0; q.w0_gs = a->w0_gs; q.bias0_gs = a->bias0_gs; q.Kp0 = a->Kp0;
    ConvP& p = q.c;
    p.w = a->w1; p.bias = a->bias1; p.w_gs = a->w1_gs; p.bias_gs = a->bias1_gs; p.Kp = a->Kp1;
    p.w2 = a->w2; p.bias2 = a->bias2; p.w2_gs = a->w2_gs; p.bias2_gs = a->bias2_gs; p.Kp2 = a->Kp2;
    p.y2 = a->y; p.y2_gs = a->y_gs; p.ldy2 = a->ldy; p.Cout2 = a->C2; p.vec_y2 = 1;
    p.B = a->B; p.H = q.Hs; p.W = q.Ws; p.Cin = S2_C0; p.Cout = S2_C1;
    p.Ho = (q.Hs - 1) / 2 + 1; p.Wo = (q.Ws - 1) / 2 + 1;                  // 3x3 / stride 2 / pad 1
    p.M = a->B * p.Ho * p.Wo; p.K = 9 * S2_C0; p.act = ICAF_ACT_SILU;
    p.alpha_acc[0] = p.alpha_acc[1] = 1.0f;
    if ((long long)a->nstreams * p.M * a->ldy >= (1LL << 31)) return fail(ICAF_ERR_UNSUPPORTED, "icaf_stem2: output exceeds 2^31 elements");
    q.tiles_x = (p.Wo + S2_TW - 1) / S2_TW; q.tiles_y = (p.Ho + S2_TH - 1) / S2_TH;
    q.npatch = a->nstreams * a->B * q.tiles_x * q.tiles_y;
    hipStream_t hs = S(s);
    const bool pair = !a->img_u8 && a->W % 4 == 0 && ((uintptr_t)a->img & 15) == 0;          // 16-byte image loads (stem2_kernel PAIR)
    if (a->dtype == ICAF_BF16) {
        if (a->img_u8) return launch_stem2<ICAF_BF16, true, false>(q, hs);
        return pair ? launch_stem2<ICAF_BF16, false, true>(q, hs) : launch_stem2<ICAF_BF16, false, false>(q, hs);
    }
    if (a->img_u8) return launch_stem2<ICAF_F16, true, false>(q, hs);
    return pair ? launch_stem2<ICAF_F16, false, true>(q, hs) : launch_stem2<ICAF_F16, false, false>(q, hs);
}
