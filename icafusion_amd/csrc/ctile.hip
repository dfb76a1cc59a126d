// 3x3 direct convolution from an LDS halo tile for gfx950 (MI355X) — the high-resolution 3x3 layers of the backbones.
//
// The implicit-GEMM kernel (igemm.hip) re-fetches every input pixel once per filter tap: for a 3x3 layer the L2 -> LDS
// feed carries 9x the tensor, and with few output channels (N <= 64) that feed — not HBM, not the MFMA pipe — is what
// bounds the layer (stem: 1.05 GB of im2col traffic per stream at ~6 TB/s).  Here a workgroup owns a TH x TW patch of
// output pixels of one image:
//   1. the (TH-1)*S+3 x (TW-1)*S+3 input halo patch is fetched ONCE with `buffer_load_dwordx4 ... lds` (out-of-image
//      pixels are zero-filled by the descriptor range check), laid out pixel-major with the 16-byte channel slots of a
//      pixel XOR-swizzled by the pixel index so that the 16 lanes of a ds_read_b128 group (consecutive pixels of one
//      row) hit 16 distinct bank groups; for stride 2 even and odd columns are stored as separate planes so that a
//      tap still reads consecutive entries;
//   2. the K loop walks (ky, kx, cin) in the SAME order and MFMA step size as igemm — results are bit-identical to
//      igemm's — with the weight slices streamed through a 3-stage LDS-DMA ring (as in igemm) and the pixel operand
//      of each MFMA step read straight from the halo patch at (lane base + tap offset);
//   3. the epilogue is igemm's (bias + SiLU in registers, LDS-staged 16-byte stores, residual).
// Requirements (checked by the host; anything else stays on igemm): 3x3, pad 1, stride 1 or 2, Cin a power of two with
// 32..256 bytes per pixel, Cout <= BN, SiLU, out dtype == dtype.
#include "conv_common.h"

// Ablation switches for timing studies (tools/quick_variant.py -DICAF_CTILE_ABL=n; results are then meaningless):
//   1 = no halo-patch DMA, 2 = no weight-ring DMA, 4 = no LDS fragment reads in the 3x3 loop, 8 = no MFMAs in the 3x3 loop
#ifndef ICAF_CTILE_ABL
#define ICAF_CTILE_ABL 0
#endif
// 1 = the fused Bottleneck + cv3 asks for its residual / cv2 vectors behind the 3x3 loop (the form before round 4; A/B builds only)
#ifndef ICAF_CTILE_LATE3
#define ICAF_CTILE_LATE3 0
#endif

namespace icaf {

template <int S, int TW> struct HaloGeom {
    static constexpr int HWD = (TW - 1) * S + 3;                      // halo width in pixels
    static constexpr int HALF = S == 2 ? (HWD + 1) / 2 : 0;           // entries of the even-column plane (stride 2)
    // row pitch in entries.  TW = 16: an MFMA sub-tile spans two patch rows, whose entry indices must differ by a
    // multiple of 16 to keep the b128 lane groups conflict-free.
    static constexpr int PITCH = TW == 16 ? ((S == 2 ? 2 * HALF : HWD) + 15) / 16 * 16 : (S == 2 ? 2 * HALF : HWD);
};

// FUSE1 = true turns the kernel into a whole Bottleneck (reference models/common.py:184-194):
//   y = x + SiLU(conv3x3(SiLU(conv1x1(x))))        (c_ -> c_ -> c_ channels, c_ = BN = 32 or 64)
// The 1x1 convolution is evaluated on the halo patch itself (one extra MFMA stage, its c_ x c_ weights are a single
// 128-byte slice), its SiLU output is written — zeroed outside the image, which is the 3x3's zero padding — as a second
// LDS patch in the same swizzled layout, and the unchanged 3x3 loop reads that patch.  The intermediate tensor of the
// two-launch form (written and re-read at full resolution) never exists; the residual is the epilogue's `res` = x.
// CHAIN3 = true (with FUSE1, c_ = BN = 32) appends the C3's cv3 to the Bottleneck (reference models/common.py:226:
//   cv3(cat(m(cv1(x)), cv2(x)))): the Bottleneck's output tile — residual added and rounded to the storage type exactly
// as it would be written — is placed beside the cv2 half of the same pixels (fetched from HBM) in an LDS tile, which is
// the pixel operand of the 2c_ -> Cout2 1x1 (weights resident in LDS); only cv3's output is written (p.w2 / bias2 / y2,
// p.x2 = the cv2 half).  Channel order of the tile = [cv2 | m] (the order the three-slot C3 buffer presents after one
// fused Bottleneck; the weights are packed to match), K order and MFMA step as igemm: bit-identical to the two launches.
constexpr int ct_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <int DT, int TH, int TW, int BN, int ACT, int S, bool FUSE1, bool CHAIN3 = false>
__device__ __forceinline__ void ctile_body(const ConvP& p, const int lsp_arg, const int lcin_arg, const int tiles_x,
                                           const int tiles_per_img, const int halo_bytes) {
    using E = Elem<DT>;
    // The fused Bottleneck is c_ -> c_ -> c_ with c_ = BN: channel count, slots per pixel and the K loop's trip count are compile-time
    // constants there, so the tap / slot arithmetic of every MFMA step folds away and the loop unrolls (SQ counters had this kernel at
    // 0.74 VALU issue against 0.14 MFMA busy: 13 vector instructions of address arithmetic per two MFMAs).  The plain 3x3 keeps the run-time values.
    const int lsp = FUSE1 ? ct_log2(BN * E::BYTES / 16) : lsp_arg, lcin = FUSE1 ? ct_log2(BN) : lcin_arg;
    using G = HaloGeom<S, TW>;
    constexpr int VEC = E::VEC;
    constexpr int RB = 128, NS = 3;
    constexpr int BK = RB / E::BYTES;              // K elements per weight slice
    constexpr int KSTEP = 2 * VEC;                 // K elements per MFMA step
    constexpr int NSTEP = BK / KSTEP;              // 4
    constexpr int BM = TH * TW, WM = BM / 4, TM = WM / 32, TN = BN / 32;
    constexpr int HH = (TH - 1) * S + 3;
    constexpr int PITCH = G::PITCH, HALF = G::HALF, HWD = G::HWD;
    constexpr int NBW = BN / 32;                   // weight DMA instructions per wave per slice (8 rows each)
    constexpr int WSTAGE = BN * RB;
    static_assert(TM >= 1 && (TW == 32 || TW == 16) && BM % 128 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* xpatch = lds;                                   // input halo patch
    // FUSE1: the 1x1's output patch OVERWRITES the input patch entry by entry (a 32-entry sub-tile is read completely — every K step
    // of its MFMAs — before its owner wave writes it back, and no other wave touches those entries in that stage): one patch
    // instead of two takes the Bottleneck from 70 to 47 KB of LDS, i.e. from two to three workgroups per CU
    unsigned char* halo = lds;                                     // the patch the 3x3 loop reads
    unsigned char* w1buf = lds + halo_bytes;                       // FUSE1: the 1x1 weights, BN rows x 128 bytes
    unsigned char* ring = FUSE1 ? w1buf + BN * RB : lds + halo_bytes;
    unsigned char* w3buf = ring + NS * WSTAGE;                     // CHAIN3: cv3's weights, 2*BN rows x 128 bytes

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const int tile = xcd_tile(p.mtiles);
    const int b = tile / tiles_per_img, tr = tile - b * tiles_per_img;
    const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;          // output-pixel origin of the patch
    const int SP = 1 << lsp;                       // 16-byte slots per pixel

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.w + g * p.w_gs), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    auto row_to_m = [&](int row) {
        const int st = row >> 5, r = row & 31;
        const int py = TW == 32 ? st : st * 2 + (r >> 4), px = TW == 32 ? r : (r & 15);
        const int gy = y0 + py, gx = x0 + px;
        return (gy < p.Ho && gx < p.Wo) ? (b * p.Ho + gy) * p.Wo + gx : -1;
    };
    // ---- 1. halo patch -> LDS (each DMA instruction fills 64 consecutive 16-byte slots) --------------------------
    {
        const int nslots = (HH * PITCH) << lsp;
        const int ninstr = (nslots + 63) >> 6;
        const unsigned img_off = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES;
        const int gy0 = y0 * S - 1, gx0 = x0 * S - 1;
        for (int j = wave; j < ninstr; j += 4) {
            const int L = (j << 6) + lane;
            const int idx = L >> lsp;
            const int cs = (L & (SP - 1)) ^ ((idx >> (4 - lsp)) & (SP - 1));
            const int hy = idx / PITCH, rem = idx - hy * PITCH;
            int hx;
            if constexpr (S == 2) { const int pl = rem >= HALF; hx = 2 * (rem - pl * HALF) + pl; }
            else hx = rem;
            const int gy = gy0 + hy, gx = gx0 + hx;
            const bool ok = hy < HH && hx < HWD && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned voff = ok ? img_off + (unsigned)((gy * p.W + gx) * p.ldx) * E::BYTES + (unsigned)(cs << 4) : OOB;
            if constexpr (!(ICAF_CTILE_ABL & 1))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(xpatch + (j << 10)), 16, voff, 0, 0, 0);
        }
    }

    // CHAIN3: the residual (the Bottleneck's input) and the cv2 half of cv3's operand at this thread's tile positions are requested NOW, beside
    // the patch — a pixel's [cv1 | cv2] channels are ONE 128-byte line of the C3's buffer, so the lines the patch DMA brings in serve these
    // loads from the L2 — and held in registers through the 3x3 loop.  (Requested behind the loop, as before round 4, they were a dependent
    // HBM round trip at the end of every workgroup's life and re-fetched lines the L2 had dropped: PMC 387 MB per launch for 210 MB read.)
    constexpr int VPR3 = BN / VEC, NIT3 = CHAIN3 ? BM * VPR3 / NTHREADS : 1;
    u32x4 rv[NIT3], cv2v[NIT3];
    if constexpr (CHAIN3 && !ICAF_CTILE_LATE3) {
        const typename E::type* __restrict__ rg = p.res ? (const typename E::type*)p.res + g * p.res_gs : nullptr;
        const typename E::type* __restrict__ x2g = (const typename E::type*)p.x2 + g * p.x2_gs;
#pragma unroll
        for (int it = 0; it < NIT3; ++it) {
            const int idx = tid + it * NTHREADS, row = idx / VPR3, cv = idx - row * VPR3;
            const int m = row_to_m(row);
            const long long mm = m < 0 ? 0 : m;
            rv[it] = rg ? *(const u32x4*)(rg + mm * p.ldr + cv * VEC) : u32x4{0u, 0u, 0u, 0u};
            cv2v[it] = *(const u32x4*)(x2g + mm * p.ldx2 + cv * VEC);
        }
    }

    // ---- 2. weight ring (identical addressing to igemm's RB = 128 pipeline) ---------------------------------------
    const int rsub = lane >> 3;
    const int dkey = ((wave & 1) << 2) | (rsub >> 1);
    const int lslot = (lane & 7) ^ dkey;
    const unsigned w_off0 = ((unsigned)(wave * 8 + rsub) * (unsigned)p.Kp + (unsigned)(lslot * VEC)) * E::BYTES;
    auto issue_w = [&](int chunk, int stage) {
        unsigned char* st = ring + stage * WSTAGE;
        const bool cvalid = chunk < p.nchunks;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const unsigned voff = cvalid ? w_off0 + (unsigned)chunk * RB + (unsigned)(32 * i) * (unsigned)p.Kp * E::BYTES : OOB;
            if constexpr (!(ICAF_CTILE_ABL & 2))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(st + (wave + 4 * i) * 1024), 16, voff, 0, 0, 0);
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int bb = 0; bb < TM; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.0f;

    // lane base entry of each 32-pixel sub-tile (tap offsets are added per MFMA step)
    int lbase[TM];
#pragma unroll
    for (int bb = 0; bb < TM; ++bb) {
        const int st = wave * TM + bb;                                 // sub-tile index within the patch
        const int py = TW == 32 ? st : st * 2 + (l31 >> 4), px = TW == 32 ? l31 : (l31 & 15);
        lbase[bb] = py * S * PITCH + px;
    }
    const int fkey = (l31 >> 1) & 7;
    int foff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) foff[s] = l31 * RB + (((2 * s + hi) ^ fkey) << 4);

    const int gsh = 4 - lsp, smask = SP - 1;
    const int cin_slots_mask = smask;                                  // Cin * BYTES / 16 - 1
    // weight-ring prologue first: with FUSE1 its L2 round trip then overlaps the 1x1 stage instead of following it
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_w(s, s);

    if constexpr (FUSE1) {
        static_assert(DT != ICAF_F32 && S == 1, "the fused Bottleneck exists for the 16-bit types, stride 1");
        // ---- 1x1 convolution + SiLU over every entry of the halo patch -> second patch --------------------------------
        const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((const typename E::type*)p.w1 + g * p.w1_gs), 0, p.w1_bytes, 0x00020000);
        const unsigned w1_off0 = ((unsigned)(wave * 8 + rsub) * (unsigned)p.Kp1 + (unsigned)(lslot * VEC)) * E::BYTES;
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
            const unsigned voff = w1_off0 + (unsigned)(32 * i) * (unsigned)p.Kp1 * E::BYTES;     // (keep it a variable, see stem.hip)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w1r, (lds_ptr_t)(w1buf + (wave + 4 * i) * 1024), 16, voff, 0, 0, 0);
        }
        if constexpr (CHAIN3) {
            const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const typename E::type*)p.w2 + g * p.w2_gs), 0, p.w2_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2 * NBW; ++i) {
                const unsigned voff = ((unsigned)(wave * 8 + rsub + 32 * i) * (unsigned)p.Kp2 + (unsigned)(lslot * VEC)) * E::BYTES;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w3r, (lds_ptr_t)(w3buf + (wave + 4 * i) * 1024), 16, voff, 0, 0, 0);
            }
        }
        wait_vmcnt<0>();
        __syncthreads();                                               // input patch + 1x1 weights visible
        // the 1x1's bias in registers: loaded inside the loop below it would be a dependent L2 round trip per use
        f32x4 b1r[TN][4];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                b1r[a][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias1) b1r[a][q] = *(const f32x4*)(p.bias1 + g * p.bias1_gs + a * 32 + 8 * q + 4 * hi);
            }
        const int nidx = HH * PITCH, nsub = (nidx + 31) >> 5;
        const int gy0 = y0 - 1, gx0 = x0 - 1;
        for (int j = wave; j < nsub; j += 4) {
            f32x16 a1[TN];
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[a][r] = 0.0f;
            const int idx = (j << 5) + l31;
#pragma unroll
            for (int s = 0; s < BN / KSTEP; ++s) {                     // K = c_ input channels
                const int slot = (idx << lsp) + (((2 * s + hi) ^ (idx >> gsh)) & cin_slots_mask);
                const u32x4 fpx = *(const u32x4*)(xpatch + (slot << 4));
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    const u32x4 fw1 = *(const u32x4*)(w1buf + (a * 32) * RB + foff[s]);
                    mma_step<DT>(a1[a], fw1, fpx);
                }
            }
            const int hy = idx / PITCH, hx = idx - hy * PITCH;
            const int gy = gy0 + hy, gx = gx0 + hx;
            const bool inside = hy < HH && hx < HWD && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            if (idx < nidx) {
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = a * 32 + 8 * q + 4 * hi;        // channel of this register quad
                        float v[4] = {0.f, 0.f, 0.f, 0.f};
                        if (inside) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = a1[a][4 * q + e] + b1r[a][q][e];
                            silu4_f(v, v);
                        }
                        u32x2 pk;
                        if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                        else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                        const int slot = (idx << lsp) + (((nl >> 3) ^ (idx >> gsh)) & cin_slots_mask);
                        *(u32x2*)(halo + (slot << 4) + ((nl & 7) << 1)) = pk;
                    }
            }
        }
        // (the first barrier of the 3x3 loop below makes the second patch visible)
    }

    auto kstep = [&](const int c) {
        wait_vmcnt<(NS - 2) * NBW>();              // slice c (and, for c = 0, the halo patch issued before it) landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_w(c + NS - 1, (c + NS - 1) % NS);
        const unsigned char* b_s = ring + (c % NS) * WSTAGE;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int k0 = c * BK + s * KSTEP;                         // wave-uniform
            const int tap = k0 >> lcin;
            if (tap < 9) {
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int toff = ky * PITCH + (S == 2 ? (kx & 1) * HALF + (kx >> 1) : kx);
                const int csl = (((k0 & ((1 << lcin) - 1)) * E::BYTES) >> 4) + hi;      // logical 16-byte slot in the pixel
                u32x4 fp[TM], fw[TN];
#pragma unroll
                for (int bb = 0; bb < TM; ++bb) {
                    const int idx = lbase[bb] + toff;
                    const int slot = (idx << lsp) + ((csl ^ (idx >> gsh)) & cin_slots_mask);
                    if constexpr (ICAF_CTILE_ABL & 4) fp[bb] = u32x4{(unsigned)slot, (unsigned)c, 0x3f803f80u, 0x3f803f80u};
                    else fp[bb] = *(const u32x4*)(halo + (slot << 4));
                }
#pragma unroll
                for (int a = 0; a < TN; ++a) {
                    if constexpr (ICAF_CTILE_ABL & 4) fw[a] = u32x4{(unsigned)foff[s], (unsigned)(c + a), 0x3f803f80u, 0x3f803f80u};
                    else fw[a] = *(const u32x4*)(b_s + (a * 32) * RB + foff[s]);
                }
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int bb = 0; bb < TM; ++bb) {
                        if constexpr (ICAF_CTILE_ABL & 8) { acc[a][bb][0] += __uint_as_float(fw[a][0] ^ fp[bb][0]); }
                        else mma_step<DT>(acc[a][bb], fw[a], fp[bb]);
                    }
            }
        }
        };
    if constexpr (FUSE1) {
        constexpr int NCH = (9 * BN + BK - 1) / BK;                    // = p.nchunks (launch_bneck fills it from the same geometry)
#pragma unroll
        for (int c = 0; c < NCH; ++c) kstep(c);
    } else {
        for (int c = 0; c < p.nchunks; ++c) kstep(c);
    }
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (CHAIN3) {
        static_assert(FUSE1 && TN == 1 && BN == 32, "cv3 chained behind the fused Bottleneck: c_ = 32");
        constexpr int N3 = 2 * BN, SO3 = N3 * E::BYTES + 16;           // tile row: [cv2 | m], stride as the epilogue's
        unsigned char* T = lds;                                        // (both patches are free: every wave left the 3x3 loop)
        // (a) SiLU(acc + bias) -> tile columns [BN, 2BN), rounded to the storage type (what the epilogue stages)
        {
            const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = bias ? *(const f32x4*)(bias + 8 * q + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = 8 * q + 4 * hi;
#pragma unroll
                for (int bb = 0; bb < TM; ++bb) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[0][bb][4 * q + j] + bq[q][j];
                    apply_act4<ACT>(v, v);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= p.alpha_acc[g];
                    u32x2 pk;
                    if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
                    else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
                    *(u32x2*)(T + (wave * WM + bb * 32 + l31) * SO3 + (BN + nl) * E::BYTES) = pk;
                }
            }
        }
        __syncthreads();
        // (b) + residual (the Bottleneck's input), rounded as the two-launch form stores it; the cv2 half beside it
        {
            constexpr int VPR = BN / VEC, NIT = BM * VPR / NTHREADS;   // 4 vectors per row and half, 4 rows per thread
            const typename E::type* __restrict__ rg = p.res ? (const typename E::type*)p.res + g * p.res_gs : nullptr;
            if constexpr (ICAF_CTILE_LATE3) {                              // (A/B build: the loads where they were before round 4)
                const typename E::type* __restrict__ x2g = (const typename E::type*)p.x2 + g * p.x2_gs;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int idx = tid + it * NTHREADS, row = idx / VPR, cv = idx - row * VPR;
                    const int m = row_to_m(row);
                    const long long mm = m < 0 ? 0 : m;
                    rv[it] = rg ? *(const u32x4*)(rg + mm * p.ldr + cv * VEC) : u32x4{0u, 0u, 0u, 0u};
                    cv2v[it] = *(const u32x4*)(x2g + mm * p.ldx2 + cv * VEC);
                }
            }
            const float alpha_res = p.alpha_res[g];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * NTHREADS, row = idx / VPR, cv = idx - row * VPR;
                unsigned char* trow = T + row * SO3;
                *(u32x4*)(trow + cv * 16) = cv2v[it];
                if (rg) {
                    float v[VEC], r[VEC];
                    unpack16<DT>(*(const u32x4*)(trow + (VPR + cv) * 16), v);
                    unpack16<DT>(rv[it], r);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);      // (as the shared epilogue)
                    *(u32x4*)(trow + (VPR + cv) * 16) = pack16<DT>(v);
                }
            }
        }
        __syncthreads();
        // (c) cv3 = W3 . tile  (K = 2 c_ = 64: four MFMA steps)
        f32x16 acc3[2][TM];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < TM; ++bb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[a][bb][r] = 0.0f;
#pragma unroll
        for (int s3 = 0; s3 < N3 / KSTEP; ++s3) {
            u32x4 fp3[TM], fw3[2];
#pragma unroll
            for (int bb = 0; bb < TM; ++bb) fp3[bb] = *(const u32x4*)(T + (wave * WM + bb * 32 + l31) * SO3 + ((2 * s3 + hi) << 4));
#pragma unroll
            for (int a = 0; a < 2; ++a) fw3[a] = *(const u32x4*)(w3buf + (a * 32) * RB + foff[s3]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < TM; ++bb) mma_step<DT>(acc3[a][bb], fw3[a], fp3[bb]);
        }
        __syncthreads();                           // the tile has been consumed: the epilogue may overwrite it
        epilogue<DT, DT, BM, N3, WM, N3, ICAF_ACT_SILU, false, true, false, true>(acc3, lds, p, g, row_to_m, 0);   // (launch_bneck: Cout2 == 64, y2 in 16-byte vectors)
    } else {
        epilogue<DT, DT, BM, BN, WM, BN, ACT, false>(acc, lds, p, g, row_to_m, 0);
    }
}

template <int DT, int TH, int TW, int BN, int ACT, int S>
__global__ __launch_bounds__(NTHREADS) void ctile_kernel(const ConvP p, const int lsp, const int lcin, const int tiles_x,
                                                         const int tiles_per_img, const int halo_bytes) {
    ctile_body<DT, TH, TW, BN, ACT, S, false>(p, lsp, lcin, tiles_x, tiles_per_img, halo_bytes);
}

template <int DT, int TH, int TW, int BN, int ACT, bool CHAIN3 = false>
__global__ __launch_bounds__(NTHREADS) void bneck_kernel(const ConvP p, const int lsp, const int lcin, const int tiles_x,
                                                         const int tiles_per_img, const int halo_bytes) {
    ctile_body<DT, TH, TW, BN, ACT, 1, true, CHAIN3>(p, lsp, lcin, tiles_x, tiles_per_img, halo_bytes);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct CtileShape { int id, th, tw, bn, s; const char* tag; };
static const CtileShape kShapes[] = {
    {1, 8, 32, 32, 1, "8x32n32"}, {2, 8, 32, 64, 1, "8x32n64"}, {3, 8, 16, 64, 1, "8x16n64"}, {4, 4, 32, 64, 2, "4x32n64s2"},
    {5, 8, 16, 128, 1, "8x16n128"},
};

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

const char* ctile_tag(int shape) { return (shape >= 1 && shape <= 5) ? kShapes[shape - 1].tag : "?"; }

// 0 if shape `id` can run these args, else an error code with the reason in last_error()
int ctile_check(const icaf_conv_args* a, const ConvP& p, int shape) {
    if (shape < 1 || shape > 5) return fail(ICAF_ERR_ARG, "ctile: unknown shape %d", shape);
    const CtileShape& sh = kShapes[shape - 1];
    const int eb = a->dtype == ICAF_F32 ? 4 : 2;
    if (a->kh != 3 || a->kw != 3 || a->ph != 1 || a->pw != 1 || a->sh != sh.s || a->sw != sh.s)
        return fail(ICAF_ERR_UNSUPPORTED, "ctile %s: needs a 3x3 / pad 1 / stride %d convolution", sh.tag, sh.s);
    const int ppx = a->Cin * eb;
    if ((a->Cin & (a->Cin - 1)) || ppx < 32 || ppx > 256) return fail(ICAF_ERR_UNSUPPORTED, "ctile: Cin=%d is not a power of two with 32..256 bytes per pixel", a->Cin);
    if (a->Cout > sh.bn) return fail(ICAF_ERR_UNSUPPORTED, "ctile %s: Cout=%d > %d", sh.tag, a->Cout, sh.bn);
    if (a->act != ICAF_ACT_SILU || a->out_dtype != a->dtype) return fail(ICAF_ERR_UNSUPPORTED, "ctile: SiLU and out dtype == dtype only");
    if (p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "ctile: operand exceeds the 2 GiB buffer-descriptor range");
    if (a->pre) return fail(ICAF_ERR_UNSUPPORTED, "ctile: no pre-activation term");
    if (a->Kp % (128 / eb)) return fail(ICAF_ERR_UNSUPPORTED, "ctile: Kp must be a multiple of 128 bytes");
    return ICAF_OK;
}

template <int DT, int TH, int TW, int BN, int S, bool FUSE1 = false, bool CHAIN3 = false>
static int launch_ctile_cfg(const ConvP& p, int groups, hipStream_t s) {
    using G = HaloGeom<S, TW>;
    constexpr int EB = Elem<DT>::BYTES;
    constexpr int HH = (TH - 1) * S + 3;
    constexpr int BM = TH * TW;
    ConvP q = p;
    const int lcin = ilog2(p.Cin), lsp = ilog2(p.Cin * EB / 16);
    const int nslots = (HH * G::PITCH) << lsp;
    const int halo_bytes = ((nslots + 63) / 64) * 1024;
    const int ring = 3 * BN * 128;
    const int stage_out = BM * ((CHAIN3 ? 2 : 1) * BN * EB + 16);
    // (CHAIN3: the cv3 tile is staged over the patch, the 1x1 weights and the 3x3 ring — all free by then — but not over cv3's weights)
    if (CHAIN3 && stage_out > halo_bytes + BN * 128 + ring) return fail(ICAF_ERR_UNSUPPORTED, "icaf_bottleneck: the cv3 tile does not fit in front of its weights");
    int lds = (FUSE1 ? halo_bytes + BN * 128 : halo_bytes) + ring + (CHAIN3 ? 2 * BN * 128 : 0);
    if (lds < stage_out) lds = stage_out;
    if (lds > 160 * 1024) return fail(ICAF_ERR_UNSUPPORTED, "ctile: %d bytes of LDS needed", lds);
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    q.mtiles = p.B * tiles_x * tiles_y;
    q.ntiles = 1;
    q.nchunks = (p.K + 128 / EB - 1) / (128 / EB);
    void (*kern)(const ConvP, const int, const int, const int, const int, const int);
    if constexpr (FUSE1) kern = bneck_kernel<DT, TH, TW, BN, ICAF_ACT_SILU, CHAIN3>;
    else kern = ctile_kernel<DT, TH, TW, BN, ICAF_ACT_SILU, S>;
    ICAF_LDS_OPTIN(kern, lds);                                  // per instantiation and device: largest dynamic LDS size enabled so far
    kern<<<dim3((unsigned)q.mtiles, 1, (unsigned)groups), dim3(NTHREADS), lds, s>>>(q, lsp, lcin, tiles_x, tiles_x * tiles_y, halo_bytes);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

template <int DT>
static int launch_ctile_dt(const ConvP& p, int groups, int shape, hipStream_t s) {
    switch (shape) {
        case 1: return launch_ctile_cfg<DT, 8, 32, 32, 1>(p, groups, s);
        case 2: return launch_ctile_cfg<DT, 8, 32, 64, 1>(p, groups, s);
        case 3: return launch_ctile_cfg<DT, 8, 16, 64, 1>(p, groups, s);
        case 4: return launch_ctile_cfg<DT, 4, 32, 64, 2>(p, groups, s);
        case 5: return launch_ctile_cfg<DT, 8, 16, 128, 1>(p, groups, s);
        default: return fail(ICAF_ERR_ARG, "ctile: unknown shape %d", shape);
    }
}

// Fused Bottleneck: shape 1 (8x32 patch, c_ = 32), 2 (8x32, c_ = 64) or 3 (8x16, c_ = 64)
int launch_bneck(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    int st = ctile_check(a, p, shape);
    if (st) return st;
    if (a->dtype == ICAF_F32) return fail(ICAF_ERR_UNSUPPORTED, "icaf_bottleneck: 16-bit types only");
    if (a->Cin != a->Cout || (a->Cin != 32 && a->Cin != 64) || kShapes[shape - 1].bn != a->Cout || kShapes[shape - 1].s != 1)
        return fail(ICAF_ERR_UNSUPPORTED, "icaf_bottleneck: c_ = %d -> %d with patch shape %d is not built (c_ in {32, 64})", a->Cin, a->Cout, shape);
    if (a->w2) {          // cv3 chained behind the Bottleneck: c_ = 32 (shape 1), 64 output channels, with a residual or not
        if (shape != 1 || a->Cout2 != 64 || !p.vec_y2 || a->Kp2 != 64 || !p.x2 || p.ldx2 % 8 || ((uintptr_t)p.x2 & 15) || (!p.vec_r && a->res))
            return fail(ICAF_ERR_UNSUPPORTED, "icaf_bottleneck: chained cv3 needs patch shape 1 (c_ = 32), Cout2 = 64 written in 16-byte vectors, Kp2 = 64, an aligned cv2 half");
        if (a->dtype == ICAF_BF16) return launch_ctile_cfg<ICAF_BF16, 8, 32, 32, 1, true, true>(p, a->groups, s);
        return launch_ctile_cfg<ICAF_F16, 8, 32, 32, 1, true, true>(p, a->groups, s);
    }
    if (a->dtype == ICAF_BF16) {
        if (shape == 1) return launch_ctile_cfg<ICAF_BF16, 8, 32, 32, 1, true>(p, a->groups, s);
        if (shape == 2) return launch_ctile_cfg<ICAF_BF16, 8, 32, 64, 1, true>(p, a->groups, s);
        return launch_ctile_cfg<ICAF_BF16, 8, 16, 64, 1, true>(p, a->groups, s);
    }
    if (shape == 1) return launch_ctile_cfg<ICAF_F16, 8, 32, 32, 1, true>(p, a->groups, s);
    if (shape == 2) return launch_ctile_cfg<ICAF_F16, 8, 32, 64, 1, true>(p, a->groups, s);
    return launch_ctile_cfg<ICAF_F16, 8, 16, 64, 1, true>(p, a->groups, s);
}

int launch_ctile(const icaf_conv_args* a, const ConvP& p, int shape, hipStream_t s) {
    int st = ctile_check(a, p, shape);
    if (st) return st;
    if (a->w2) return fail(ICAF_ERR_UNSUPPORTED, "ctile: no chained 1x1 (icaf_bottleneck chains the C3's cv3)");
    if (a->dtype == ICAF_BF16) return launch_ctile_dt<ICAF_BF16>(p, a->groups, shape, s);
    if (a->dtype == ICAF_F16) return launch_ctile_dt<ICAF_F16>(p, a->groups, shape, s);
    return launch_ctile_dt<ICAF_F32>(p, a->groups, shape, s);
}

}  // namespace icaf
