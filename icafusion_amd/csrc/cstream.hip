// Persistent 3x3 / stride 1 / pad 1 convolution for the 64 -> 64 channel layers at the highest resolutions (yolov5s rows 2-6 / 12-16
// and the 80 x 80 head C3: Bottleneck.cv2 of reference models/common.py:184-194 inside C3 :216-227), 16-bit types:
//
//   y  = alpha_res * res + SiLU(conv3x3(x) + bias)                       [+ chained 1x1:  y2 = SiLU(W2 . y + bias2)]
//
// These layers move 157 MB (batch 32, both backbones) for 30 GFLOP: they are HBM-bound streaming jobs, and both existing kernels run
// them at 2x the memory floor for a different reason each (measured by ablation, lab/probes/abl_ctile.sh):
//   * igemm.hip / igemm_stream.hip gather the 3x3 window by LDS-DMA — nine times the tensor through the CU's vector-memory port,
//     whose ~10 TB/s (whole chip) is what bounds them;
//   * ctile.hip fetches a halo patch once (1.4x), but a workgroup's life is load patch -> wait -> K loop with a weight ring and a
//     barrier per slice -> epilogue (residual load, wait, store): 46 of its 72 us remain when the K loop does nothing at all.
// Here a workgroup is PERSISTENT and nothing it waits for is on the critical path of a tile:
//   * the WHOLE filter (64 x 576 weights = 72 KiB, igemm's swizzled slice layout) is resident in LDS for the workgroup's life —
//     no weight traffic, no barrier inside the K loop: 36 MFMAs per wave straight from LDS;
//   * tiles are 8 x 16 output pixels; the 10 x 18 halo patch of tile t + 1 travels by LDS-DMA into the second patch buffer
//     while tile t is multiplied (XCD-aware tile walk: neighbouring patches share their halo rows in ONE L2);
//   * the epilogue is software-pipelined: tile t's accumulators are staged in LDS (rounded to the storage type), and the
//     residual read + global stores of tile t are issued at the START of tile t + 1, before its K loop — by the time anything
//     waits on the vector-memory counter again they are long complete;
//   * optional chained 1x1 (icaf_conv_args.w2: a Bottleneck's 3x3 + shortcut followed by the next Bottleneck's 1x1, with
//     chain_keep): W2 is resident as well, the 1x1 runs from the staged tile exactly as igemm's CHAIN does.
// Arithmetic = ctile.hip / igemm.hip: K order (tap, channel), MFMA step, epilogue expressions => bit-identical results.
// Patch layout: ctile's — pixel-major 128-byte entries, 16-byte slots XOR-swizzled with (entry >> 1) & 7 — with PITCH = 24 and
// 4 x 8-pixel MFMA sub-tiles: the 16 lanes of every ds_read_b128 group then hit 16 distinct bank groups (PITCH = 8 mod 16).
#include "conv_common.h"

namespace icaf {

constexpr int CS_C = 64, CS_TH = 8, CS_TW = 16, CS_HH = CS_TH + 2, CS_HWD = CS_TW + 2, CS_PITCH = 24;
constexpr int CS_NIDX = CS_HH * CS_PITCH;                       // 240 entries of 128 bytes
constexpr int CS_PATCH = CS_NIDX * 128;                         // 30 KiB = 30 DMA instructions
constexpr int CS_WBYTES = 9 * CS_C * 128;                       // 72 KiB: nine 128-byte K slices (one per tap) x 64 rows
constexpr int CS_W2BYTES = CS_C * 128;                          // 8 KiB
constexpr int CS_SO = CS_C * 2 + 16;                            // staging row stride (the shared epilogue's)
constexpr int CS_STG = 128 * CS_SO;
constexpr int CS_LDS = CS_WBYTES + CS_W2BYTES + 2 * CS_PATCH + CS_STG;      // 161,792 bytes: one workgroup per CU
static_assert(CS_LDS <= 160 * 1024, "LDS capacity");

struct CsGeom { int tiles_x, tiles_y, ntile; };

template <int DT, bool CHAIN>
// (amdgpu_waves_per_eu: one 8-wave workgroup per CU — 72 KB of filter + two patches — is two waves per SIMD whatever the register count; told so, the
//  scheduler issues the fragment reads of several K steps ahead of their MFMAs instead of read -> s_waitcnt lgkmcnt(0) -> MFMA per step: 158 -> 155 us)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void cstream_kernel(const ConvP p, const CsGeom gm) {
    using E = Elem<DT>;
    static_assert(DT != ICAF_F32, "16-bit types");
    constexpr int RB = 128, NSTEP = 4, VEC = E::VEC;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* wbuf = lds;                                       // [9 taps][64 rows][128 bytes], slots swizzled by (row >> 1) & 7
    unsigned char* w2buf = lds + CS_WBYTES;                          // [64 rows][128 bytes]
    unsigned char* patch0 = w2buf + CS_W2BYTES;
    unsigned char* stg = patch0 + 2 * CS_PATCH;                      // [128 pixels][CS_SO]

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.z;
    const int wm = wave & 3, wn = wave >> 2;                         // sub-tile (4 rows x 8 columns of pixels) / channel half
    const int sy = wm >> 1, sx = wm & 1;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.x + g * p.x_gs), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const typename E::type*)p.w + g * p.w_gs), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- resident weights: 72 DMA instructions (tap c, rows 8 j .. 8 j + 7), igemm's swizzle through the source address -----------
    {
        const int rsub = lane >> 3;
        for (int t = wave; t < 72; t += 8) {
            const int c = t >> 3, j = t & 7;
            const int sl = (lane & 7) ^ (((j & 1) << 2) | (rsub >> 1));
            const unsigned voff = ((unsigned)(j * 8 + rsub) * (unsigned)p.Kp + (unsigned)(c * 64 + sl * VEC)) * E::BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(wbuf + (c * 64 + j * 8) * RB), 16, voff, 0, 0, 0);
        }
        if constexpr (CHAIN) {
            const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((const typename E::type*)p.w2 + g * p.w2_gs), 0, p.w2_bytes, 0x00020000);
            const int j = wave;                                      // 8 instructions, one per wave
            const int sl = (lane & 7) ^ (((j & 1) << 2) | (rsub >> 1));
            const unsigned voff = ((unsigned)(j * 8 + rsub) * (unsigned)p.Kp2 + (unsigned)(sl * VEC)) * E::BYTES;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2r, (lds_ptr_t)(w2buf + (j * 8) * RB), 16, voff, 0, 0, 0);
        }
    }

    // ---- tile walk: XCD x owns the x-th contiguous eighth of the (image, tile row, tile column) list -------------------------------
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, wgx = gridDim.x >> 3;
    const int per_xcd = (gm.ntile + 7) >> 3, t_lo = xcd * per_xcd, t_hi = min(t_lo + per_xcd, gm.ntile);
    const int per_img = gm.tiles_x * gm.tiles_y;
    auto decode = [&](int t, int& b, int& y0, int& x0) {
        b = t / per_img;
        const int r = t - b * per_img, ty = r / gm.tiles_x;
        y0 = ty * CS_TH;
        x0 = (r - ty * gm.tiles_x) * CS_TW;
    };
    // halo patch of tile t -> patch buffer `buf` (30 DMA instructions of 64 consecutive 16-byte slots; zero outside the image)
    auto issue_patch = [&](int t, int buf) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const unsigned img_off = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.ldx * (unsigned)E::BYTES;
        unsigned char* dst = patch0 + buf * CS_PATCH;
        for (int j = wave; j < CS_PATCH / 1024; j += 8) {
            const int L = (j << 6) + lane, idx = L >> 3;
            const int cs = (L & 7) ^ ((idx >> 1) & 7);
            const int hy = idx / CS_PITCH, hx = idx - hy * CS_PITCH;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = hx < CS_HWD && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            const unsigned voff = ok ? img_off + (unsigned)((gy * p.W + gx) * p.ldx) * E::BYTES + (unsigned)(cs << 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(dst + (j << 10)), 16, voff, 0, 0, 0);
        }
    };

    // per-lane constants: bias quads (registers for the workgroup's life, consumed here — see stream_core.h), fragment offsets
    f32x4 bq[4], bq2[4];
    {
        const float* __restrict__ bias = p.bias ? p.bias + g * p.bias_gs : nullptr;
        const float* __restrict__ bias2 = (CHAIN && p.bias2) ? p.bias2 + g * p.bias2_gs : nullptr;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int n = wn * 32 + 8 * qd + 4 * hi;
            const f32x4 t1 = *(const f32x4*)((bias && n < p.Cout) ? bias + n : (const float*)p.w);
            bq[qd] = (bias && n < p.Cout) ? t1 : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 t2 = *(const f32x4*)((bias2 && n < p.Cout2) ? bias2 + n : (const float*)p.w);
            bq2[qd] = (bias2 && n < p.Cout2) ? t2 : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) asm volatile("" : "+v"(bq[qd]), "+v"(bq2[qd]));
    }
    const int fkey = (l31 >> 1) & 7;
    int foff[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) foff[s] = (wn * 32 + l31) * RB + (((2 * s + hi) ^ fkey) << 4);
    // this lane's pixel of its sub-tile: row l31 >> 3, column l31 & 7; patch entry of tap (0, 0)
    const int lpy = sy * 4 + (l31 >> 3), lpx = sx * 8 + (l31 & 7);
    const int lbase = lpy * CS_PITCH + lpx;
    const int srow = wm * 32 + l31;                                  // staging row of this lane's pixel: sub-tile major

    int t = t_lo + lb;
    if (t >= t_hi) { wait_vmcnt<0>(); return; }                      // (workgroup-uniform)
    issue_patch(t, 0);

    typename E::type* __restrict__ yg = (typename E::type*)p.y + g * p.y_gs;
    const typename E::type* __restrict__ rg = p.res ? (const typename E::type*)p.res + g * p.res_gs : nullptr;
    typename E::type* __restrict__ y2g = CHAIN ? (typename E::type*)p.y2 + g * p.y2_gs : nullptr;
    const float alpha_acc = p.alpha_acc[g], alpha_res = p.alpha_res[g];

    // staging row r (sub-tile major) -> output pixel index, or -1 outside the tensor
    auto row_to_m = [&](int r, int b, int y0, int x0) {
        const int st = r >> 5, q = r & 31;
        const int gy = y0 + (st >> 1) * 4 + (q >> 3), gx = x0 + (st & 1) * 8 + (q & 7);
        return (gy < p.Ho && gx < p.Wo) ? (b * p.Ho + gy) * p.Wo + gx : -1;
    };
    // staged tile -> global memory: 128 rows x 8 vectors of 16 bytes, two per thread; `rr` = the residual vectors of the same two
    // positions (fetched a whole tile earlier, see the main loop) or nullptr
    u32x4 rres[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    auto flush = [&](typename E::type* dst, int ldd, int cout, int b, int y0, int x0, bool rr) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * 512, row = idx >> 3, cv = idx & 7;
            const int m = row_to_m(row, b, y0, x0), n = cv * VEC;
            if (m >= 0 && n < cout) {
                u32x4 sv = *(const u32x4*)(stg + row * CS_SO + cv * 16);
                if (rr) {                              // the shared epilogue's arithmetic: staged value + alpha_res * residual
                    float v[VEC], r[VEC];
                    unpack16<DT>(sv, v);
                    unpack16<DT>(rres[it], r);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                    sv = pack16<DT>(v);
                }
                *(u32x4*)(dst + (long long)m * ldd + n) = sv;
            }
        }
    };
    // residual vectors of tile (b, y0, x0) at this thread's two flush positions: plain loads from clamped addresses, issued at the
    // START of the tile and consumed after a later `s_waitcnt vmcnt(0)` that is there anyway — never a wait of their own
    auto load_res = [&](int b, int y0, int x0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * 512, row = idx >> 3, cv = idx & 7;
            const int m = row_to_m(row, b, y0, x0);
            rres[it] = *(const u32x4*)(rg + (long long)(m < 0 ? 0 : m) * p.ldr + cv * VEC);
        }
    };
    auto stage = [&](const f32x16& acc, const f32x4 (&bv)[4], float scale) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int nl = wn * 32 + 8 * qd + 4 * hi;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[4 * qd + j] + bv[qd][j] + 0.0f;
            silu4_f(v, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= scale;
            u32x2 pk;
            if constexpr (DT == ICAF_BF16) { pk[0] = pack2_bf16(v[0], v[1]); pk[1] = pack2_bf16(v[2], v[3]); }
            else { pk[0] = pack2_f16(v[0], v[1]); pk[1] = pack2_f16(v[2], v[3]); }
            *(u32x2*)(stg + srow * CS_SO + nl * E::BYTES) = pk;
        }
    };

    int cur = 0, pb = -1, py0 = 0, px0 = 0;                          // pb >= 0: a finished tile sits in the staging buffer
    while (true) {
        int b, y0, x0;
        decode(t, b, y0, x0);
        const int tn = t + wgx;
        wait_vmcnt<0>();                           // this wave's share of patch t (and of the weights); residual vectors; old stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // patch t complete; every wave has left the previous tile's K loop
        if (pb >= 0) {                             // the PREVIOUS tile's output leaves now: its stores are old by the next wait
            if constexpr (CHAIN) flush(y2g, p.ldy2, p.Cout2, pb, py0, px0, false);
            else flush(yg, p.ldy, p.Cout, pb, py0, px0, rg != nullptr);
        }
        if (rg) load_res(b, y0, x0);         // this tile's residual: in flight during the K loop
        if (tn < t_hi) issue_patch(tn, cur ^ 1);   // the next patch travels during everything below
        // ---- K loop: 9 taps x 4 MFMA steps, operands straight from the resident patch / filter --------------------------------
        const unsigned char* patch = patch0 + cur * CS_PATCH;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < 9; ++c) {
            const int ky = c / 3, kx = c - 3 * ky;
            const int idx = lbase + ky * CS_PITCH + kx;
            const unsigned char* pe = patch + (idx << 7);
            const int key = (idx >> 1) & 7;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const u32x4 fp = *(const u32x4*)(pe + (((2 * s + hi) ^ key) << 4));
                const u32x4 fw = *(const u32x4*)(wbuf + c * (CS_C * RB) + foff[s]);
                mma_step<DT>(acc, fw, fp);
            }
        }
        lds_barrier();                             // the staging buffer is free (every wave has done the previous tile's flush reads)
        stage(acc, bq, alpha_acc);
        pb = b; py0 = y0; px0 = x0;
        if constexpr (CHAIN) {
            // chained layer: y is completed NOW — staged vector + alpha_res * residual, written to y when the chain keeps it and BACK
            // into the staging tile, which the chained 1x1 consumes as stored (igemm's CHAIN + WB); y2 then leaves with the next tile
            lds_barrier();
            if (rg) { wait_vmcnt<0>(); }           // (the residual vectors: issued before this tile's K loop)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int idx = tid + it * 512, row = idx >> 3, cv = idx & 7;
                const int m = row_to_m(row, b, y0, x0), n = cv * VEC;
                if (m >= 0 && n < p.Cout) {
                    u32x4 sv = *(const u32x4*)(stg + row * CS_SO + cv * 16);
                    if (rg) {
                        float v[VEC], r[VEC];
                        unpack16<DT>(sv, v);
                        unpack16<DT>(rres[it], r);
#pragma unroll
                        for (int j = 0; j < VEC; ++j) v[j] = __builtin_fmaf(alpha_res, r[j], v[j]);
                        sv = pack16<DT>(v);
                        *(u32x4*)(stg + row * CS_SO + cv * 16) = sv;
                    }
                    if (p.keep1) *(u32x4*)(yg + (long long)m * p.ldy + n) = sv;
                }
            }
            lds_barrier();                         // the completed tile is visible
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {      // K = 64 channels of the tile: four MFMA steps
                const u32x4 fp2 = *(const u32x4*)(stg + srow * CS_SO + ((2 * s + hi) << 4));
                const u32x4 fw2 = *(const u32x4*)(w2buf + foff[s]);
                mma_step<DT>(acc2, fw2, fp2);
            }
            lds_barrier();                         // the tile has been consumed
            stage(acc2, bq2, 1.0f);
        }
        if (tn >= t_hi) break;
        t = tn;
        cur ^= 1;
    }
    wait_vmcnt<0>();
    lds_barrier();
    if constexpr (CHAIN) flush(y2g, p.ldy2, p.Cout2, pb, py0, px0, false);
    else flush(yg, p.ldy, p.Cout, pb, py0, px0, rg != nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int cstream_check(const icaf_conv_args* a, const ConvP& p) {
    if (a->dtype == ICAF_F32 || a->out_dtype != a->dtype) return fail(ICAF_ERR_UNSUPPORTED, "cstream: 16-bit types, out dtype == dtype");
    if (a->kh != 3 || a->kw != 3 || a->sh != 1 || a->sw != 1 || a->ph != 1 || a->pw != 1) return fail(ICAF_ERR_UNSUPPORTED, "cstream: 3x3 / stride 1 / pad 1 layers");
    if (a->Cin != CS_C || a->Cout > CS_C || a->Cout % 8 || a->Kp != 9 * CS_C) return fail(ICAF_ERR_UNSUPPORTED, "cstream: built for 64 -> (<= 64) channels (Cin = %d, Cout = %d, Kp = %d)", a->Cin, a->Cout, a->Kp);
    if (a->act != ICAF_ACT_SILU || a->pre) return fail(ICAF_ERR_UNSUPPORTED, "cstream: SiLU layers without a pre-activation term");
    if (p.x_bytes == 0) return fail(ICAF_ERR_UNSUPPORTED, "cstream: operand exceeds the 2 GiB buffer-descriptor range");
    if (!p.vec_y || (a->res && !p.vec_r)) return fail(ICAF_ERR_UNSUPPORTED, "cstream: y / res must take 16-byte vectors");
    if (a->w2) {
        if (a->Cout2 > CS_C || a->Cout2 % 8 || a->Kp2 != CS_C || !p.vec_y2 || a->Cout != CS_C) return fail(ICAF_ERR_UNSUPPORTED, "cstream: chained 1x1 of 64 -> (<= 64) channels with Kp2 = 64");
        if (a->res && !a->chain_keep) return fail(ICAF_ERR_UNSUPPORTED, "cstream: a residual needs chain_keep");
        if (a->chain_keep && (a->alpha_acc[0] != 1.0f || a->alpha_acc[1] != 1.0f)) return fail(ICAF_ERR_UNSUPPORTED, "cstream: chain_keep with alpha_acc != 1");
    }
    return ICAF_OK;
}

template <int DT, bool CHAIN>
static int launch_cstream_cfg(const ConvP& p, int groups, hipStream_t s) {
    CsGeom gm;
    gm.tiles_x = (p.Wo + CS_TW - 1) / CS_TW;
    gm.tiles_y = (p.Ho + CS_TH - 1) / CS_TH;
    gm.ntile = p.B * gm.tiles_x * gm.tiles_y;
    int dev = 0, cus = 256;
    ICAF_HIP(hipGetDevice(&dev));
    ICAF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int grid = (cus / groups) & ~7;                // one workgroup per CU (161 KB of LDS), the groups side by side; 8 XCDs
    if (grid < 8) grid = 8;
    ICAF_LDS_OPTIN((cstream_kernel<DT, CHAIN>), CS_LDS);
    cstream_kernel<DT, CHAIN><<<dim3((unsigned)grid, 1, (unsigned)groups), dim3(512), CS_LDS, s>>>(p, gm);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}

int launch_cstream(const icaf_conv_args* a, const ConvP& p, hipStream_t s) {
    int st = cstream_check(a, p);
    if (st) return st;
    if (a->dtype == ICAF_BF16) return a->w2 ? launch_cstream_cfg<ICAF_BF16, true>(p, a->groups, s) : launch_cstream_cfg<ICAF_BF16, false>(p, a->groups, s);
    return a->w2 ? launch_cstream_cfg<ICAF_F16, true>(p, a->groups, s) : launch_cstream_cfg<ICAF_F16, false>(p, a->groups, s);
}

}  // namespace icaf
