// Non-maximum suppression for gfx950 — replaces utils/general.py:518-607 + torchvision.ops.nms (greedy, IoU > thr).
//
// The reference filters candidates, sorts ALL of them by score and walks the sorted list greedily; only the first max_det
// (300) survivors are returned.  The walk therefore reaches at most a few hundred to a few thousand of the (up to 25 200 x nc)
// candidates, so nothing here sorts more than it walks:
//   0. hipMemsetAsync          clears the per-image score histograms and candidate counters.
//   1. nms_keys_kernel         every prediction row in parallel (B x rows / 4096 workgroups — the whole chip, not one
//                              workgroup per image): obj > conf, conf = obj * cls > conf, class filter; writes ONE 32-bit word
//                              per candidate slot (order-preserving score bits, 0 = not a candidate), the class of
//                              single-label rows, a 2178-bin histogram of the score's upper 16 bits per image (LDS
//                              atomics, flushed once per workgroup) and the candidate count of every 4096-row chunk.
//   2. nms_walk_kernel         one 1024-thread workgroup per image, rounds of
//                                select   the highest-score histogram bins that are still unvisited and hold <= 1024 (first
//                                         round) / 4096 keys — one LDS prefix scan over the histogram;
//                                gather   those candidates into LDS as unique 64-bit keys (score bits << 32 | ~slot);
//                                sort     bitonic network in LDS, descending (unique keys: the order is total; a rank sort — every
//                                         thread counting the keys above its own — was 5x slower: n^2 64-bit compares);
//                                walk     greedy suppression, 64 candidates per step: all 16 wavefronts test the step against
//                                         the kept boxes (<= max_det, in LDS) AND build the step's own 64 x 64 suppression
//                                         relation (one ballot per candidate); wave 0 then resolves the step with scalar mask
//                                         arithmetic only — no IoU in the serial part — and the kept lanes write at once;
//                              until max_det boxes are kept or the candidates (capped at max_nms, by score) are exhausted.
//                              Rounds visit disjoint, descending score ranges, each sorted by the full key, so the visiting
//                              order equals a stable descending sort of all candidates — ties fall back to the slot index,
//                              which increases with the reference's candidate index.  A histogram bin holding more than 4096
//                              keys (thousands of near-identical scores) is split by an 8-digit radix refinement of the
//                              64-bit key range — slow, exact, and never taken by real score distributions.
//                              Boxes are decoded from the prediction rows on demand; no candidate list is materialised.
//   3. nms_rank_kernel         (only when keep_idx is requested) converts the kept slots into torchvision's indices: the
//                              rank of the slot among the image's candidates = chunk counts before it + candidates before
//                              it inside its chunk.
// All box arithmetic is fp32 with contraction disabled, in the reference's operation order, so kept indices are
// bit-identical to the CPU algorithm on the same prediction tensor.
#pragma clang fp contract(off)
#include <cstring>
#include "icaf_common.h"

namespace icaf {

struct ClassMask { unsigned int w[8]; };     // classes 0..255

__device__ __forceinline__ bool class_ok(const ClassMask& cm, int use, int c) {
    return !use || (c < 256 && ((cm.w[c >> 5] >> (c & 31)) & 1u));
}

constexpr int NMS_NB = 2178;                 // histogram bins over the upper 16 bits of the ordered score word
constexpr unsigned NMS_BIN_LO = 0xB700u;     // bin 0 = everything below 2^-17 (and negative scores), bin NB-1 = [1.0, inf)
constexpr int NMS_CAP = 4096;                // keys gathered / sorted per round
constexpr int NMS_FIRST = 1024;              // target of the first round (the walk usually ends inside it)
constexpr int NMS_CHUNK_ROWS = 4096;         // prediction rows per nms_keys workgroup (one histogram flush per 4096 rows)
constexpr int MAX_KEEP = 1024;
constexpr int WALK_THREADS = 1024;

// float -> unsigned with the same ordering (negative floats below positive ones); never 0 for a candidate (c > conf, not NaN)
__device__ __forceinline__ unsigned int ordered_bits(float c) {
    const unsigned int b = __float_as_uint(c);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ int bin_of(unsigned int u) {
    const int v = (int)(u >> 16) - (int)NMS_BIN_LO;
    return v < 0 ? 0 : (v > NMS_NB - 1 ? NMS_NB - 1 : v);
}
// smallest 64-bit key of histogram bin t
__device__ __forceinline__ unsigned long long bin_floor_key(int t) {
    return t <= 0 ? 0ull : ((unsigned long long)((unsigned)(t + (int)NMS_BIN_LO) << 16)) << 32;
}

__global__ __launch_bounds__(256) void nms_keys_kernel(const float* __restrict__ pred, long long rows, int nc, float conf, int multi,
                                                       ClassMask cm, int use_cm, long long cap, int nchunks,
                                                       unsigned int* __restrict__ key32, unsigned short* __restrict__ cls16,
                                                       unsigned int* __restrict__ ghist, unsigned int* __restrict__ ncand,
                                                       unsigned int* __restrict__ chunk_cnt) {
    __shared__ unsigned int hist[NMS_NB];
    __shared__ int wsum[4];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int no = 5 + nc;
    for (int i = tid; i < NMS_NB; i += 256) hist[i] = 0;
    __syncthreads();
    const float* pb = pred + (long long)b * rows * no;
    unsigned int* kb = key32 + (long long)b * cap;
    int cnt = 0;
    for (int it0 = 0; it0 < NMS_CHUNK_ROWS / 256; it0 += 4) {
        // four rows per thread at a time: their objectness loads are issued together (the kernel is latency-bound otherwise)
        long long rr[4];
        float objs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rr[u] = (long long)chunk * NMS_CHUNK_ROWS + (it0 + u) * 256 + tid;
            objs[u] = rr[u] < rows ? pb[rr[u] * no + 4] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long r = rr[u];
            if (r >= rows) continue;
            const float* p = pb + r * no;
            const float obj = objs[u];
            if (multi) {
                for (int j = 0; j < nc; ++j) {
                    unsigned int k = 0;
                    if (obj > conf) {
                        const float c = p[5 + j] * obj;
                        if (c > conf && class_ok(cm, use_cm, j)) k = ordered_bits(c);
                    }
                    kb[r * nc + j] = k;
                    if (k) { atomicAdd(&hist[bin_of(k)], 1u); ++cnt; }
                }
            } else {
                unsigned int k = 0;
                int best_j = 0;
                if (obj > conf) {
                    float best = -INFINITY;
                    for (int j = 0; j < nc; ++j) {
                        const float c = p[5 + j] * obj;
                        if (c > best) { best = c; best_j = j; }
                    }
                    if (best > conf && class_ok(cm, use_cm, best_j)) k = ordered_bits(best);
                }
                kb[r] = k;
                cls16[(long long)b * rows + r] = (unsigned short)best_j;
                if (k) { atomicAdd(&hist[bin_of(k)], 1u); ++cnt; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) wsum[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        chunk_cnt[(long long)b * nchunks + chunk] = (unsigned)total;
        if (total) atomicAdd(&ncand[b], (unsigned)total);
    }
    for (int i = tid; i < NMS_NB; i += 256) {
        const unsigned int v = hist[i];
        if (v) atomicAdd(&ghist[(long long)b * NMS_NB + i], v);
    }
}

__device__ __forceinline__ bool iou_gt_exact(float inter, float uni, float thr) {
    // torchvision nms_kernel: ovr = inter / (area_a + area_b - inter); suppress if ovr > thr
    const float ovr = inter / uni;
    return ovr > thr;
}
// IoU(a, b) > thr with the reference's arithmetic (w = max(0, min(x2) - max(x1)), inter = w * h, union = area_a + area_b - inter,
// all fp32, no contraction) and the reference's DECISION, without its division in the common case: the correctly rounded quotient
// q = fl(inter / union) differs from the exact ratio by at most 2^-24 relative, so inter > thr * union * (1 + 2^-20) implies q > thr
// and inter < thr * union * (1 - 2^-20) implies q <= thr (the two products add < 2^-22 of rounding); only inside that band — and for
// degenerate unions (<= 0, NaN) or thr <= 0 — the division itself decides.  ~12 VALU instructions instead of ~25.
__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2,
                                       float by2, float barea, float thr) {
    const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1), xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float uni = aarea + barea - inter;
    const float t = thr * uni;
    const bool sure_yes = inter > t * 1.00000095367431640625f;      // 1 + 2^-20
    const bool sure_no = inter < t * 0.99999904632568359375f;       // 1 - 2^-20
    const bool regular = uni > 0.0f && thr > 0.0f;                    // (false for NaN as well)
    if (__builtin_expect(!__all((sure_yes || sure_no) && regular), 0)) {
        const bool exact = iou_gt_exact(inter, uni, thr);
        return (regular && (sure_yes || sure_no)) ? sure_yes : exact;
    }
    return sure_yes;
}

// inclusive prefix sum over the 1024 threads of the walk kernel (v per thread); total returned through `total`
__device__ __forceinline__ unsigned int block_incl_scan_1024(unsigned int v, unsigned int* wsum, unsigned int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned int s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc;
}

struct WalkSmem {
    unsigned long long keys[NMS_CAP];                                     // gathered / sorted keys of the round
    float kx1[MAX_KEEP], ky1[MAX_KEEP], kx2[MAX_KEEP], ky2[MAX_KEEP], kar[MAX_KEEP];      // kept boxes (class-offset), areas
    float cx1[WALK_THREADS], cy1[WALK_THREADS], cx2[WALK_THREADS], cy2[WALK_THREADS], car[WALK_THREADS];   // decoded candidates
    unsigned int kslot[MAX_KEEP];                                         // candidate slot of every kept box
    unsigned int kpos[MAX_KEEP];                                          // its position in the sorted order
    unsigned int hist[NMS_NB];
    unsigned int dh[256];
    unsigned int wsum[16];
    unsigned long long deadmask;
    unsigned long long supp[64];                                          // supp[j]: lanes of the step that candidate j suppresses
    unsigned long long lo, hi;
    unsigned int batch_n, sel_cnt, sel_val, nkept;
    int mode_b_bin;
};

// decode the box of candidate slot `slot`: xywh -> xyxy exactly as the reference (utils/general.py:332-339)
__device__ __forceinline__ void decode_slot(const float* __restrict__ pb, const unsigned short* __restrict__ cb, int nc, int multi,
                                            unsigned int slot, float& x1, float& y1, float& x2, float& y2, float& sc, int& cl) {
    const unsigned int row = multi ? slot / (unsigned)nc : slot;
    cl = multi ? (int)(slot - row * (unsigned)nc) : (int)cb[row];
    const float* p = pb + (long long)row * (5 + nc);
    const float hw = p[2] / 2.0f, hh = p[3] / 2.0f;
    x1 = p[0] - hw; y1 = p[1] - hh; x2 = p[0] + hw; y2 = p[1] + hh;
    sc = p[5 + cl] * p[4];
}

#ifdef ICAF_NMS_DEBUG
#define NMS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define NMS_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(WALK_THREADS) void nms_walk_kernel(const float* __restrict__ pred, long long rows, int nc, int multi,
                                                                long long cap, const unsigned int* __restrict__ key32,
                                                                const unsigned short* __restrict__ cls16,
                                                                const unsigned int* __restrict__ ghist,
                                                                const unsigned int* __restrict__ ncand, float iou_thr, float cls_off,
                                                                int max_det, int max_nms, float* __restrict__ det,
                                                                int* __restrict__ count, int* __restrict__ keep_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char walk_smem_raw[];
    WalkSmem& sm = *reinterpret_cast<WalkSmem*>(walk_smem_raw);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pb = pred + (long long)b * rows * (5 + nc);
    const unsigned int* kb = key32 + (long long)b * cap;
    const unsigned short* cb = cls16 + (long long)b * rows;
#ifdef ICAF_NMS_DEBUG
    __shared__ long long stamps[16];
    for (int i = tid; i < 16; i += WALK_THREADS) stamps[i] = 0;
    __syncthreads();
#endif
    NMS_STAMP(0);
    const unsigned int n_all = ncand[b];
    const bool reordered = n_all > (unsigned)max_nms;             // reference re-indexes x by score rank in that case (:586)
    const unsigned int n = reordered ? (unsigned)max_nms : n_all;
    for (int i = tid; i < NMS_NB; i += WALK_THREADS) sm.hist[i] = ghist[(long long)b * NMS_NB + i];
    if (tid == 0) { sm.nkept = 0; sm.deadmask = 0ull; sm.mode_b_bin = -1; }
    __syncthreads();
    int hib = NMS_NB;                              // histogram bins [0, hib) are entirely unvisited
    unsigned long long hi = ~0ull;                 // keys >= hi have been visited
    unsigned int processed = 0, nkept = 0;
    unsigned int in_bin_left = 0;                  // refinement mode: unvisited keys left in bin hib (which is then partially visited)
    int round = 0;
    while (nkept < (unsigned)max_det && processed < n) {
        // ------------------------------------------------------------------ select [lo, hi)
        unsigned long long lo = 0ull;
        unsigned int expect = 0;
        int mode_b = -1;
        if (in_bin_left == 0) {
            // P(d) = keys in bins (hib-1-d .. hib-1]; largest d with P(d) <= target
            unsigned int target = round == 0 ? NMS_FIRST : NMS_CAP;
            for (;;) {
                unsigned int loc[3], s = 0;
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const int d = tid * 3 + e;
                    loc[e] = d < hib ? sm.hist[hib - 1 - d] : 0u;
                    s += loc[e];
                }
                unsigned int total;
                const unsigned int incl = block_incl_scan_1024(s, sm.wsum, total);
                if (tid == 0) { sm.sel_cnt = 0; sm.sel_val = 0; }
                __syncthreads();
                unsigned int run = incl - s, nle = 0, last = 0;
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    run += loc[e];
                    if (tid * 3 + e < hib && run <= target) { ++nle; last = run; }
                }
                if (nle) { atomicAdd(&sm.sel_cnt, nle); atomicMax(&sm.sel_val, last); }
                __syncthreads();
                const unsigned int dcnt = sm.sel_cnt, val = sm.sel_val;     // dcnt bins from the top hold `val` keys in total
                __syncthreads();
                if (total == 0) { expect = 0; hib = 0; break; }             // nothing left anywhere
                if (val > 0) { expect = val; hib -= (int)dcnt; lo = bin_floor_key(hib); break; }
                // the next non-empty bin alone exceeds the target
                hib -= (int)dcnt;                                            // skip the empty bins above it
                if (target < NMS_CAP) { target = NMS_CAP; continue; }
                mode_b = hib - 1;                                            // > NMS_CAP keys in one bin: refine inside it
                break;
            }
            if (expect == 0 && mode_b < 0) break;
            if (mode_b >= 0) { in_bin_left = sm.hist[mode_b]; hib = mode_b; }      // (bins between it and `hi` are empty: hi stays a valid bound)
        }
        if (in_bin_left != 0) {
            // radix refinement inside bin `hib` over the keys in [bin_floor_key(hib), hi): find lo with 1 <= #[lo, hi) <= NMS_CAP
            const unsigned long long blo = bin_floor_key(hib);
            if (in_bin_left <= NMS_CAP) {
                lo = blo; expect = in_bin_left;
            } else {
                unsigned long long prefix = 0ull;      // the digits chosen so far (upper bits of lo)
                unsigned int acc = 0;                  // keys above the current prefix range that the batch already includes
                for (int level = 0; level < 8; ++level) {
                    const int shift = 56 - 8 * level;
                    for (int i = tid; i < 256; i += WALK_THREADS) sm.dh[i] = 0;
                    __syncthreads();
                    for (long long i = tid; i < cap; i += WALK_THREADS) {
                        const unsigned int k32 = kb[i];
                        if (!k32) continue;
                        const unsigned long long k = ((unsigned long long)k32 << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
                        if (k < blo || k >= hi) continue;
                        if (level > 0 && (k >> (shift + 8)) != (prefix >> (shift + 8))) continue;
                        atomicAdd(&sm.dh[(unsigned)(k >> shift) & 255u], 1u);
                    }
                    __syncthreads();
                    if (tid == 0) {
                        int dmax = 255;
                        while (dmax > 0 && sm.dh[dmax] == 0) --dmax;
                        if (acc + sm.dh[dmax] > NMS_CAP) {                   // descend into the top digit
                            sm.lo = prefix | ((unsigned long long)dmax << shift);
                            sm.sel_val = 0;
                        } else {
                            unsigned int suf = 0;
                            int d = dmax;
                            while (d >= 0 && acc + suf + sm.dh[d] <= NMS_CAP) { suf += sm.dh[d]; --d; }
                            sm.lo = prefix | ((unsigned long long)(d + 1) << shift);
                            sm.sel_val = acc + suf;                          // >= 1
                        }
                    }
                    __syncthreads();
                    prefix = sm.lo;
                    const unsigned int got = sm.sel_val;
                    __syncthreads();
                    if (got) { lo = prefix; expect = got; break; }
                }
                if (lo < blo) lo = blo;
            }
        }
        if (round == 0) NMS_STAMP(1);
        // ------------------------------------------------------------------ gather the keys of [lo, hi) into LDS (any order)
        if (tid == 0) sm.batch_n = 0;
        __syncthreads();
        for (long long i0 = 0; i0 < cap; i0 += 8 * WALK_THREADS) {
            unsigned int k32[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                      // eight independent loads in flight per thread (the scan is latency-bound)
                const long long i = i0 + u * WALK_THREADS + tid;
                k32[u] = i < cap ? kb[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long i = i0 + u * WALK_THREADS + tid;
                const unsigned long long k = ((unsigned long long)k32[u] << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
                const bool take = k32[u] != 0u && k >= lo && k < hi;
                const unsigned long long m = __ballot(take);
                if (m) {
                    unsigned int base = 0;
                    if (lane == 0) base = atomicAdd(&sm.batch_n, (unsigned)__popcll(m));
                    base = __shfl(base, 0);
                    if (take) {
                        const unsigned int slot = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
                        if (slot < NMS_CAP) sm.keys[slot] = k;
                    }
                }
            }
        }
        __syncthreads();
        const unsigned int bn = sm.batch_n < NMS_CAP ? sm.batch_n : NMS_CAP;       // (== expect by construction)
        if (round == 0) NMS_STAMP(2);
        // ------------------------------------------------------------------ bitonic sort in LDS, descending (padding = key 0, the smallest)
        // (a rank sort — every thread counting the keys greater than its own through broadcast reads — measured 150 k cycles for 1020
        //  keys: 64-bit compares x n^2 are VALU-bound; the network below is 55 barrier stages for 1024 keys, 78 for 4096)
        {
            unsigned int P = 64;
            while (P < bn) P <<= 1;
            for (unsigned int i = bn + tid; i < P; i += WALK_THREADS) sm.keys[i] = 0ull;
            __syncthreads();
            for (unsigned int k = 2; k <= P; k <<= 1)
                for (unsigned int j = k >> 1; j > 0; j >>= 1) {
                    for (unsigned int t = tid; t < P / 2; t += WALK_THREADS) {
                        const unsigned int i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
                        const unsigned int l = i | j;
                        const bool up = (i & k) != 0u;                // ascending sub-blocks of the network; the last merge is all-descending
                        const unsigned long long a = sm.keys[i], c = sm.keys[l];
                        if ((a < c) != up) { sm.keys[i] = c; sm.keys[l] = a; }
                    }
                    __syncthreads();
                }
        }
        if (round == 0) NMS_STAMP(3);
        // ------------------------------------------------------------------ greedy walk over the sorted round
        const unsigned int limit = (n - processed) < bn ? (n - processed) : bn;     // max_nms cap (by score rank)
        for (unsigned int base = 0; base < limit && nkept < (unsigned)max_det; base += WALK_THREADS) {
            {
                const unsigned int pos = base + tid;
                if (pos < limit) {
                    const unsigned int slot = 0xffffffffu - (unsigned)(sm.keys[pos] & 0xffffffffull);
                    float x1, y1, x2, y2, sc;
                    int cl;
                    decode_slot(pb, cb, nc, multi, slot, x1, y1, x2, y2, sc, cl);
                    const float off = (float)cl * cls_off;                 // boxes + cls * max_wh (reference :589-590)
                    x1 = x1 + off; y1 = y1 + off; x2 = x2 + off; y2 = y2 + off;
                    sm.cx1[tid] = x1; sm.cy1[tid] = y1; sm.cx2[tid] = x2; sm.cy2[tid] = y2;
                    sm.car[tid] = (x2 - x1) * (y2 - y1);
                }
            }
            __syncthreads();
            const unsigned int span = (limit - base) < WALK_THREADS ? (limit - base) : WALK_THREADS;
            for (unsigned int c0 = 0; c0 < span && nkept < (unsigned)max_det; c0 += 64) {
                const unsigned int ci = c0 + lane;
                const bool valid = ci < span;
                float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f;
                if (valid) { x1 = sm.cx1[ci]; y1 = sm.cy1[ci]; x2 = sm.cx2[ci]; y2 = sm.cy2[ci]; area = sm.car[ci]; }
                // phase A: the 16 wavefronts test the step against interleaved sixteenths of the kept list
                bool dead = !valid;
                for (unsigned int k = wave; k < nkept && !__all(dead); k += 16) {
                    const bool hit = iou_gt(sm.kx1[k], sm.ky1[k], sm.kx2[k], sm.ky2[k], sm.kar[k], x1, y1, x2, y2, area, iou_thr);    // (all lanes: wave-uniform control flow)
                    dead = dead || hit;
                }
                const unsigned long long dm = __ballot(dead);
                if (lane == 0 && dm) atomicOr(&sm.deadmask, dm);
                // ... and the step's own 64 x 64 suppression relation, four candidates j per wavefront: supp[j] = the lanes that
                // overlap candidate j (box broadcast from lane j's registers); the serial pass below then needs no IoU at all
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = wave * 4 + u;
                    const float jx1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), j));
                    const float jy1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y1), j));
                    const float jx2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x2), j));
                    const float jy2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y2), j));
                    const float jar = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, area), j));
                    const bool over = iou_gt(jx1, jy1, jx2, jy2, jar, x1, y1, x2, y2, area, iou_thr);
                    const unsigned long long m = __ballot(over && valid);
                    if (lane == 0) sm.supp[j] = m;
                }
                __syncthreads();
                if (wave == 0) {
                    // phase B: resolve the step in order — a kept candidate j removes the LATER lanes of supp[j].  Lane l holds supp[l]
                    // in registers; per kept box the serial pass is a find-first-set, two v_readlane and scalar mask arithmetic.
                    const unsigned long long mysupp = sm.supp[lane];
                    const int slo = (int)(unsigned)(mysupp & 0xffffffffull), shi = (int)(unsigned)(mysupp >> 32);
                    unsigned long long alive = ~sm.deadmask, keptmask = 0ull;
                    const unsigned int nkept0 = nkept;
                    while (alive && nkept < (unsigned)max_det) {
                        const int j = __ffsll((long long)alive) - 1;          // wave-uniform
                        keptmask |= 1ull << j;
                        ++nkept;
                        const unsigned long long sj = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(shi, j) << 32) |
                                                      (unsigned long long)(unsigned)__builtin_amdgcn_readlane(slo, j);
                        alive &= ~((sj & ~((2ull << j) - 1ull)) | (1ull << j));
                    }
                    if ((keptmask >> lane) & 1ull) {
                        const unsigned int kk = nkept0 + (unsigned)__popcll(keptmask & ((1ull << lane) - 1ull));
                        sm.kx1[kk] = x1; sm.ky1[kk] = y1; sm.kx2[kk] = x2; sm.ky2[kk] = y2; sm.kar[kk] = area;
                        sm.kslot[kk] = 0xffffffffu - (unsigned)(sm.keys[base + ci] & 0xffffffffull);
                        sm.kpos[kk] = processed + base + ci;
                    }
                    if (lane == 0) { sm.nkept = nkept; sm.deadmask = 0ull; }
                }
                __syncthreads();
                nkept = sm.nkept;
            }
        }
        if (round == 0) NMS_STAMP(4);
        processed += bn;
        hi = lo;
        if (in_bin_left != 0) {
            in_bin_left = in_bin_left > bn ? in_bin_left - bn : 0u;       // bin `hib` done once nothing is left in it
        }
        ++round;
        if (bn == 0) break;                                                // defensive: no progress is impossible by construction
    }
    NMS_STAMP(5);
#ifdef ICAF_NMS_DEBUG
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0)
        printf("nms_walk b0: n_all=%u rounds=%d processed=%u kept=%u | clocks: select %lld gather %lld sort %lld walk(round0) %lld later rounds %lld\n", n_all, round,
               processed, nkept, stamps[1] - stamps[0], stamps[2] - stamps[1], stamps[3] - stamps[2], stamps[4] - stamps[3], stamps[5] - stamps[4]);
#endif
    // ---------------------------------------------------------------------- outputs, all kept boxes in parallel
    if (tid == 0) count[b] = (int)nkept;
    for (unsigned int k = tid; k < nkept; k += WALK_THREADS) {
        float x1, y1, x2, y2, sc;
        int cl;
        decode_slot(pb, cb, nc, multi, sm.kslot[k], x1, y1, x2, y2, sc, cl);
        float* o = det + ((long long)b * max_det + k) * 6;
        o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = sc; o[5] = (float)cl;
        if (keep_idx) keep_idx[(long long)b * max_det + k] = (int)(reordered ? sm.kpos[k] : sm.kslot[k]);
    }
}

// keep_idx holds candidate SLOTS (row * nc + class) after the walk; torchvision returns indices into the candidate list,
// i.e. the rank of the slot among the image's candidates (slots are visited in candidate order).  One wavefront per kept box.
__global__ __launch_bounds__(256) void nms_rank_kernel(const unsigned int* __restrict__ key32, long long cap, int ncm, int nchunks,
                                                       const unsigned int* __restrict__ chunk_cnt,
                                                       const unsigned int* __restrict__ ncand, const int* __restrict__ count,
                                                       int max_det, int max_nms, int* __restrict__ keep_idx) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= count[b] || ncand[b] > (unsigned)max_nms) return;           // re-ordered images already hold sorted positions
    const unsigned int slot = (unsigned)keep_idx[(long long)b * max_det + k];
    const unsigned int per_chunk = (unsigned)NMS_CHUNK_ROWS * (unsigned)ncm;
    const unsigned int chunk = slot / per_chunk;
    unsigned int r = 0;
    for (unsigned int c = lane; c < chunk; c += 64) r += chunk_cnt[(long long)b * nchunks + c];
    const unsigned int* kb = key32 + (long long)b * cap;
    for (unsigned int i = chunk * per_chunk + lane; i < slot; i += 64) r += kb[i] != 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
    if (lane == 0) keep_idx[(long long)b * max_det + k] = (int)r;
}

struct NmsWs {
    unsigned int* key32; unsigned short* cls16; unsigned int* ghist; unsigned int* ncand; unsigned int* chunk_cnt;
    size_t zero_bytes; int nchunks; size_t total;
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int nms_layout(int B, long long rows, int nc, int multi, void* base, NmsWs& ws) {
    if (B < 1 || rows < 1 || nc < 1) return fail(ICAF_ERR_ARG, "icaf_nms: bad B/rows/nc");
    if (nc > 65535) return fail(ICAF_ERR_UNSUPPORTED, "icaf_nms: at most 65535 classes");
    const long long cap = rows * ((multi && nc > 1) ? nc : 1);
    if (cap > 0xfffffff0LL || (long long)B > 65535) return fail(ICAF_ERR_UNSUPPORTED, "icaf_nms: rows*nc exceeds 2^32 candidate slots per image");
    size_t off = 0;
    unsigned char* p = (unsigned char*)base;
    auto take = [&](size_t bytes) { unsigned char* r = p ? p + off : nullptr; off = align_up(off + bytes, 256); return r; };
    ws.nchunks = (int)((rows + NMS_CHUNK_ROWS - 1) / NMS_CHUNK_ROWS);
    ws.ghist = (unsigned int*)take((size_t)B * NMS_NB * 4 + (size_t)B * 4);       // histograms + candidate counters: cleared per call
    ws.ncand = ws.ghist ? ws.ghist + (size_t)B * NMS_NB : nullptr;
    ws.zero_bytes = (size_t)B * NMS_NB * 4 + (size_t)B * 4;
    ws.key32 = (unsigned int*)take((size_t)B * cap * 4);
    ws.cls16 = (unsigned short*)take((size_t)B * rows * 2);
    ws.chunk_cnt = (unsigned int*)take((size_t)B * ws.nchunks * 4);
    ws.total = off;
    return ICAF_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Validation statistics (reference test.py:196-230): TP flags of every detection at every IoU threshold
// ---------------------------------------------------------------------------------------------------------------
// One workgroup per image.  Detections arrive as the NMS output block (letterboxed pixel space) and are first mapped to
// the native image (scale_coords + clip_coords, utils/general.py:386-407: subtract the pad, divide by the gain, clip).
// Phase 1, one thread per detection: best IoU over the labels of its class and the FIRST label reaching it (box_iou +
// torch max).  Phase 2, one thread: detections in index order claim their best label if nobody has and the IoU exceeds
// iouv[0]; a claim marks correct[i][t] = best > iouv[t].  (The reference walks class by class; claims of different
// classes never touch the same label, so index order gives the same flags.)  fp32, no contraction: the IoU values are
// bit-identical to the numpy statement of the same formula (icafusion_amd/utils/metrics.py, the test oracle).
constexpr int MATCH_MAX_DET = 1024, MATCH_MAX_LABELS = 2048;

__global__ __launch_bounds__(256) void match_predictions_kernel(const float* __restrict__ det, const int* __restrict__ count, int max_det,
                                                                const float* __restrict__ labels, const int* __restrict__ label_off,
                                                                const float* __restrict__ scale, const float* __restrict__ iouv, int T,
                                                                unsigned char* __restrict__ correct, float* __restrict__ predn) {
    __shared__ float best[MATCH_MAX_DET];
    __shared__ int arg[MATCH_MAX_DET];
    __shared__ unsigned char claimed[MATCH_MAX_LABELS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(count[b], max_det), l0 = label_off[b], nl = label_off[b + 1] - l0;
    const float* db = det + (long long)b * max_det * 6;
    float gain = 1.0f, padx = 0.0f, pady = 0.0f, w0 = 3.0e38f, h0 = 3.0e38f;
    if (scale) { gain = scale[b * 5]; padx = scale[b * 5 + 1]; pady = scale[b * 5 + 2]; w0 = scale[b * 5 + 3]; h0 = scale[b * 5 + 4]; }
    for (int i = tid; i < nl; i += 256) claimed[i] = 0;
    for (int i = tid; i < n; i += 256) {
        float x1 = db[i * 6], y1 = db[i * 6 + 1], x2 = db[i * 6 + 2], y2 = db[i * 6 + 3];
        const float cls = db[i * 6 + 5];
        if (scale) {
            x1 = (x1 - padx) / gain; x2 = (x2 - padx) / gain; y1 = (y1 - pady) / gain; y2 = (y2 - pady) / gain;
            x1 = fminf(fmaxf(x1, 0.0f), w0); x2 = fminf(fmaxf(x2, 0.0f), w0);
            y1 = fminf(fmaxf(y1, 0.0f), h0); y2 = fminf(fmaxf(y2, 0.0f), h0);
        }
        if (predn) {
            float* o = predn + ((long long)b * max_det + i) * 4;
            o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
        }
        const float area_a = (x2 - x1) * (y2 - y1);
        float bst = -1.0f;
        int ba = -1;
        for (int j = 0; j < nl; ++j) {
            const float* lb = labels + (long long)(l0 + j) * 5;
            if (lb[0] != cls) continue;
            const float iw = fmaxf(fminf(x2, lb[3]) - fmaxf(x1, lb[1]), 0.0f), ih = fmaxf(fminf(y2, lb[4]) - fmaxf(y1, lb[2]), 0.0f);
            const float inter = iw * ih, area_b = (lb[3] - lb[1]) * (lb[4] - lb[2]);
            const float iou = inter / (area_a + area_b - inter);
            if (iou > bst) { bst = iou; ba = j; }
        }
        best[i] = bst;
        arg[i] = ba;
        for (int t = 0; t < T; ++t) correct[((long long)b * max_det + i) * T + t] = 0;
    }
    __syncthreads();
    if (tid == 0) {
        const float thr0 = iouv[0];
        int found = 0;
        for (int i = 0; i < n && found < nl; ++i) {
            if (arg[i] < 0 || !(best[i] > thr0) || claimed[arg[i]]) continue;
            claimed[arg[i]] = 1;
            ++found;
            for (int t = 0; t < T; ++t) correct[((long long)b * max_det + i) * T + t] = best[i] > iouv[t] ? 1 : 0;
        }
    }
}

}  // namespace icaf

using namespace icaf;

extern "C" int icaf_nms_workspace_bytes(int B, long long rows, int nc, int multi_label, size_t* bytes) {
    if (!bytes) return fail(ICAF_ERR_ARG, "icaf_nms_workspace_bytes: null pointer");
    NmsWs ws;
    int st = nms_layout(B, rows, nc, multi_label, nullptr, ws);
    if (st) return st;
    *bytes = ws.total;
    return ICAF_OK;
}

extern "C" int icaf_nms(const float* pred, int B, long long rows, int nc, float conf_thres, float iou_thres, int multi_label, int agnostic,
                        const int* classes_host, int n_classes, int max_det, int max_nms, float max_wh, float* det, int* count,
                        int* keep_idx, void* workspace, size_t workspace_bytes, icaf_stream_t s) {
    if (!pred || !det || !count || !workspace) return fail(ICAF_ERR_ARG, "icaf_nms: null pointer");
    if (max_det < 1 || max_det > MAX_KEEP) return fail(ICAF_ERR_ARG, "icaf_nms: max_det must be in [1, %d]", MAX_KEEP);
    if (max_nms < 1) return fail(ICAF_ERR_ARG, "icaf_nms: max_nms must be positive");
    if (n_classes < 0 || (n_classes > 0 && !classes_host)) return fail(ICAF_ERR_ARG, "icaf_nms: class filter of %d ids without a list", n_classes);
    if (((uintptr_t)workspace & 255) != 0) return fail(ICAF_ERR_ARG, "icaf_nms: workspace must be 256-byte aligned");
    const int multi = multi_label && nc > 1;
    NmsWs ws;
    int st = nms_layout(B, rows, nc, multi, workspace, ws);
    if (st) return st;
    if (ws.total > workspace_bytes) return fail(ICAF_ERR_ARG, "icaf_nms: workspace too small (%zu < %zu)", workspace_bytes, ws.total);
    ClassMask cm;
    memset(&cm, 0, sizeof(cm));
    for (int i = 0; i < n_classes; ++i) {
        const int c = classes_host[i];
        if (c < 0 || c > 255) return fail(ICAF_ERR_UNSUPPORTED, "icaf_nms: class filter supports ids 0..255");
        cm.w[c >> 5] |= 1u << (c & 31);
    }
    const long long cap = rows * (multi ? nc : 1);
    hipStream_t hs = S(s);
    static std::atomic<bool> attr_set[ICAF_MAX_DEVICES];      // hipFuncSetAttribute is per device
    int dev = 0;
    ICAF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= ICAF_MAX_DEVICES) return fail(ICAF_ERR_UNSUPPORTED, "icaf_nms: device ordinal %d", dev);
    if (!attr_set[dev]) {
        ICAF_HIP(hipFuncSetAttribute((const void*)nms_walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(WalkSmem)));
        attr_set[dev] = true;
    }
    ICAF_HIP(hipMemsetAsync(ws.ghist, 0, ws.zero_bytes, hs));
    nms_keys_kernel<<<dim3((unsigned)ws.nchunks, (unsigned)B), dim3(256), 0, hs>>>(pred, rows, nc, conf_thres, multi, cm, n_classes > 0 ? 1 : 0,
                                                                                   cap, ws.nchunks, ws.key32, ws.cls16, ws.ghist, ws.ncand,
                                                                                   ws.chunk_cnt);
    ICAF_LAUNCH_CHECK();
    nms_walk_kernel<<<dim3((unsigned)B), dim3(WALK_THREADS), sizeof(WalkSmem), hs>>>(pred, rows, nc, multi, cap, ws.key32, ws.cls16, ws.ghist,
                                                                                     ws.ncand, iou_thres, agnostic ? 0.0f : max_wh, max_det,
                                                                                     max_nms, det, count, keep_idx);
    ICAF_LAUNCH_CHECK();
    if (keep_idx) {
        nms_rank_kernel<<<dim3((unsigned)((max_det + 3) / 4), (unsigned)B), dim3(256), 0, hs>>>(ws.key32, cap, multi ? nc : 1, ws.nchunks,
                                                                                              ws.chunk_cnt, ws.ncand, count, max_det, max_nms,
                                                                                              keep_idx);
        ICAF_LAUNCH_CHECK();
    }
    return ICAF_OK;
}

extern "C" int icaf_match_predictions(const float* det, const int* count, int B, int max_det, const float* labels, const int* label_off,
                                      int max_labels_per_image, const float* scale, const float* iouv, int T, unsigned char* correct,
                                      float* predn, icaf_stream_t s) {
    if (!det || !count || !label_off || !iouv || !correct) return fail(ICAF_ERR_ARG, "icaf_match_predictions: null pointer");
    if (B < 1 || T < 1 || max_det < 1 || max_det > MATCH_MAX_DET) return fail(ICAF_ERR_ARG, "icaf_match_predictions: max_det must be in [1, %d]", MATCH_MAX_DET);
    if (max_labels_per_image > MATCH_MAX_LABELS) return fail(ICAF_ERR_UNSUPPORTED, "icaf_match_predictions: at most %d labels per image", MATCH_MAX_LABELS);
    if (max_labels_per_image > 0 && !labels) return fail(ICAF_ERR_ARG, "icaf_match_predictions: labels missing");
    match_predictions_kernel<<<dim3((unsigned)B), dim3(256), 0, S(s)>>>(det, count, max_det, labels, label_off, scale, iouv, T, correct, predn);
    ICAF_LAUNCH_CHECK();
    return ICAF_OK;
}
