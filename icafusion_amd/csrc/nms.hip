// Non-maximum suppression for gfx950 — replaces utils/general.py:518-607 + torchvision.ops.nms (greedy, IoU > thr).
//
// Three launches per batch, all images in parallel, nothing returns to the host:
//   1. nms_candidates_kernel  one 1024-thread workgroup per image walks the prediction rows in order and compacts the
//                             candidates (obj > conf, conf = obj*cls > conf, class filter) with a workgroup prefix
//                             scan, so the candidate list has exactly the reference's order; emits a 64-bit sort
//                             key (score bits << 32 | ~index) per candidate.
//   2. rocPRIM segmented radix sort (descending) of the keys — unique keys make the order total: descending score,
//      ties by ascending candidate index, i.e. a stable descending sort like torchvision's.
//   3. nms_greedy_kernel      one workgroup per image visits candidates in sorted order, 64 at a time: four wavefronts
//                             test the chunk against the (<= max_det) kept boxes held in LDS, wave 0 resolves the chunk
//                             internally, and the walk stops as soon as max_det boxes are kept — the reference computes
//                             the full keep list and truncates it, which yields the same first max_det boxes.
// All box arithmetic is fp32 with contraction disabled, in the reference's operation order, so kept indices are
// bit-identical to the CPU algorithm on the same prediction tensor.
#pragma clang fp contract(off)
#include <cstring>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include "icaf_common.h"

namespace icaf {

struct ClassMask { unsigned int w[8]; };     // classes 0..255

__device__ __forceinline__ bool class_ok(const ClassMask& cm, int use, int c) {
    return !use || (c < 256 && ((cm.w[c >> 5] >> (c & 31)) & 1u));
}

__device__ __forceinline__ int block_excl_scan_1024(int v, int* wsum, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 16; ++w) {
        const int s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(1024) void nms_candidates_kernel(const float* __restrict__ pred, long long rows, int nc, float conf, int multi,
                                                              ClassMask cm, int use_cm, long long cap,
                                                              unsigned long long* __restrict__ keys, float* __restrict__ cdet,
                                                              int* __restrict__ ncand, unsigned int* __restrict__ seg_begin,
                                                              unsigned int* __restrict__ seg_end) {
    __shared__ int wsum[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int no = 5 + nc;
    const float* pb = pred + (long long)b * rows * no;
    unsigned long long* kb = keys + (long long)b * cap;
    float* db = cdet + (long long)b * cap * 6;
    int base = 0;
    for (long long r0 = 0; r0 < rows; r0 += 1024) {
        const long long r = r0 + tid;
        int cnt = 0, best_j = 0;
        float obj = 0.0f, best = -INFINITY;
        const float* p = pb + r * no;
        if (r < rows) {
            obj = p[4];
            if (obj > conf) {
                if (multi) {
                    for (int j = 0; j < nc; ++j) {
                        const float c = p[5 + j] * obj;
                        if (c > conf && class_ok(cm, use_cm, j)) ++cnt;
                    }
                } else {
                    for (int j = 0; j < nc; ++j) {
                        const float c = p[5 + j] * obj;
                        if (c > best) { best = c; best_j = j; }
                    }
                    if (best > conf && class_ok(cm, use_cm, best_j)) cnt = 1;
                }
            }
        }
        int total;
        int pos = base + block_excl_scan_1024(cnt, wsum, total);
        if (cnt) {
            const float hw = p[2] / 2.0f, hh = p[3] / 2.0f;
            const float x1 = p[0] - hw, y1 = p[1] - hh, x2 = p[0] + hw, y2 = p[1] + hh;
            if (multi) {
                for (int j = 0; j < nc; ++j) {
                    const float c = p[5 + j] * obj;
                    if (c > conf && class_ok(cm, use_cm, j)) {
                        float* d = db + (long long)pos * 6;
                        d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = c; d[5] = (float)j;
                        kb[pos] = ((unsigned long long)__float_as_uint(c) << 32) | (unsigned long long)(0xffffffffu - (unsigned)pos);
                        ++pos;
                    }
                }
            } else {
                float* d = db + (long long)pos * 6;
                d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2; d[4] = best; d[5] = (float)best_j;
                kb[pos] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned long long)(0xffffffffu - (unsigned)pos);
            }
        }
        base += total;
    }
    if (tid == 0) {
        ncand[b] = base;
        seg_begin[b] = (unsigned int)((long long)b * cap);
        seg_end[b] = (unsigned int)((long long)b * cap + base);
    }
}

constexpr int MAX_KEEP = 1024;

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1, float bx2,
                                       float by2, float barea, float thr) {
    // torchvision nms_kernel: w = max(0, min(x2) - max(x1)); ovr = inter / (area_a + area_b - inter); suppress if ovr > thr
    const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1), xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
    const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (aarea + barea - inter);
    return ovr > thr;
}

__global__ __launch_bounds__(256) void nms_greedy_kernel(const unsigned long long* __restrict__ keys_sorted, const float* __restrict__ cdet,
                                                         const int* __restrict__ ncand, long long cap, float iou_thr, float cls_off,
                                                         int max_det, int max_nms, float* __restrict__ det, int* __restrict__ count,
                                                         int* __restrict__ keep_idx) {
    __shared__ float kx1[MAX_KEEP], ky1[MAX_KEEP], kx2[MAX_KEEP], ky2[MAX_KEEP], kar[MAX_KEEP];
    __shared__ float cx1[64], cy1[64], cx2[64], cy2[64], car[64];
    __shared__ int sup[4][64];
    __shared__ int nkept_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_all = ncand[b];
    const bool reordered = n_all > max_nms;          // reference re-indexes x by score rank in that case (:586)
    const int n = reordered ? max_nms : n_all;
    const unsigned long long* kb = keys_sorted + (long long)b * cap;
    const float* db = cdet + (long long)b * cap * 6;
    float* ob = det + (long long)b * max_det * 6;
    int* ib = keep_idx ? keep_idx + (long long)b * max_det : nullptr;
    if (tid == 0) nkept_s = 0;
    __syncthreads();
    int nkept = 0;
    for (int c0 = 0; c0 < n && nkept < max_det; c0 += 64) {
        const int pos = c0 + lane;
        const bool valid = pos < n;
        int ci = 0;
        float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, area = 0.f, sc = 0.f, cl = 0.f, ux1 = 0.f, uy1 = 0.f, ux2 = 0.f, uy2 = 0.f;
        if (valid) {
            ci = (int)(0xffffffffu - (unsigned)(kb[pos] & 0xffffffffull));
            const float* d = db + (long long)ci * 6;
            ux1 = d[0]; uy1 = d[1]; ux2 = d[2]; uy2 = d[3]; sc = d[4]; cl = d[5];
            const float off = cl * cls_off;                     // boxes + cls * max_wh (reference :589-590)
            x1 = ux1 + off; y1 = uy1 + off; x2 = ux2 + off; y2 = uy2 + off;
            area = (x2 - x1) * (y2 - y1);
        }
        // phase A: every wave tests the chunk against a quarter of the kept list
        bool dead = !valid;
        for (int k = wave; k < nkept && !__all(dead); k += 4)
            dead = dead || iou_gt(kx1[k], ky1[k], kx2[k], ky2[k], kar[k], x1, y1, x2, y2, area, iou_thr);
        sup[wave][lane] = dead ? 1 : 0;
        if (wave == 0) { cx1[lane] = x1; cy1[lane] = y1; cx2[lane] = x2; cy2[lane] = y2; car[lane] = area; }
        __syncthreads();
        if (wave == 0) {
            dead = (sup[0][lane] | sup[1][lane] | sup[2][lane] | sup[3][lane]) != 0;
            unsigned long long alive = __ballot(!dead);
            // phase B: resolve the chunk in order; each surviving candidate is kept and suppresses later lanes
            while (alive && nkept < max_det) {
                const int j = __ffsll((long long)alive) - 1;
                alive &= ~(1ull << j);
                const float jx1 = cx1[j], jy1 = cy1[j], jx2 = cx2[j], jy2 = cy2[j], jar = car[j];
                if (lane == j) {
                    kx1[nkept] = x1; ky1[nkept] = y1; kx2[nkept] = x2; ky2[nkept] = y2; kar[nkept] = area;
                    float* o = ob + (long long)nkept * 6;
                    o[0] = ux1; o[1] = uy1; o[2] = ux2; o[3] = uy2; o[4] = sc; o[5] = cl;
                    if (ib) ib[nkept] = reordered ? pos : ci;
                }
                ++nkept;
                const bool hit = lane > j && iou_gt(jx1, jy1, jx2, jy2, jar, x1, y1, x2, y2, area, iou_thr);
                alive &= ~__ballot(hit);
            }
            if (lane == 0) nkept_s = nkept;
        }
        __syncthreads();
        nkept = nkept_s;
    }
    if (tid == 0) count[b] = nkept;
}

struct NmsWs {
    unsigned long long* keys_in; unsigned long long* keys_out; float* cdet; int* ncand; unsigned int* seg_begin; unsigned int* seg_end;
    void* sort_tmp; size_t sort_tmp_bytes; size_t total;
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int nms_layout(int B, long long rows, int nc, int multi, void* base, NmsWs& ws) {
    if (B < 1 || rows < 1 || nc < 1) return fail(ICAF_ERR_ARG, "icaf_nms: bad B/rows/nc");
    const long long cap = rows * ((multi && nc > 1) ? nc : 1);
    if ((long long)B * cap > 0xffffffffLL) return fail(ICAF_ERR_UNSUPPORTED, "icaf_nms: B*rows*nc exceeds 2^32 candidates");
    size_t tmp = 0;
    hipError_t e = rocprim::segmented_radix_sort_keys_desc(nullptr, tmp, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                           (unsigned int)((long long)B * cap), (unsigned int)B, (unsigned int*)nullptr,
                                                           (unsigned int*)nullptr, 0, 64, (hipStream_t) nullptr);
    if (e != hipSuccess) return fail(ICAF_ERR_HIP, "rocprim temp-size query -> %s", hipGetErrorString(e));
    size_t off = 0;
    unsigned char* p = (unsigned char*)base;
    auto take = [&](size_t bytes) { unsigned char* r = p ? p + off : nullptr; off = align_up(off + bytes, 256); return r; };
    ws.keys_in = (unsigned long long*)take((size_t)B * cap * 8);
    ws.keys_out = (unsigned long long*)take((size_t)B * cap * 8);
    ws.cdet = (float*)take((size_t)B * cap * 6 * 4);
    ws.ncand = (int*)take((size_t)B * 4);
    ws.seg_begin = (unsigned int*)take((size_t)B * 4);
    ws.seg_end = (unsigned int*)take((size_t)B * 4);
    ws.sort_tmp = take(tmp);
    ws.sort_tmp_bytes = tmp;
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
    nms_candidates_kernel<<<dim3((unsigned)B), dim3(1024), 0, hs>>>(pred, rows, nc, conf_thres, multi, cm, n_classes > 0 ? 1 : 0, cap,
                                                                     ws.keys_in, ws.cdet, ws.ncand, ws.seg_begin, ws.seg_end);
    ICAF_LAUNCH_CHECK();
    size_t tmp = ws.sort_tmp_bytes;
    ICAF_HIP(rocprim::segmented_radix_sort_keys_desc(ws.sort_tmp, tmp, ws.keys_in, ws.keys_out, (unsigned int)((long long)B * cap),
                                                     (unsigned int)B, ws.seg_begin, ws.seg_end, 0, 64, hs));
    nms_greedy_kernel<<<dim3((unsigned)B), dim3(256), 0, hs>>>(ws.keys_out, ws.cdet, ws.ncand, cap, iou_thres, agnostic ? 0.0f : max_wh,
                                                                max_det, max_nms, det, count, keep_idx);
    ICAF_LAUNCH_CHECK();
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
