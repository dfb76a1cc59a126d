/* CPU oracle for the greedy NMS core — TEST INFRASTRUCTURE ONLY (see oracle/icaf_oracle.py header).
 *
 * Restates the semantics of torchvision.ops.nms, which the reference calls at utils/general.py:591 and which is a
 * third-party, un-vendored dependency (requirements.txt:11, torchvision>=0.8.1): boxes are visited in the given
 * order (descending score, stable), a box is kept unless an already-kept box overlaps it with IoU > thr, where
 * IoU = inter / (area_i + area_j - inter) in fp32.  Build flags must include -ffp-contract=off so that no fused
 * multiply-add changes the fp32 rounding of w*h or of the denominator.  "Parity unpinned": no golden vectors for
 * this core exist in the reference.
 */
#include <stdlib.h>

int nms_greedy_f32(const float *boxes, const int *order, int n, float thr, int *keep)
{
    unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
    float *area = (float *)malloc(sizeof(float) * (size_t)n);
    int kept = 0;
    if (!dead || !area) { free(dead); free(area); return -1; }
    for (int i = 0; i < n; ++i)
        area[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    for (int p = 0; p < n; ++p) {
        const int a = order[p];
        if (dead[a]) continue;
        keep[kept++] = a;
        const float ax1 = boxes[4 * a], ay1 = boxes[4 * a + 1], ax2 = boxes[4 * a + 2], ay2 = boxes[4 * a + 3];
        const float aa = area[a];
        for (int q = p + 1; q < n; ++q) {
            const int b = order[q];
            if (dead[b]) continue;
            const float xx1 = ax1 > boxes[4 * b] ? ax1 : boxes[4 * b];
            const float yy1 = ay1 > boxes[4 * b + 1] ? ay1 : boxes[4 * b + 1];
            const float xx2 = ax2 < boxes[4 * b + 2] ? ax2 : boxes[4 * b + 2];
            const float yy2 = ay2 < boxes[4 * b + 3] ? ay2 : boxes[4 * b + 3];
            const float w = xx2 - xx1 > 0.0f ? xx2 - xx1 : 0.0f;
            const float h = yy2 - yy1 > 0.0f ? yy2 - yy1 : 0.0f;
            const float inter = w * h;
            const float ovr = inter / (aa + area[b] - inter);
            if (ovr > thr) dead[b] = 1;
        }
    }
    free(dead);
    free(area);
    return kept;
}
