/*
 * TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
 *
 * Scalar CPU restatement of greedy non-maximum suppression with the semantics
 * of torchvision.ops.nms 0.11.1 (CPU kernel), the third-party routine the
 * reference calls at
 *   object_detection/yolov5/utils/general.py:722
 *   object_detection/detectors/centernet/dataset.py:263
 * torchvision is not present in /root/reference nor importable in the build
 * container, so this follows the published algorithm:
 *   - order boxes by score, descending;
 *   - area_k = (x2-x1)*(y2-y1) in fp32;
 *   - walk the order; an unsuppressed box i is kept and every later box j is
 *     suppressed when  inter/(area_i+area_j-inter) > iou_threshold, with
 *     inter = max(0,min(x2)-max(x1)) * max(0,min(y2)-max(y1)) in fp32 and the
 *     final comparison done in double (the threshold is a C double there);
 *   - kept original indices are returned in score-descending order.
 * The library sort in torchvision is unstable, so the order of equal scores is
 * implementation-defined upstream; this file (and the HIP path it checks) pins
 * the stable refinement: equal scores keep ascending original index.
 * Parity status: pinned by hand-computed known-answer tests only (the reference
 * holds no fixtures for this boundary).
 *
 * Build:  gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float score;
    int64_t idx;
} scored_t;

static int by_score_desc_then_index(const void *pa, const void *pb)
{
    const scored_t *a = (const scored_t *)pa, *b = (const scored_t *)pb;
    if (a->score > b->score) return -1;
    if (a->score < b->score) return 1;
    return (a->idx > b->idx) - (a->idx < b->idx);
}

/* boxes: n x 4 (x1,y1,x2,y2) row-major fp32; scores: n fp32.
 * keep: caller buffer of n int64. Returns number kept. */
int64_t oracle_nms(const float *boxes, const float *scores, int64_t n,
                   double iou_threshold, int64_t *keep)
{
    if (n <= 0) return 0;
    scored_t *order = (scored_t *)malloc((size_t)n * sizeof(scored_t));
    float *area = (float *)malloc((size_t)n * sizeof(float));
    unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
    for (int64_t k = 0; k < n; ++k) {
        order[k].score = scores[k];
        order[k].idx = k;
        const float *b = boxes + 4 * k;
        area[k] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    qsort(order, (size_t)n, sizeof(scored_t), by_score_desc_then_index);

    int64_t nkeep = 0;
    for (int64_t p = 0; p < n; ++p) {
        const int64_t i = order[p].idx;
        if (dead[i]) continue;
        keep[nkeep++] = i;
        const float ix1 = boxes[4 * i + 0], iy1 = boxes[4 * i + 1];
        const float ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        const float ia = area[i];
        for (int64_t q = p + 1; q < n; ++q) {
            const int64_t j = order[q].idx;
            if (dead[j]) continue;
            const float *b = boxes + 4 * j;
            const float xx1 = ix1 > b[0] ? ix1 : b[0];
            const float yy1 = iy1 > b[1] ? iy1 : b[1];
            const float xx2 = ix2 < b[2] ? ix2 : b[2];
            const float yy2 = iy2 < b[3] ? iy2 : b[3];
            float w = xx2 - xx1;
            float h = yy2 - yy1;
            if (!(w > 0.0f)) w = 0.0f;
            if (!(h > 0.0f)) h = 0.0f;
            const float inter = w * h;
            const float ovr = inter / (ia + area[j] - inter);
            if ((double)ovr > iou_threshold) dead[j] = 1;
        }
    }
    free(order);
    free(area);
    free(dead);
    return nkeep;
}
