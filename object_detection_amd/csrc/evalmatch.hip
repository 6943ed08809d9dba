// Matching of detections to labels for evaluation, whole batches per launch, nothing returns to the host.
//
// Reference: object_detection/yolov5/val.py:71-92 (process_batch -> correct[N, n_iou])
//            object_detection/yolov5/utils/metrics.py:125-163 (ConfusionMatrix.process_batch)
//            object_detection/yolov5/utils/metrics.py:242-264 (box_iou, the IoU both use)
// Both build the list of (label, detection, iou) pairs that pass a test, sort it by IoU descending and reduce
// it with two np.unique(return_index) passes.  Written as a direct rule:
//   1. every detection d keeps its best passing label  l*(d) = argmax_l iou(l, d)   (equal IoUs: the larger l,
//      which is what a stable argsort followed by [::-1] leaves first)
//   2. every label l keeps one detection among { d : l*(d) = l }:
//        val.py (no re-sort after the first unique):   the smallest d
//        ConfusionMatrix (re-sorted by IoU):           the largest IoU, equal IoUs: the larger d
//   3. val.py: correct[d, k] = iou(l, d) >= iouv[k] for the kept pairs;
//      ConfusionMatrix: matrix[class(d), class(l)] += 1 per kept pair, matrix[nc, class(l)] += 1 per label without one,
//      and -- only if the image has at least one kept pair -- matrix[class(d), nc] += 1 per detection without one.
// One workgroup owns one image: labels are staged through LDS in tiles, each thread walks them for its
// detections (rule 1), rule 2 is one 64-bit atomicMax per detection on a per-label key, rule 3 reads it back.
#include "dh_common.h"
#include "../../include/dethead.h"

namespace {

typedef unsigned long long u64;

#define EM_THREADS 256
#define EM_TILE 512

struct em_span {
    int d0, d1, l0, l1;
};

__device__ __forceinline__ em_span image_span(const int* __restrict__ det_first, const int* __restrict__ label_first,
                                              int n_det, int n_labels)
{
    em_span s;
    if (det_first) { s.d0 = det_first[blockIdx.x]; s.d1 = det_first[blockIdx.x + 1]; } else { s.d0 = 0; s.d1 = n_det; }
    if (label_first) { s.l0 = label_first[blockIdx.x]; s.l1 = label_first[blockIdx.x + 1]; } else { s.l0 = 0; s.l1 = n_labels; }
    return s;
}

// CONFUSION = false: val.py rule (pair passes when iou >= thr and the classes are equal; label keeps the smallest d)
// CONFUSION = true:  ConfusionMatrix rule (detections with conf <= conf_thr do not exist; pair passes when iou > thr;
//                    label keeps the largest IoU)
template <bool CONFUSION>
__global__ void __launch_bounds__(EM_THREADS) k_eval_match(const float* __restrict__ det, const float* __restrict__ labels,
                                                           const int* __restrict__ det_first,
                                                           const int* __restrict__ label_first, int n_det, int n_labels,
                                                           float thr, float conf_thr, const float* __restrict__ iouv,
                                                           int n_iou, int nc, unsigned char* __restrict__ correct,
                                                           u64* __restrict__ matrix, int* __restrict__ best_label,
                                                           float* __restrict__ best_iou, u64* __restrict__ label_key)
{
    __shared__ float4 lbox[EM_TILE];
    __shared__ float larea[EM_TILE];
    __shared__ float lcls[EM_TILE];
    __shared__ int s_any;
    const em_span S = image_span(det_first, label_first, n_det, n_labels);
    if (threadIdx.x == 0) s_any = 0;
    const float thr0 = CONFUSION ? thr : iouv[0];

    // rule 1: detections are handled in rounds of EM_THREADS so that every thread sees every label tile
    for (int dbase = S.d0; dbase < S.d1; dbase += EM_THREADS) {
        const int d = dbase + threadIdx.x;
        bool live = d < S.d1;
        float4 db = make_float4(0.f, 0.f, 0.f, 0.f);
        float dcls = 0.f;
        if (live) {
            const float* __restrict__ p = det + (size_t)d * 6;
            db = make_float4(p[0], p[1], p[2], p[3]);
            dcls = p[5];
            if (CONFUSION && !(p[4] > conf_thr)) live = false;                            // metrics.py:135
        }
        const float darea = (db.z - db.x) * (db.w - db.y);
        int bl = -1;
        float bi = 0.f;
        for (int t0 = S.l0; t0 < S.l1; t0 += EM_TILE) {
            __syncthreads();
            for (int k = threadIdx.x; k < EM_TILE && t0 + k < S.l1; k += EM_THREADS) {
                const float* __restrict__ q = labels + (size_t)(t0 + k) * 5;
                lbox[k] = make_float4(q[1], q[2], q[3], q[4]);
                larea[k] = (q[3] - q[1]) * (q[4] - q[2]);
                lcls[k] = q[0];
            }
            __syncthreads();
            const int nt = (S.l1 - t0) < EM_TILE ? (S.l1 - t0) : EM_TILE;
            if (live) {
                for (int k = 0; k < nt; ++k) {
                    const float4 lb = lbox[k];
                    const float inter = fmaxf(fminf(lb.z, db.z) - fmaxf(lb.x, db.x), 0.0f) *
                                        fmaxf(fminf(lb.w, db.w) - fmaxf(lb.y, db.y), 0.0f);       // metrics.py:263
                    const float iou = __fdiv_rn(inter, larea[k] + darea - inter);                 // metrics.py:264
                    const bool pass = CONFUSION ? (iou > thr0) : (iou >= thr0 && lcls[k] == dcls);  // metrics.py:140, val.py:82
                    if (pass && (bl < 0 || iou >= bi)) { bi = iou; bl = t0 + k; }
                }
            }
        }
        if (d < S.d1) {
            best_label[d] = bl;
            best_iou[d] = bi;
            if (bl >= 0) {
                // rule 2 as one atomicMax: passing IoUs are positive floats, whose bit patterns order like the values
                const u64 key = CONFUSION ? (((u64)__float_as_uint(bi) << 32) | (uint32_t)(d - S.d0))
                                          : (u64)(0xFFFFFFFFu - (uint32_t)(d - S.d0));
                atomicMax(label_key + bl, key);
                s_any = 1;
            }
        }
    }
    __threadfence();      // the per-label keys of this workgroup's atomics are read back below by other lanes
    __syncthreads();
    const bool any = s_any != 0;

    // rule 3
    for (int d = S.d0 + threadIdx.x; d < S.d1; d += EM_THREADS) {
        const int bl = best_label[d];
        bool kept = false;
        if (bl >= 0) {
            const u64 key = __hip_atomic_load(label_key + bl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t owner = CONFUSION ? (uint32_t)key : 0xFFFFFFFFu - (uint32_t)key;
            kept = owner == (uint32_t)(d - S.d0);
        }
        if (!CONFUSION) {
            const float iou = best_iou[d];
            for (int k = 0; k < n_iou; ++k) correct[(size_t)d * n_iou + k] = (kept && iou >= iouv[k]) ? 1 : 0;   // val.py:91
        } else {
            const float* __restrict__ p = det + (size_t)d * 6;
            if (!(p[4] > conf_thr)) continue;
            const int dc = (int)p[5];                                                               // .int(): truncation
            if (dc < 0 || dc > nc) continue;
            if (kept) {
                const int gc = (int)labels[(size_t)bl * 5];
                if (gc >= 0 && gc <= nc) atomicAdd(matrix + (size_t)dc * (nc + 1) + gc, 1ull);      // metrics.py:156
            } else if (any) {
                atomicAdd(matrix + (size_t)dc * (nc + 1) + nc, 1ull);                                // metrics.py:163
            }
        }
    }
    if (CONFUSION) {
        for (int l = S.l0 + threadIdx.x; l < S.l1; l += EM_THREADS) {
            const u64 key = __hip_atomic_load(label_key + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (key == 0) {                                                                         // no detection kept this label
                const int gc = (int)labels[(size_t)l * 5];
                if (gc >= 0 && gc <= nc) atomicAdd(matrix + (size_t)nc * (nc + 1) + gc, 1ull);      // metrics.py:158
            }
        }
    }
}

struct em_layout {
    size_t off_key, off_label, off_iou, total, zero_bytes;
};

em_layout make_em_layout(int64_t n_det, int64_t n_labels)
{
    em_layout L;
    size_t o = 0;
    L.off_key = o; o = (o + sizeof(u64) * (size_t)(n_labels > 0 ? n_labels : 1) + 255) / 256 * 256;
    L.zero_bytes = o;
    L.off_label = o; o = (o + sizeof(int) * (size_t)(n_det > 0 ? n_det : 1) + 255) / 256 * 256;
    L.off_iou = o; o = (o + sizeof(float) * (size_t)(n_det > 0 ? n_det : 1) + 255) / 256 * 256;
    L.total = o;
    return L;
}

int launch_match(bool confusion, const float* det, const float* labels, const int* det_first, const int* label_first,
                 int n_img, int64_t n_det, int64_t n_labels, float thr, float conf_thr, const float* iouv, int n_iou, int nc,
                 unsigned char* correct, u64* matrix, void* workspace, size_t workspace_bytes, void* stream)
{
    if (n_img < 0 || n_det < 0 || n_labels < 0 || n_det >= (1ll << 31) || n_labels >= (1ll << 31)) return DH_ERR_ARG;
    if (n_img > 1 && (!det_first || !label_first)) return DH_ERR_ARG;
    if (n_img == 0 || (n_det == 0 && n_labels == 0)) return DH_OK;
    if ((n_det > 0 && !det) || (n_labels > 0 && !labels) || !workspace) return DH_ERR_ARG;
    const em_layout L = make_em_layout(n_det, n_labels);
    if (workspace_bytes < L.total) return DH_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 255u) return DH_ERR_ALIGN;
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (dh_fill_async(ws + L.off_key, 0, L.zero_bytes, st) != DH_OK) return DH_ERR_LAUNCH;   // key 0 = label kept nothing
    if (confusion)
        hipLaunchKernelGGL(k_eval_match<true>, dim3(n_img), dim3(EM_THREADS), 0, st, det, labels, det_first, label_first,
                           (int)n_det, (int)n_labels, thr, conf_thr, iouv, n_iou, nc, correct, matrix,
                           (int*)(ws + L.off_label), (float*)(ws + L.off_iou), (u64*)(ws + L.off_key));
    else
        hipLaunchKernelGGL(k_eval_match<false>, dim3(n_img), dim3(EM_THREADS), 0, st, det, labels, det_first, label_first,
                           (int)n_det, (int)n_labels, thr, conf_thr, iouv, n_iou, nc, correct, matrix,
                           (int*)(ws + L.off_label), (float*)(ws + L.off_iou), (u64*)(ws + L.off_key));
    return dh_launch_status();
}

}  // namespace

extern "C" size_t dh_eval_match_workspace(int64_t n_det, int64_t n_labels)
{
    if (n_det < 0 || n_labels < 0) return 0;
    return make_em_layout(n_det, n_labels).total;
}

extern "C" int dh_eval_match(const float* det, const float* labels, const int* det_first, const int* label_first, int n_img,
                             int64_t n_det, int64_t n_labels, const float* iouv, int n_iou, unsigned char* correct,
                             void* workspace, size_t workspace_bytes, void* stream)
{
    if (n_iou < 1 || !iouv || (n_det > 0 && !correct)) return DH_ERR_ARG;
    return launch_match(false, det, labels, det_first, label_first, n_img, n_det, n_labels, 0.0f, 0.0f, iouv, n_iou, 0,
                        correct, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int dh_confusion_update(const float* det, const float* labels, const int* det_first, const int* label_first,
                                   int n_img, int64_t n_det, int64_t n_labels, int nc, double conf, double iou_thres,
                                   unsigned long long* matrix, void* workspace, size_t workspace_bytes, void* stream)
{
    if (nc < 1 || !matrix) return DH_ERR_ARG;
    return launch_match(true, det, labels, det_first, label_first, n_img, n_det, n_labels, (float)iou_thres, (float)conf,
                        nullptr, 0, nc, nullptr, matrix, workspace, workspace_bytes, stream);
}
