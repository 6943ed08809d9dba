/*
 * libdethead -- C ABI of the MI355X (gfx950) detection-head kernels.
 *
 * The reference (vkosolapov/object_detection) is pure Python and has no FFI of its own; its
 * "interface" for this path is a set of Python callables.  Each entry point below is what the
 * Python drop-in for one of those callables binds (file:line of the callable it serves is cited).
 *
 * Conventions
 *   - plain pointers and sizes only; all tensors are contiguous fp32 unless a stride is passed;
 *   - the caller owns every buffer, including `workspace` (size from the matching *_workspace());
 *     the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t; kernels are enqueued on it and the call returns immediately;
 *   - return value: 0 = ok, DH_ERR_* < 0 otherwise (no kernel is launched on an argument error);
 *   - vector paths need 16-byte aligned pointers (DH_ERR_ALIGN otherwise).
 */
#ifndef DETHEAD_H
#define DETHEAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_MAX_LEVELS 5
#define DH_MAX_ANCHORS 8

/* IoU family selector: 0 IoU, 1 GIoU, 2 DIoU, 3 CIoU (flag precedence of metrics.py:225-239). */

/* Element-wise IoU-family value between n box pairs.
 * Serves bbox_iou of object_detection/yolov5/utils/metrics.py:201 ((4,n) vs (n,4)) and of
 * object_detection/loss.py:118 ([...,4]).  Component k of box i is read at
 * box[k*s_comp + i*s_elem] (strides in floats).  xyxy != 0: corners, else centre/size. */
int dh_bbox_iou_fwd(const float* box1, int64_t s1_comp, int64_t s1_elem, const float* box2, int64_t s2_comp,
                    int64_t s2_elem, float* out, int64_t n, int xyxy, int mode, double eps, void* stream);

/* Gradient of the above w.r.t. box1 (box2 is a constant, as in utils/loss.py:135), autograd
 * conventions of the reference's torch ops (ties of min/max split, alpha of CIoU constant). */
int dh_bbox_iou_bwd(const float* box1, int64_t s1_comp, int64_t s1_elem, const float* box2, int64_t s2_comp,
                    int64_t s2_elem, const float* grad_out, float* grad_box1, int64_t sg_comp, int64_t sg_elem,
                    int64_t n, int xyxy, int mode, double eps, void* stream);

/* Fused masked CIoU loss over dense (B,2,W,H) centre / size planes.
 * Serves IoULossWithMask.forward (object_detection/loss.py:109-115) and, with add_coords != 0,
 * the box branch of compute_losses incl. offset_to_coord
 * (object_detection/detectors/centernet/loss.py:4-13,22-28).
 * out2[0] = 1 - sum(CIoU)/(4*sum(mask)+1e-4), out2[1] = that divisor (kept for the backward). */
size_t dh_centernet_box_loss_workspace(int B, int W, int H);
int dh_centernet_box_loss_fwd(const float* pred_xy, int64_t pred_xy_bstride, const float* pred_wh,
                              int64_t pred_wh_bstride, const float* tgt_xy, int64_t tgt_xy_bstride,
                              const float* tgt_wh, int64_t tgt_wh_bstride, const float* mask, int B, int W, int H,
                              int add_coords, double eps, float* out2, void* workspace, size_t workspace_bytes,
                              void* stream);
int dh_centernet_box_loss_bwd(const float* pred_xy, int64_t pred_xy_bstride, const float* pred_wh,
                              int64_t pred_wh_bstride, const float* tgt_xy, int64_t tgt_xy_bstride,
                              const float* tgt_wh, int64_t tgt_wh_bstride, const float* mask, int B, int W, int H,
                              int add_coords, double eps, const float* fwd_out2, const float* grad_out,
                              float* grad_xy, int64_t grad_xy_bstride, float* grad_wh, int64_t grad_wh_bstride,
                              void* stream);

/* Detect.forward inference decode of all levels into (bs, sum na*ny*nx, no)
 * (object_detection/yolov5/models/yolo.py:56-70,72-81).  level_ptrs/ny/nx/strides are host
 * arrays of nl entries; anchor_px is host (nl,na,2) = anchors[l]*stride[l] as _make_grid forms it;
 * level l input is (bs,na,ny,nx,no) contiguous on the device. */
int dh_yolo_decode(const float* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                   const float* anchor_px, const float* strides, float* out, void* stream);

/* Batched non_max_suppression (object_detection/yolov5/utils/general.py:647-738) over
 * pred (bs, A, 5+nc) = [x,y,w,h,obj,cls...].  Output rows are [x1,y1,x2,y2,conf,cls] in the
 * reference's order (score descending; equal scores in candidate order), out is (bs,max_det,6),
 * out_count (bs) the rows valid per image, cand_count (bs) the candidates found per image:
 * cand_count[b] > cap means the per-image candidate capacity overflowed and the call must be
 * repeated with a larger cap (only possible with multi_label; cap = A always suffices otherwise).
 * class_keep: nc bytes on the device (1 = keep class) or NULL (general.py:705-706).
 * max_nms = 30000 and max_wh = 7680 in the reference (general.py:663-664).  All of out,
 * out_count, cand_count, workspace are device memory; workspace must be 256-byte aligned. */
size_t dh_nms_workspace(int bs, int cap, int max_nms);
int dh_nms_batched(const float* pred, int bs, int A, int nc, double conf_thres, double iou_thres, int multi_label,
                   int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
                   float* out, int* out_count, int* cand_count, void* workspace, size_t workspace_bytes, void* stream);

/* torchvision.ops.nms(boxes[n,4] xyxy, scores[n], iou_threshold) as called at
 * object_detection/yolov5/utils/general.py:722 and object_detection/detectors/centernet/dataset.py:263:
 * keep_index (n int64, device) receives the kept original indices in score-descending order
 * (equal scores: ascending index), keep_count (1 int, device) how many. */
size_t dh_nms_plain_workspace(int64_t n);
int dh_nms_plain(const float* boxes, const float* scores, int64_t n, double iou_thres, int64_t* keep_index,
                 int* keep_count, void* workspace, size_t workspace_bytes, void* stream);

/* xywh2xyxy (mode 0) / xyxy2xywh (mode 1) on the first four columns of n rows
 * (object_detection/yolov5/utils/general.py:543-560); strides in floats; in == out is allowed. */
int dh_box_convert(const float* in, int64_t in_row_stride, float* out, int64_t out_row_stride, int64_t n, int mode,
                   void* stream);

/* In-place scale_coords (do_scale != 0: (v - pad)/gain, then clip) or clip_coords (do_scale == 0)
 * on the first four columns of n rows (general.py:619-644).  max_x/max_y = image width/height. */
int dh_scale_clip_coords(float* boxes, int64_t row_stride, int64_t n, int do_scale, double pad_x, double pad_y,
                         double gain, double max_x, double max_y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DETHEAD_H */
