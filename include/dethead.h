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
 *     the library never allocates, frees or synchronises (dh_nms_plain_noalloc brings its own fixed workspace);
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
 * out2[0] = 1 - sum(CIoU)/(4*sum(mask)+1e-4), out2[1] = that divisor (kept for the backward).
 * Layout: W*H a multiple of 4, 16-byte aligned pointers; a *_bstride is the distance in floats between two images of
 * that tensor (a multiple of 4, at least 2*W*H), so the planes may be slices of a wider head output. */
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
/* The two calls above as one sequence of two launches, for a training step that takes the gradients directly
 * (`losses["box"] = criterion["box"](...)` followed by backward(), object_detection/detectors/centernet/loss.py:24-28 and
 * object_detection/train_loop.py:296-311): the gradient pass reduces the forward's partial sums itself, so no reduction
 * launch sits between them.  grad_out: upstream gradient of the loss (1 float, device) or NULL for 1; out2, gradients and
 * workspace as above.  Values and gradients equal the separate calls bit for bit. */
int dh_centernet_box_loss_fwd_bwd(const float* pred_xy, int64_t pred_xy_bstride, const float* pred_wh,
                                  int64_t pred_wh_bstride, const float* tgt_xy, int64_t tgt_xy_bstride,
                                  const float* tgt_wh, int64_t tgt_wh_bstride, const float* mask, int B, int W, int H,
                                  int add_coords, double eps, const float* grad_out, float* out2, float* grad_xy,
                                  int64_t grad_xy_bstride, float* grad_wh, int64_t grad_wh_bstride, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* Detect.forward inference decode of all levels into (bs, sum na*ny*nx, no)
 * (object_detection/yolov5/models/yolo.py:56-70,72-81).  level_ptrs/ny/nx/strides are host
 * arrays of nl entries; anchor_px is host (nl,na,2) = anchors[l]*stride[l] as _make_grid forms it;
 * level l input is (bs,na,ny,nx,no) contiguous on the device.  obj_plane (optional, may be NULL):
 * (bs, A) floats that receive a compact copy of the decoded objectness column out[:, :, 4], which
 * dh_nms_batched can read instead of the 340-byte-strided column.  obj_logit_plane (optional, needs
 * obj_plane): (bs, A) floats that receive the raw objectness logits (input channel 4, same row order),
 * which dh_yolo_loss_fwd can read instead of the strided input column when the loss is evaluated on the
 * same maps (as val.py does: object_detection/yolov5/val.py:194-198). */
int dh_yolo_decode(const float* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                   const float* anchor_px, const float* strides, float* out, float* obj_plane, float* obj_logit_plane,
                   void* stream);
/* The same for a half-precision head (model.half() of object_detection/yolov5/val.py:115,129): the level maps hold fp16
 * elements, which are widened on load; the arithmetic and every output stay fp32, so the result equals
 * dh_yolo_decode on the maps converted to fp32, without that conversion pass (170 instead of 340 B read per anchor). */
int dh_yolo_decode_f16(const void* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                       const float* anchor_px, const float* strides, float* out, float* obj_plane, float* obj_logit_plane,
                       void* stream);

/* Detect.forward's inference branch as a half-precision model computes it (model.half(),
 * object_detection/yolov5/val.py:129-130; models/yolo.py:56-70 with fp16 maps and float grid / anchor_grid, yolo.py:78-80):
 * fp16 roundings after the sigmoid, after -0.5 and after **2, fp32 for + grid, * stride and * anchor_grid, and one more
 * rounding into the fp16 result.  inplace != 0 (yolo.py:59-61): out is (bs, A, no) fp16.  inplace == 0 (the torch.cat form,
 * yolo.py:63-65): out is (bs, A, no) fp32, xy and wh are the fp32 values before that last rounding and the rest are the
 * fp16 sigmoids widened, which is what the cat of float and half tensors yields.  anchor_px must be the model's anchor_grid
 * values, (anchors.half() * stride).float().  obj_plane as in dh_yolo_decode (the fp16 objectness values as fp32). */
int dh_yolo_decode_half(const void* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                        const float* anchor_px, const float* strides, int inplace, void* out, float* obj_plane, void* stream);

/* Detect.forward from the outputs of its 1x1 convolutions onward (object_detection/yolov5/models/yolo.py:53-70):
 * conv_ptrs[l] is the convolution output of level l, (bs, na*no, ny, nx) contiguous; map_ptrs[l] receives
 * x[i].view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous(), i.e. the (bs, na, ny, nx, no) head map (yolo.py:54);
 * out / obj_plane / obj_logit_plane receive what dh_yolo_decode writes for those maps (yolo.py:56-70), bit for bit.
 * One launch reads the convolution outputs once.  out == NULL: layout change only (the training branch; anchor_px and
 * strides may then be NULL and the planes must be).  map_ptrs == NULL: decoded tensor only.  Needs no <= 256. */
int dh_yolo_head_maps(const float* const* conv_ptrs, float* const* map_ptrs, const int* ny, const int* nx, int nl, int bs,
                      int na, int no, const float* anchor_px, const float* strides, float* out, float* obj_plane,
                      float* obj_logit_plane, void* stream);
/* The same for a half-precision head (model.half() of object_detection/yolov5/val.py:115,129): convolution outputs and
 * maps hold fp16 elements; out and the planes stay fp32 and equal dh_yolo_decode_f16 on those maps. */
int dh_yolo_head_maps_f16(const void* const* conv_ptrs, void* const* map_ptrs, const int* ny, const int* nx, int nl, int bs,
                          int na, int no, const float* anchor_px, const float* strides, float* out, float* obj_plane,
                          float* obj_logit_plane, void* stream);

/* The same for a half-precision model (model.half(), object_detection/yolov5/val.py:129-130): fp16 convolution outputs in,
 * fp16 maps (optional) and the fp16 decoded tensor of dh_yolo_decode_half (in-place form, yolo.py:59-61, same roundings)
 * out, in one launch.  anchor_px as for dh_yolo_decode_half; obj_plane (optional): channel 4 of the decoded tensor,
 * widened, (bs, A). */
int dh_yolo_head_maps_half(const void* const* conv_ptrs, void* const* map_ptrs, const int* ny, const int* nx, int nl, int bs,
                           int na, int no, const float* anchor_px, const float* strides, void* out_f16, float* obj_plane,
                           void* stream);
/* Backward of that layout change (what autograd does for the permute + contiguous of yolo.py:54, followed by the
 * contiguous copy the convolution's backward asks for): grad_map_ptrs[l] (bs, na, ny, nx, no) contiguous ->
 * grad_conv_ptrs[l] (bs, na*no, ny, nx) contiguous. */
int dh_yolo_head_maps_bwd(const float* const* grad_map_ptrs, float* const* grad_conv_ptrs, const int* ny, const int* nx,
                          int nl, int bs, int na, int no, void* stream);
/* The same on fp16 gradients (training under autocast keeps the head in half precision). */
int dh_yolo_head_maps_bwd_f16(const void* const* grad_map_ptrs, void* const* grad_conv_ptrs, const int* ny, const int* nx,
                              int nl, int bs, int na, int no, void* stream);

/* Batched non_max_suppression (object_detection/yolov5/utils/general.py:647-738) over
 * pred (bs, A, 5+nc) = [x,y,w,h,obj,cls...].  Output rows are [x1,y1,x2,y2,conf,cls] in the
 * reference's order (score descending; equal scores in candidate order).  The rows of all images are
 * packed back to back in out (room for bs*max_det rows of 6 floats); out_span (bs pairs of ints) gives
 * each image's {first row, row count} (images are packed in completion order, not index order).
 * cand_count (bs) is the number of candidates found per image: cand_count[b] > cap means the per-image
 * candidate capacity overflowed and the call must be repeated with a larger cap (only possible with
 * multi_label; cap = A always suffices otherwise).
 * class_keep: nc bytes on the device (1 = keep class) or NULL (general.py:705-706).
 * obj_plane: NULL, or (bs, A) floats holding exactly pred[:, :, 4] (see dh_yolo_decode).
 * max_nms = 30000 and max_wh = 7680 in the reference (general.py:663-664).
 * k_select: 0 = every candidate is sorted and goes through NMS.  > 0 = only the k_select best-scoring candidates
 * of an image (rounded up so that at least 8*max_det fit; ignored when that exceeds the 16384-key LDS sorter) do:
 * the output is the first max_det kept rows in score order and a row's fate depends only on rows ranked above
 * it, so the result is exact whenever that prefix yields max_det rows.  needs_full[b] = 1 marks the images where it
 * did not (their output rows are incomplete): repeat the call with k_select = 0.  With k_select = 0 needs_full is all 0.
 * k_select = -1: everything after the candidate search runs in one workgroup per image (three launches in all, for the
 * usual few dozen candidates per image); needs max_det <= 1024, and an image with more than 1024 candidates sets
 * needs_full[b] = 1 and gets no rows: repeat the call with another k_select.
 * k_select = -2: the best-scoring candidates of an image (the best score bins that together hold at least 3584 keys,
 * at most 4096) go through sort, box fetch, per-class greedy NMS and collection in one launch, several workgroups per
 * image, each taking a share of the classes (three launches in all, for the thousands of candidates of validation
 * thresholds).  needs_full[b] = 1 as for k_select > 0; needs_full[b] = 2 marks an image left to the separate launches
 * (too many equal scores for the cut, or a box outside the window in which the class offset alone rules out overlap
 * across classes): repeat with k_select = 3840.  Where its tables do not fit (agnostic, nc > 512, max_det > 448,
 * A > 2^23) the call runs as k_select = 3840.
 * All of out, out_span, cand_count, needs_full (bs ints), workspace are device memory; workspace must be 256-byte aligned. */
size_t dh_nms_workspace(int bs, int nc, int cap, int max_nms);
int dh_nms_batched(const float* pred, const float* obj_plane, int bs, int A, int nc, double conf_thres, double iou_thres,
                   int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms,
                   double max_wh, int cap, int k_select, float* out, int* out_span, int* cand_count, int* needs_full,
                   void* workspace, size_t workspace_bytes, void* stream);

/* non_max_suppression on a decoded tensor in fp16, as the reference runs it behind a half-precision model
 * (object_detection/yolov5/val.py:129-130,203): conf = obj * cls rounded to fp16 (utils/general.py:691), corners formed in
 * fp16 (general.py:694,553-560), fp32 from the torch.cat with the float class column on (general.py:699,702).
 * dh_nms_prepare_f16 does the fp16 part: pred_f16 (bs, A, 5+nc) fp16 -> out (bs, A, 5+nc) fp32 holding x1, y1, x2, y2,
 * the outcome of obj > conf_thres (1 or 0) and the conf values.  image_fp32 (optional, bs bytes): images flagged there are
 * computed in fp32 from the widened values, which is what the cat with the float label rows does to an image that has
 * apriori labels (general.py:678-684).  obj_plane (optional): (bs, A) floats that receive a compact copy of channel 4 of
 * out, for dh_nms_batched_prepared to read.  dh_nms_batched_prepared is dh_nms_batched on such a tensor (arguments as
 * there; the plane must be the one written with `pred`, or NULL). */
int dh_nms_prepare_f16(const void* pred_f16, int bs, int A, int nc, double conf_thres, const unsigned char* image_fp32,
                       float* out, float* obj_plane, void* stream);
int dh_nms_batched_prepared(const float* pred, const float* obj_plane, int bs, int A, int nc, double conf_thres, double iou_thres, int multi_label,
                            int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
                            int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
                            size_t workspace_bytes, void* stream);

/* dh_nms_batched on the fp16 decoded tensor of a half-precision model (non_max_suppression behind model.half(),
 * object_detection/yolov5/val.py:129-130,203) without the fp32 copy of dh_nms_prepare_f16: conf = obj * cls rounded to
 * fp16 (utils/general.py:691) and the corners formed in fp16 (general.py:694,553-560) as the rows are looked at; same rows
 * as dh_nms_prepare_f16 + dh_nms_batched_prepared, bit for bit.  obj_plane: optional fp32 copy of channel 4, (bs, A).
 * Images with apriori labels (general.py:678-684) are widened before that arithmetic and need the prepared form. */
int dh_nms_batched_f16(const void* pred_f16, const float* obj_plane, int bs, int A, int nc, double conf_thres, double iou_thres,
                       int multi_label, int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh,
                       int cap, int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Fused inference path: the NMS of dh_nms_batched fed directly by the raw head maps of dh_yolo_decode (same level
 * description), i.e. non_max_suppression(Detect(x)[0]) of object_detection/yolov5/val.py:194,203 and detect.py:126,132
 * without materialising the decoded (bs, A, no) tensor: candidate rows are decoded on the fly with the arithmetic of
 * dh_yolo_decode, so the output is bit-identical to the two-call sequence.  A = sum_l na*ny*nx; cap as in dh_nms_batched. */
int dh_yolo_decode_nms(const float* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                       const float* anchor_px, const float* strides, double conf_thres, double iou_thres, int multi_label,
                       int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
                       int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
                       size_t workspace_bytes, void* stream);
/* The same on fp16 level maps (see dh_yolo_decode_f16): equal to dh_yolo_decode_nms on the maps converted to fp32. */
int dh_yolo_decode_nms_f16(const void* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                           const float* anchor_px, const float* strides, double conf_thres, double iou_thres, int multi_label,
                           int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
                           int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
                           size_t workspace_bytes, void* stream);

/* The same on fp16 level maps with the arithmetic of a half-precision model (model.half(),
 * object_detection/yolov5/val.py:129-130): what dh_yolo_decode_half (in-place form) followed by dh_nms_prepare_f16 and
 * dh_nms_batched_prepared produce, bit for bit, without the fp16 decoded tensor or its fp32 copy -- the sigmoid rounded to
 * fp16 (yolo.py:58), the box with the roundings of yolo.py:59-61, conf = obj * cls rounded to fp16
 * (utils/general.py:691), the corners formed in fp16 (general.py:694,553-560), fp32 from there on.  anchor_px as for
 * dh_yolo_decode_half: (anchors.half() * stride).float(). */
int dh_yolo_decode_nms_half(const void* const* level_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                            const float* anchor_px, const float* strides, double conf_thres, double iou_thres, int multi_label,
                            int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
                            int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
                            size_t workspace_bytes, void* stream);

/* The same fed by the outputs of the head's 1x1 convolutions, conv_ptrs[l] = (bs, na*no, ny, nx) contiguous, i.e. the
 * tensors Detect.forward holds before the layout change of object_detection/yolov5/models/yolo.py:54: neither the head
 * maps nor the decoded tensor are written or read.  The candidate search reads the objectness planes (contiguous in this
 * layout) and the class planes of the waves that hold a candidate; detections are bit-identical to
 * dh_yolo_head_maps + dh_nms_batched.  For callers that only want detections (detect.py:126-132, val.py:194-203 without
 * compute_loss). */
int dh_yolo_conv_nms(const float* const* conv_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                     const float* anchor_px, const float* strides, double conf_thres, double iou_thres, int multi_label,
                     int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
                     int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
                     size_t workspace_bytes, void* stream);
int dh_yolo_conv_nms_f16(const void* const* conv_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                         const float* anchor_px, const float* strides, double conf_thres, double iou_thres, int multi_label,
                         int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
                         int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
                         size_t workspace_bytes, void* stream);

/* dh_yolo_conv_nms_f16 with the arithmetic of a half-precision model, as dh_yolo_decode_nms_half: the detections of
 * dh_yolo_head_maps_f16 + dh_yolo_decode_half (in-place form) + dh_nms_prepare_f16 + dh_nms_batched_prepared, bit for bit. */
int dh_yolo_conv_nms_half(const void* const* conv_ptrs, const int* ny, const int* nx, int nl, int bs, int na, int no,
                          const float* anchor_px, const float* strides, double conf_thres, double iou_thres, int multi_label,
                          int agnostic, const unsigned char* class_keep, int max_det, int max_nms, double max_wh, int cap,
                          int k_select, float* out, int* out_span, int* cand_count, int* needs_full, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Merge-NMS, the step of non_max_suppression that the reference keeps behind its local switch `merge`
 * (object_detection/yolov5/utils/general.py:668,725-731), applied to the result of the dh_nms_batched /
 * dh_yolo_decode_nms call that has just run with k_select = 0 on the same workspace and stream (same bs, nc, cap, max_nms,
 * max_det): in every image with 1 < n < 3000 candidates each output row's box becomes the score-weighted mean of all
 * candidate boxes whose IoU with it (object_detection/yolov5/utils/metrics.py:242-264 on the class-shifted boxes) exceeds
 * iou_thres; with redundant != 0 (general.py:666) rows that matched nothing but themselves are dropped, the image's rows
 * close up in order and out_span[b] gets the new count.  max_det <= 4096. */
int dh_nms_merge(float* out, int* out_span, int bs, int nc, int cap, int max_nms, int max_det, double iou_thres, int agnostic,
                 double max_wh, int redundant, void* workspace, size_t workspace_bytes, void* stream);

/* torchvision.ops.nms(boxes[n,4] xyxy, scores[n], iou_threshold) as called at
 * object_detection/yolov5/utils/general.py:722 and object_detection/detectors/centernet/dataset.py:263:
 * keep_index (n int64, device) receives the kept original indices in score-descending order
 * (equal scores: ascending index), keep_count (1 int, device) how many. */
size_t dh_nms_plain_workspace(int64_t n);
int dh_nms_plain(const float* boxes, const float* scores, int64_t n, double iou_thres, int64_t* keep_index,
                 int* keep_count, void* workspace, size_t workspace_bytes, void* stream);

/* dh_nms_plain for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): no counterpart in the reference.  Same inputs, outputs and results as dh_nms_plain
 * with no workspace argument: the kernels work on a fixed 2.25 MiB device array that is part of the library (every GPU
 * the library runs on holds its own copy until the process ends).  n <= 32768, larger inputs return DH_ERR_TOO_LARGE (-5).
 * The call allocates and frees nothing.  All calls on one GPU share that array: calls that overlap on one GPU (two
 * streams, two threads) race; serialise them.  The first call on a device looks the array's address up
 * (hipGetSymbolAddress), so make one call outside graph capture before capturing.  `stream` must belong to the device
 * that is current when the call is made (every GPU has its own copy of the array): a stream of another device returns
 * DH_ERR_ARG. */
int dh_nms_plain_noalloc(const float* boxes, const float* scores, int64_t n, double iou_thres, int64_t* keep_index,
                         int* keep_count, void* stream);

/* xywh2xyxy (mode 0) / xyxy2xywh (mode 1) on the first four columns of n rows
 * (object_detection/yolov5/utils/general.py:543-560); strides in floats; in == out is allowed. */
int dh_box_convert(const float* in, int64_t in_row_stride, float* out, int64_t out_row_stride, int64_t n, int mode,
                   void* stream);

/* In-place scale_coords (do_scale != 0: (v - pad)/gain, then clip) or clip_coords (do_scale == 0)
 * on the first four columns of n rows (general.py:619-644).  max_x/max_y = image width/height. */
int dh_scale_clip_coords(float* boxes, int64_t row_stride, int64_t n, int do_scale, double pad_x, double pad_y,
                         double gain, double max_x, double max_y, void* stream);

/* In-place de-scaling of augmented-inference predictions (object_detection/yolov5/models/yolo.py:155-163, the
 * in-place branch): columns 0..3 of n rows are divided by scale, then flips == 2 sets y = img_h - y and flips == 3
 * x = img_w - x.  row_stride in floats (>= 4). */
int dh_descale_pred(float* rows, int64_t row_stride, int64_t n, double scale, int flips, double img_h, double img_w,
                    void* stream);

/* ComputeLoss.build_targets (object_detection/yolov5/utils/loss.py:169-222) for all nl levels.
 * targets (nt,6) [image,class,x,y,w,h] and anchors (nl,na,2, grid units) on the device; ny/nx host
 * arrays.  Per level l, rows [0,counts[l]) of:  idx[l][0..3][cap] = image, anchor, gj, gi planes
 * (int64), tbox[l][cap][4], anch[l][cap][2], tcls[l][cap] (int64), in the reference's row order.
 * cap >= 5*na*nt rows are reserved per level. */
size_t dh_build_targets_workspace(int nl, int na, int64_t nt);
int dh_build_targets(const float* targets, int64_t nt, const float* anchors, int nl, int na, const int* ny, const int* nx,
                     double anchor_t, int64_t cap, int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Hyper-parameters of ComputeLoss (loss.py:93-115): gains box/obj/cls, BCE pos_weights, focal gamma
 * (0 = plain BCE), smoothed class targets cp/cn (loss.py:13-15,103), iou ratio gr (loss.py:113), the
 * per-level objectness balance (loss.py:111) and sort_obj_iou (loss.py:94,140-142; non-zero: where several
 * assigned rows share a cell, the objectness target comes from the row with the largest IoU instead of from
 * the last row). */
typedef struct dh_yolo_hyp {
    float box, obj, cls, cls_pw, obj_pw, fl_gamma, cp, cn, gr;
    float balance[DH_MAX_LEVELS];
    int sort_obj_iou;
} dh_yolo_hyp;

/* ComputeLoss.__call__ (loss.py:117-167) given the assignment produced by dh_build_targets.
 * p / grad_p: host arrays of nl device pointers to (bs,na,ny,nx,5+nc) maps.
 * out (device, 6+nl floats): [0] loss, [1..3] lbox, lobj, lcls after gains, [4] assigned rows over all
 * levels, [5..5+nl) mean objectness BCE per level (what autobalance reads, loss.py:158), [5+nl] rows
 * ignored because their image/anchor index was outside the maps.  The forward leaves in `workspace` what the backward needs (per-row IoU,
 * the per-cell lists of assigned rows, the stored objectness derivative), so the same workspace must be passed to
 * dh_yolo_loss_bwd unmodified.
 * The backward writes every element of every grad_p[l] (no pre-zeroing needed); grad_loss is the
 * upstream gradient of out[0] (1 float, device).  The rows assigned to one cell are summed in row order by one wave and
 * stored, so the gradient is the same bit for bit on every run (index_put(accumulate=True) applied sequentially, as the
 * reference's backward does on the CPU); only a cell shared by more than 64 rows falls back to floating-point atomics.
 * obj_logits: NULL, or the (bs, sum_l na*ny*nx) plane dh_yolo_decode wrote for exactly these maps
 * (obj_logits[b][first_l + (a*ny+gj)*nx+gi] == p[l][b][a][gj][gi][4] bit for bit); the dense objectness
 * pass then reads it instead of the strided column. */
size_t dh_yolo_loss_workspace(const int* ny, const int* nx, int nl, int bs, int na, int nc, int64_t cap);
/* Byte offset inside that workspace of `double sums[nl][5]`, which every forward (and fused forward+backward) leaves there
 * beside `out`: per level [sum of (1 - iou) over the assigned rows, sum of the class BCE terms, sum of the objectness BCE
 * terms, assigned rows, cells] -- the numerators and divisors of the means of loss.py:136,149,155 before the division.
 * Ranks of a data-parallel job (object_detection/yolov5/train.py:254-255) add these up (one all-reduce of 5*nl doubles) to
 * form the means of the whole batch; the reference itself keeps per-rank means (train.py:331-332). */
size_t dh_yolo_loss_sums_offset(const int* ny, const int* nx, int nl, int bs, int na, int nc, int64_t cap);
int dh_yolo_loss_fwd(const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                     const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls, const int* counts,
                     int64_t cap, const dh_yolo_hyp* hyp, const float* obj_logits, float* out, void* workspace,
                     size_t workspace_bytes, void* stream);
int dh_yolo_loss_bwd(const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                     const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls, const int* counts,
                     int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss, float* const* grad_p,
                     const void* workspace, size_t workspace_bytes, void* stream);

/* dh_build_targets + dh_yolo_loss_fwd + dh_yolo_loss_bwd as one sequence of five launches (a training step's
 * `loss = compute_loss(pred, targets); loss.backward()`, object_detection/yolov5/train.py:329,337, when the caller
 * takes the gradients directly): the fill of the per-cell map rides in spare workgroups of the assignment launch and the
 * forward's final reduction in workgroup 0 of the dense gradient launch, so neither costs a launch of its own.
 * Arguments as in the three entries; idx/tbox/anch/tcls/counts are outputs (the assignment), assign_workspace is
 * dh_build_targets_workspace bytes, workspace dh_yolo_loss_workspace bytes.  out and grad_p as in the separate calls. */
int dh_yolo_loss_fwd_bwd(const float* targets, int64_t nt, const float* anchors, double anchor_t, const float* const* p,
                         const int* ny, const int* nx, int nl, int bs, int na, int nc, int64_t* idx, float* tbox, float* anch,
                         int64_t* tcls, int* counts, int64_t cap, const dh_yolo_hyp* hyp, const float* obj_logits,
                         const float* grad_loss, float* const* grad_p, float* out, void* assign_workspace,
                         size_t assign_workspace_bytes, void* workspace, size_t workspace_bytes, void* stream);
/* The three loss entries on the convolution layout: conv[l] is the output of the head's 1x1 convolution of level l as it
 * is, (bs, na*(5+nc), ny, nx) contiguous -- what Detect.forward holds before the permuted copy of
 * object_detection/yolov5/models/yolo.py:54 -- and grad_conv[l] receives d loss / d conv[l] in the same layout, so a training
 * step (object_detection/yolov5/train.py:329-337) needs neither that copy nor its backward.  Values as in the entries
 * above on the permuted maps: the assigned rows and the gradient element for element, the objectness sum up to the order
 * of its partial sums.  There is no compact logit plane to pass: the objectness logits are contiguous in this layout. */
int dh_yolo_loss_fwd_conv(const float* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                          const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls, const int* counts,
                          int64_t cap, const dh_yolo_hyp* hyp, float* out, void* workspace, size_t workspace_bytes, void* stream);
int dh_yolo_loss_bwd_conv(const float* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                          const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls, const int* counts,
                          int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss, float* const* grad_conv,
                          const void* workspace, size_t workspace_bytes, void* stream);
int dh_yolo_loss_fwd_bwd_conv(const float* targets, int64_t nt, const float* anchors, double anchor_t, const float* const* conv,
                              const int* ny, const int* nx, int nl, int bs, int na, int nc, int64_t* idx, float* tbox,
                              float* anch, int64_t* tcls, int* counts, int64_t cap, const dh_yolo_hyp* hyp,
                              const float* grad_loss, float* const* grad_conv, float* out, void* assign_workspace,
                              size_t assign_workspace_bytes, void* workspace, size_t workspace_bytes, void* stream);
/* The same three for a head that runs in half precision (training under autocast, object_detection/yolov5/train.py:328-330:
 * the convolution outputs are fp16): conv[l] holds fp16 elements, which are widened on load; the arithmetic and out stay
 * fp32, so the forward equals the fp32 entry on the outputs converted with .float().
 * grad_f16 == 0: grad_conv[l] are fp32 tensors and equal the fp32 entry's gradient.
 * grad_f16 != 0: grad_conv[l] are fp16 tensors, the type autograd hands back to an fp16 convolution: an element that
 * receives one term is that term rounded to fp16; a cell assigned to several target rows gets the fp32 sum of their terms
 * (in row order) rounded once -- beyond 64 rows on one cell, the hardware's packed fp16 atomic add, one rounding per term. */
int dh_yolo_loss_fwd_conv_f16(const void* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                              const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls, const int* counts,
                              int64_t cap, const dh_yolo_hyp* hyp, float* out, void* workspace, size_t workspace_bytes,
                              void* stream);
int dh_yolo_loss_bwd_conv_f16(const void* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                              const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls, const int* counts,
                              int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss, int grad_f16, void* const* grad_conv,
                              const void* workspace, size_t workspace_bytes, void* stream);
int dh_yolo_loss_fwd_bwd_conv_f16(const float* targets, int64_t nt, const float* anchors, double anchor_t,
                                  const void* const* conv, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                                  int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts, int64_t cap,
                                  const dh_yolo_hyp* hyp, const float* grad_loss, int grad_f16, void* const* grad_conv,
                                  float* out, void* assign_workspace, size_t assign_workspace_bytes, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* dh_yolo_loss_fwd_bwd in two halves, for callers that can start the first one early.  dh_yolo_loss_positives: target
 * assignment (loss.py:169-222) and the forward of the assigned rows (loss.py:126-150) -- they read only the targets and
 * the raw maps, so they may run on another stream beside whatever produces or decodes the maps.  dh_yolo_loss_dense: the
 * dense objectness forward (loss.py:152-156), the whole backward and the reduction into out.  Same arguments, buffers and
 * workspaces as dh_yolo_loss_fwd_bwd; the caller orders the second call after the first (same stream, or an event). */
int dh_yolo_loss_positives(const float* targets, int64_t nt, const float* anchors, double anchor_t, const float* const* p,
                           const int* ny, const int* nx, int nl, int bs, int na, int nc, int64_t* idx, float* tbox,
                           float* anch, int64_t* tcls, int* counts, int64_t cap, const dh_yolo_hyp* hyp,
                           void* assign_workspace, size_t assign_workspace_bytes, void* workspace, size_t workspace_bytes,
                           void* stream);
int dh_yolo_loss_dense(const float* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc, const int64_t* idx,
                       const float* tbox, const float* anch, const int64_t* tcls, const int* counts, int64_t cap,
                       const dh_yolo_hyp* hyp, const float* obj_logits, const float* grad_loss, float* const* grad_p,
                       float* out, void* workspace, size_t workspace_bytes, void* stream);
/* dh_yolo_loss_bwd (layout 0), dh_yolo_loss_bwd_conv (1) or dh_yolo_loss_bwd_conv_f16 (2: fp32 gradient, 3: fp16 gradient)
 * with a factor per level on top of grad_loss: level_scale (device, 2*nl floats) holds for level l the factor on its box and
 * class terms and the factor on its objectness term.  With factor = world size * this rank's divisor / the job's divisor
 * (rows for box and class, cells for objectness; the job's divisors from the all-reduced sums at dh_yolo_loss_sums_offset)
 * the gradients that DDP averages (object_detection/yolov5/train.py:254-255,331-332) are those of the loss normalised over
 * the whole batch instead of per rank (loss.py:136,149,155). */
int dh_yolo_loss_bwd_scaled(int layout, const void* const* p, const int* ny, const int* nx, int nl, int bs, int na, int nc,
                            const int64_t* idx, const float* tbox, const float* anch, const int64_t* tcls, const int* counts,
                            int64_t cap, const dh_yolo_hyp* hyp, const float* grad_loss, const float* level_scale,
                            void* const* grad_p, const void* workspace, size_t workspace_bytes, void* stream);


/* N x M IoU of xyxy boxes, no eps (object_detection/yolov5/utils/metrics.py:242-264) and of
 * (w,h) pairs (metrics.py:291-296).  out is (N,M) row-major. */
int dh_box_iou(const float* box1, const float* box2, float* out, int64_t N, int64_t M, void* stream);
int dh_wh_iou(const float* wh1, const float* wh2, float* out, int64_t N, int64_t M, void* stream);

/* pool_nms (object_detection/detectors/centernet/dataset.py:164-169) over `planes` (W,H) maps:
 * out = heat where heat equals its kernel x kernel neighbourhood maximum, heat*0 elsewhere. */
int dh_centernet_pool_nms(const float* heat, float* out, int64_t planes, int W, int H, int kernel, void* stream);

/* decode_bbox (dataset.py:172-216): heat (B,C,W,H), size/offset (B,2,W,H) -> counts (B) and det, room for B*W*H rows
 * [x1,y1,x2,y2,conf,cls] (normalised): the rows of the whole batch packed, image after image, each image's in cell
 * order -- image b owns counts[b] rows starting at row counts[0] + ... + counts[b-1]. */
size_t dh_centernet_decode_workspace(int B, int W, int H);
int dh_centernet_decode(const float* heat, const float* size, const float* offset, int B, int C, int W, int H,
                        double confidence, float* det, int* counts, void* workspace, size_t workspace_bytes,
                        void* stream);

/* CenterNet postprocess, device part (object_detection/detectors/centernet/dataset.py:246-270): the detections of each
 * image regrouped by ascending class; with need_nms every class goes through torchvision nms(boxes, scores, nms_thres)
 * and keeps the survivors in score order (equal scores: input order), without it a class keeps its input order.
 * rows (*, 6) [x1,y1,x2,y2,conf,cls]: image b owns count[b] rows starting at row first[b] (a concatenation, the
 * packed buffer of dh_centernet_decode, or a padded one); max_rows >= every count, nc > every class id.
 * out: image b's result starts at row out_first[b] and has out_count[b] <= count[b] rows.  first, count, out_first,
 * out_count are device arrays of n_img ints.  One sequence of launches for the whole batch; nothing returns to the host. */
size_t dh_group_nms_workspace(int n_img, int max_rows, int nc);
int dh_group_nms(const float* rows, const int* first, const int* count, const int* out_first, int n_img, int max_rows, int nc,
                 int need_nms, double nms_thres, float* out, int* out_count, void* workspace, size_t workspace_bytes,
                 void* stream);

/* Target maps of CenternetDataset.__getitem__ (dataset.py:103-150) for a batch: labels (B,Lmax,5)
 * [x1,y1,x2,y2,cls] in input pixels with nlabels (B) valid rows each; outputs t_cls (B,C,W,H),
 * t_size / t_offset (B,2,W,H), t_mask (B,W,H); owner_ws is B*W*H ints of scratch.  All device memory. */
int dh_centernet_targets(const float* labels, const int* nlabels, int B, int Lmax, double input_w, double input_h, int C,
                         int W, int H, float* t_cls, float* t_size, float* t_offset, float* t_mask, int* owner_ws,
                         void* stream);

/* Evaluation-time matching of detections to labels for a batch of images in one launch
 * (object_detection/yolov5/val.py:71-92 process_batch; object_detection/yolov5/utils/metrics.py:125-163
 * ConfusionMatrix.process_batch; both on the IoU of metrics.py:242-264).
 * det (n_det, 6) rows [x1,y1,x2,y2,conf,cls] and labels (n_labels, 5) rows [cls,x1,y1,x2,y2] hold the images back to
 * back; det_first / label_first (device, n_img+1 ints) give each image's row range, or NULL when n_img == 1.
 * dh_eval_match: iouv (device, n_iou floats, ascending; iouv[0] is the pairing threshold) -> correct (n_det, n_iou)
 * bytes 0/1.  dh_confusion_update: adds this batch's counts to matrix ((nc+1) x (nc+1) uint64 on the device,
 * row = predicted class, column = true class, index nc = background), thresholds compared in fp32 as torch does.
 * Equal IoUs are resolved as a stable argsort followed by [::-1] would (the reference's argsort is unstable).
 * workspace: device, 256-byte aligned, dh_eval_match_workspace(n_det, n_labels) bytes. */
size_t dh_eval_match_workspace(int64_t n_det, int64_t n_labels);
int dh_eval_match(const float* det, const float* labels, const int* det_first, const int* label_first, int n_img,
                  int64_t n_det, int64_t n_labels, const float* iouv, int n_iou, unsigned char* correct, void* workspace,
                  size_t workspace_bytes, void* stream);
int dh_confusion_update(const float* det, const float* labels, const int* det_first, const int* label_first, int n_img,
                        int64_t n_det, int64_t n_labels, int nc, double conf, double iou_thres, unsigned long long* matrix,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Per-class average precision and precision / recall / F1 at the best mean-F1 confidence, from the statistics an
 * evaluation run collects (object_detection/yolov5/utils/metrics.py:21-86 ap_per_class with plot=False, :89-114
 * compute_ap).  All pointers are device memory; float64 arithmetic in the reference's operation order.
 * dh_ap_classes: target_cls (m doubles, integer class ids in [0, class_cap)) -> unique (ascending class ids) and nt
 *   (labels per class), both class_cap ints; meta[0] = number of classes with labels, meta[1] != 0 when an id was not an
 *   integer in range; bins is class_cap ints of scratch.  The caller reads meta back to size the outputs.
 * dh_ap_per_class: tp (n, niou) bytes 0/1, conf (n) and pred_cls (n) floats, unique / nt / n_unique from above ->
 *   ap (n_unique, niou), p_curve and r_curve (n_unique, 1000), out5 (5, n_unique) rows tp, fp, p, r, f1, best_index (the
 *   column of the curves that out5 was taken at).  Detections of one class with equal confidence keep their input order.
 * workspace: device, 256-byte aligned, dh_ap_per_class_workspace(n, niou) bytes (needs a GPU to size: the sort's
 *   scratch depends on the device). */
int dh_ap_classes(const double* target_cls, int64_t m, int class_cap, int* unique, int* nt, int* meta, int* bins, void* stream);
size_t dh_ap_per_class_workspace(int64_t n, int niou);
int dh_ap_per_class(const unsigned char* tp, const float* conf, const float* pred_cls, int64_t n, int niou, const int* unique,
                    const int* nt, int n_unique, int class_cap, double eps, double* ap, double* p_curve, double* r_curve,
                    double* out5, int* best_index, void* workspace, size_t workspace_bytes, void* stream);

/* AutoAnchor's fit of n anchors k (n,2) to N label sizes wh (N,2), all device fp32
 * (object_detection/yolov5/utils/autoanchor.py:86-90 metric, :92-94 anchor_fitness, :35-41 the check_anchors metric).
 * x (N,n) and best (N) are optional outputs (NULL to skip); sums (3 device doubles) receives the sum over labels of
 * best*(best>thr), the number of labels with best > thr, and the number of (label, anchor) pairs with x > thr.
 * thr is the inverted threshold (autoanchor.py:84), compared in fp32.  n <= 64. */
int dh_anchor_metric(const float* wh, int64_t N, const float* k, int n, double thr, float* x, float* best, double* sums,
                     void* stream);

/* LabelSmoothingFocalLoss.forward and its gradient (object_detection/loss.py:8-83; the CenterNet class loss of
 * object_detection/train.py:147-153, called at detectors/centernet/loss.py:17).  logits (N, C, S) fp32 with the spatial
 * axes flattened into S; one_hot != 0: labels (N, C, S), fp32 (label_i64 = 0) or int64 (label_i64 = 1), truncated to
 * integers as .long() does; one_hot == 0: labels (N, S) int64 class indices with label smoothing and, when has_ignore,
 * ignore_index.  out: 1 float (the mean over the N*S cells).  The forward leaves one byte per cell in the workspace
 * (dh_focal_loss_workspace(N*S) bytes, device, 256-byte aligned) that the backward reads; grad_out is 1 device float,
 * grad_logits (N, C, S) is written completely.  gamma >= 0; size_average and alpha have no effect in the reference. */
size_t dh_focal_loss_workspace(int64_t cells);
int dh_focal_loss_fwd(const float* logits, const void* labels, int64_t N, int C, int64_t S, int one_hot, int label_i64,
                      double gamma, double smoothing, int has_ignore, int64_t ignore_index, float* out, void* workspace,
                      size_t workspace_bytes, void* stream);
int dh_focal_loss_bwd(const float* logits, const void* labels, int64_t N, int C, int64_t S, int one_hot, int label_i64,
                      double gamma, double smoothing, int has_ignore, int64_t ignore_index, const float* grad_out,
                      float* grad_logits, const void* workspace, void* stream);

/* RegressionLossWithMask.forward and its gradient (object_detection/loss.py:86-100; the size / offset criteria that
 * detectors/centernet/loss.py:19-21 calls when no box criterion is set).  pred, target (B, 2, S) and mask (B, S) fp32 with
 * the spatial axes flattened into S; smooth != 0 selects smooth-L1 (beta 1).  out2: 2 device floats, the loss and its
 * denominator sum(mask)*2 + 1e-4 (the backward reads it); grad_out 1 device float; grad_pred (B, 2, S) written completely.
 * workspace: device, dh_regression_loss_workspace(B*S) bytes. */
size_t dh_regression_loss_workspace(int64_t cells);
int dh_regression_loss_fwd(const float* pred, const float* target, const float* mask, int64_t B, int64_t S, int smooth,
                           float* out2, void* workspace, size_t workspace_bytes, void* stream);
int dh_regression_loss_bwd(const float* pred, const float* target, const float* mask, int64_t B, int64_t S, int smooth,
                           const float* out2, const float* grad_out, float* grad_pred, void* stream);

/* Measurement aid, no counterpart in the reference: copies `bytes` (a multiple of 16, both pointers 16-byte aligned) with
 * one 16-byte non-temporal load and store per thread.  bench.py times it on the bytes of one decode launch: the copy rate
 * this memory system gives a kernel that only moves data, reported beside the nominal 8 TB/s. */
int dh_stream_copy(const void* src, void* dst, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DETHEAD_H */
