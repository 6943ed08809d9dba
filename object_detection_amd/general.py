"""Drop-ins for object_detection/yolov5/utils/general.py:543-560,619-738
(xyxy2xywh, xywh2xyxy, scale_coords, clip_coords, non_max_suppression)."""
import numpy as np
import torch

from . import _lib as L
from .yolo import _anchor_pixels, _anchor_pixels_half, objectness_plane

MAX_WH = 7680      # general.py:663
MAX_NMS = 30000    # general.py:664
K_SELECT = 3840    # candidates per image that go through NMS on the first attempt (0: all); see nms_launch
K_SMALL = -1       # k_select value: everything after the prefilter in one workgroup per image (<= SMALL_MAX candidates each)
K_CHAIN = -2       # k_select value: a prefix of about K_SELECT candidates, everything after the prefilter in one launch (class groups)
SMALL_MAX = 1024   # what that workgroup holds (SMALL_MAX of csrc/nms.hip)
SMALL_BELOW = 768  # candidate count of the busiest image up to which the next call of the same kind tries the small path

# Which first attempt to make is a guess about the data, so it is made from the last call of the same kind (same shape,
# thresholds and options): a detector at its usual confidence threshold leaves a few dozen candidates per image, every
# time, and a validation pass at conf 0.001 thousands, every time.  A wrong guess costs a repeat on the general path.
_SMALL_HINT = {}


def _first_attempt(key, max_det, k_select):
    if k_select is not None:
        return int(k_select)
    return K_SMALL if (max_det <= SMALL_MAX and _SMALL_HINT.get(key, False)) else K_CHAIN


def _remember(rows, key, used):
    rows._dh_attempt = (key, used)          # read back by nms_views, which sees the candidate counts


def _next_attempt(rows, need=-1):
    """After needs_full: the small attempt is followed by the prefix selection in one launch (class groups); that one by a
    prefix through the separate launches when some image was left to them (need = -2: a box far outside the image, too
    many equal scores), else -- like the separate launches themselves -- by the full sort."""
    used = getattr(rows, "_dh_attempt", (None, 0))[1]
    if used == K_SMALL:
        return K_CHAIN
    if used == K_CHAIN and need == -2:
        return K_SELECT
    return 0


def _convert(x, mode):
    if isinstance(x, torch.Tensor):
        src = L.dev_f32(x, "x")
        if src.dim() != 2 or src.shape[1] < 4:
            raise RuntimeError(f"expected (n, >=4) boxes, got {tuple(x.shape)}")
        if src.stride(1) != 1:
            src = src.contiguous()
        y = src.clone(memory_format=torch.contiguous_format)
        with L.device_guard(src.device):
            L.check(L.lib().dh_box_convert(L.ptr(src), src.stride(0), L.ptr(y), y.stride(0), src.shape[0], mode,
                                           L.stream()), "dh_box_convert")
        return y if x.dtype == torch.float32 else y.to(x.dtype)
    # numpy inputs stay on the host: that interface has no device to run on
    y = np.copy(x)
    if mode == 0:
        y[:, 0], y[:, 1] = x[:, 0] - x[:, 2] / 2, x[:, 1] - x[:, 3] / 2
        y[:, 2], y[:, 3] = x[:, 0] + x[:, 2] / 2, x[:, 1] + x[:, 3] / 2
    else:
        y[:, 0], y[:, 1] = (x[:, 0] + x[:, 2]) / 2, (x[:, 1] + x[:, 3]) / 2
        y[:, 2], y[:, 3] = x[:, 2] - x[:, 0], x[:, 3] - x[:, 1]
    return y


def xywh2xyxy(x):
    """(n,>=4) [x,y,w,h,...] -> [x1,y1,x2,y2,...] copy (general.py:553-560)."""
    return _convert(x, 0)


def xyxy2xywh(x):
    """(n,>=4) [x1,y1,x2,y2,...] -> [x,y,w,h,...] copy (general.py:543-550)."""
    return _convert(x, 1)


def _scale_clip(boxes, do_scale, pad, gain, shape):
    if boxes.dtype != torch.float32 or not boxes.is_cuda:
        raise RuntimeError("scale_coords/clip_coords modify fp32 ROCm tensors in place; got "
                           f"{boxes.dtype} on {boxes.device}")
    if boxes.dim() != 2 or boxes.shape[1] < 4 or boxes.stride(1) != 1:
        raise RuntimeError(f"expected a row-major (n, >=4) view, got shape {tuple(boxes.shape)} stride {boxes.stride()}")
    with L.device_guard(boxes.device):
        L.check(L.lib().dh_scale_clip_coords(L.ptr(boxes), boxes.stride(0), boxes.shape[0], int(do_scale), float(pad[0]),
                                             float(pad[1]), float(gain), float(shape[1]), float(shape[0]), L.stream()),
                "dh_scale_clip_coords")
    torch.autograd.graph.increment_version(boxes)      # written through the raw pointer: keep the version counter honest


def clip_coords(boxes, shape):
    """Clip xyxy boxes to (height, width) in place (general.py:635-644)."""
    if isinstance(boxes, torch.Tensor):
        _scale_clip(boxes, False, (0.0, 0.0), 1.0, shape)
    else:
        boxes[:, [0, 2]] = boxes[:, [0, 2]].clip(0, shape[1])
        boxes[:, [1, 3]] = boxes[:, [1, 3]].clip(0, shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Map xyxy coords from the letterboxed img1_shape back to img0_shape in place (general.py:619-632)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    _scale_clip(coords, True, pad, gain, img0_shape)
    return coords


def _with_apriori_labels(pred, labels, nc, corners=False):
    """general.py:678-684: label rows [cls,x,y,w,h] join an image's candidates with obj = cls score = 1.

    They are appended as extra prediction rows (objectness 0 rows pad shorter images and can never be
    candidates), which keeps them behind every real anchor in candidate order, as in the reference.
    corners: `pred` holds x1, y1, x2, y2 (_prepared_from_half), so the label boxes are converted the same way.
    """
    extra = max((len(l) for l in labels), default=0)
    if extra == 0:
        return pred
    bs = pred.shape[0]
    pad = torch.zeros((bs, extra, nc + 5), dtype=torch.float32, device=pred.device)
    for i, l in enumerate(labels):
        if len(l):
            l = l.to(pred.device).float()
            pad[i, :len(l), :4] = xywh2xyxy(l[:, 1:5]) if corners else l[:, 1:5]
            pad[i, :len(l), 4] = 1.0
            pad[i, torch.arange(len(l), device=pred.device), l[:, 0].long() + 5] = 1.0
    return torch.cat((pred, pad), 1)


def _prepared_from_half(prediction, conf_thres, labels):
    """The fp16 part of the reference's NMS on an fp16 prediction tensor (what a half-precision model hands it,
    val.py:129-130,203): conf = obj * cls and the corners rounded to fp16 (general.py:691,694), returned as the fp32
    tensor torch.cat leaves at general.py:699,702.  Images with apriori labels were widened before that, by the cat with the
    float label rows (general.py:684), and are computed in fp32.  See dh_nms_prepare_f16."""
    pred = prediction.contiguous()
    bs, A, no = pred.shape
    wide = None
    if labels and len(labels):
        wide = torch.tensor([1 if len(l) else 0 for l in labels], dtype=torch.uint8).to(pred.device)
    out = torch.empty((bs, A, no), dtype=torch.float32, device=pred.device)
    plane = torch.empty((bs, A), dtype=torch.float32, device=pred.device)       # channel 4 of `out` once more, compact
    with L.device_guard(pred.device):
        L.check(L.lib().dh_nms_prepare_f16(L.ptr(pred), bs, A, no - 5, float(conf_thres), L.ptr(wide), L.ptr(out), L.ptr(plane),
                                           L.stream()), "dh_nms_prepare_f16")
    return out, plane


def nms_launch(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
               labels=(), max_det=300, cap=None, k_select=None, merge=False):
    """Enqueue the whole NMS pipeline; nothing is copied to the host (safe inside HIP graph capture).

    Returns (rows, meta, cap): rows (bs*max_det, 6) packed detections, meta int32 [bs x {first row, row
    count}, bs x candidates found, bs x needs-full flag]; pass both to nms_finish.  (None, None, 0) when there is
    nothing to do.

    k_select > 0: only the best-scoring k_select candidates of each image (at least 8*max_det) are sorted and go
    through NMS, which is exact whenever they yield max_det rows (a row's fate depends only on rows ranked above
    it); images where they do not are flagged in meta and nms_views asks for a repeat with k_select=0.
    k_select = K_SMALL: one workgroup per image does everything after the prefilter (three launches instead of
    seven); images with more than SMALL_MAX candidates are flagged the same way.  k_select = K_CHAIN: about K_SELECT
    of the best candidates of each image go through sort, box fetch, per-class greedy NMS and collection in one
    launch (several workgroups per image, each with a share of the classes, LDS resident); images it leaves to the
    separate launches are flagged 2.  k_select = None picks K_SMALL
    when the last call of the same kind stayed below SMALL_BELOW candidates per image, K_CHAIN otherwise.
    merge: merge-NMS (general.py:725-731) on the result; it needs every candidate, so the general path runs (k_select=0).
    """
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    if isinstance(prediction, torch.Tensor) and (prediction.dim() != 3 or prediction.shape[2] < 6):
        raise RuntimeError(f"prediction must be (bs, A, 5+nc) with nc >= 1, got {tuple(prediction.shape)}")
    # an fp16 tensor gets the reference's fp16 arithmetic: where the rows are looked at (dh_nms_batched_f16), or, when some
    # image has apriori labels and is therefore widened first, in a pass of its own (see _prepared_from_half); anything else
    # is computed in fp32
    half = isinstance(prediction, torch.Tensor) and prediction.is_cuda and prediction.dtype == torch.float16 \
        and prediction.numel() > 0
    prepared = half and bool(labels) and len(labels) > 0
    plane = None
    if prepared:
        pred, plane = _prepared_from_half(prediction, conf_thres, labels)
    elif half:
        pred = prediction.contiguous()
    else:
        pred = L.dev_f32(prediction, "prediction").contiguous()
    bs, _, no = pred.shape
    nc = no - 5
    multi_label = bool(multi_label) and nc > 1                                      # general.py:667
    if labels and len(labels):
        pred = _with_apriori_labels(pred, labels, nc, corners=prepared)
        plane = None                                  # the label rows are not in it
    elif pred is prediction:
        plane = objectness_plane(prediction)          # compact obj column left behind by detect_decode, if still valid
    A = pred.shape[1]
    dev = pred.device
    max_det = int(max_det)
    if bs == 0 or A == 0 or max_det < 1:
        return None, None, 0
    keep_mask = _class_keep(classes, nc, dev)
    lib = L.lib()
    cap = A if cap is None else int(cap)
    rows = torch.empty((bs * max_det, 6), dtype=torch.float32, device=dev)
    meta = torch.empty(4 * bs, dtype=torch.int32, device=dev)
    ws_bytes = lib.dh_nms_workspace(bs, nc, cap, MAX_NMS)
    ws = L.workspace(ws_bytes, dev)
    key = ("nms", bs, A, nc, float(conf_thres), float(iou_thres), multi_label, bool(agnostic), keep_mask is None, max_det, prepared, half)
    k_select = 0 if merge else _first_attempt(key, max_det, k_select)
    _remember(rows, key, k_select)
    tail = (int(multi_label), int(bool(agnostic)), L.ptr(keep_mask), max_det, MAX_NMS, float(MAX_WH), cap, int(k_select),
            L.ptr(rows), L.ptr(meta), L.c_void_p(meta.data_ptr() + 8 * bs), L.c_void_p(meta.data_ptr() + 12 * bs), L.ptr(ws),
            ws_bytes, L.stream())
    with L.device_guard(dev):
        if prepared:
            L.check(lib.dh_nms_batched_prepared(L.ptr(pred), L.ptr(plane), bs, A, nc, float(conf_thres), float(iou_thres), *tail),
                    "dh_nms_batched_prepared")
        elif half:
            L.check(lib.dh_nms_batched_f16(L.ptr(pred), L.ptr(plane), bs, A, nc, float(conf_thres), float(iou_thres), *tail),
                    "dh_nms_batched_f16")
        else:
            L.check(lib.dh_nms_batched(L.ptr(pred), L.ptr(plane), bs, A, nc, float(conf_thres), float(iou_thres), *tail),
                    "dh_nms_batched")
        if merge:
            _merge(lib, rows, meta, bs, nc, cap, max_det, iou_thres, agnostic, ws, ws_bytes)
    return rows, meta, cap


def _merge(lib, rows, meta, bs, nc, cap, max_det, iou_thres, agnostic, ws, ws_bytes):
    """Merge-NMS over the result the NMS call left in rows / meta and the candidates it left in its workspace."""
    L.check(lib.dh_nms_merge(L.ptr(rows), L.ptr(meta), bs, nc, cap, MAX_NMS, max_det, float(iou_thres), int(bool(agnostic)),
                             float(MAX_WH), 1, L.ptr(ws), ws_bytes, L.stream()), "dh_nms_merge")


def _class_keep(classes, nc, dev):
    if classes is None:
        return None
    flags = torch.zeros(nc, dtype=torch.uint8)
    idx = [int(c) for c in classes if float(c) == int(c) and 0 <= int(c) < nc]
    if idx:
        flags[idx] = 1
    return flags.to(dev)


def detect_nms_launch(levels, anchors, stride, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                      multi_label=False, max_det=300, cap=None, k_select=None, merge=False, conv_na=None, half_arithmetic=False):
    """nms_launch fed by the raw head maps: the same (rows, meta, cap) as
    nms_launch(detect_decode(levels, anchors, stride), ...), without writing or reading the decoded tensor.
    conv_na = na: `levels` are the outputs of the head's convolutions, (bs, na*no, ny, nx), instead of the maps.
    half_arithmetic (fp16 maps): the roundings of a half-precision model and of NMS on its fp16 tensor, i.e. the rows of
    nms_launch(detect_decode(levels, anchors, stride, half_arithmetic=True), ...)."""
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    xs, half = L.head_maps(levels)
    nl = len(xs)
    if half_arithmetic and not half:
        raise RuntimeError("half_arithmetic needs fp16 head maps or fp16 convolution outputs")
    if conv_na:
        na, bs = int(conv_na), xs[0].shape[0]
        no = xs[0].shape[1] // na
        for x in xs:
            if x.dim() != 4 or x.shape[0] != bs or x.shape[1] != na * no:
                raise RuntimeError(f"conv_nms: convolution outputs must be (bs, na*no, ny, nx), got {tuple(x.shape)}")
    else:
        bs, na, _, _, no = xs[0].shape
        for x in xs:
            if x.dim() != 5 or x.shape[0] != bs or x.shape[1] != na or x.shape[4] != no:
                raise RuntimeError(f"detect_nms: inconsistent level shape {tuple(x.shape)}")
    if no < 6:
        raise RuntimeError(f"detect_nms: maps must have 5+nc channels with nc >= 1, got {no}")
    nc = no - 5
    ny = [int(x.shape[-3 if not conv_na else 2]) for x in xs]
    nx = [int(x.shape[-2 if not conv_na else 3]) for x in xs]
    A = sum(na * a * b for a, b in zip(ny, nx))
    dev = xs[0].device
    max_det = int(max_det)
    if bs == 0 or A == 0 or max_det < 1:
        return None, None, 0
    multi_label = bool(multi_label) and nc > 1
    keep_mask = _class_keep(classes, nc, dev)
    anchor_px, strides = _anchor_pixels_half(anchors, stride) if half_arithmetic else _anchor_pixels(anchors, stride)
    lib = L.lib()
    cap = A if cap is None else int(cap)
    rows = torch.empty((bs * max_det, 6), dtype=torch.float32, device=dev)
    meta = torch.empty(4 * bs, dtype=torch.int32, device=dev)
    ws_bytes = lib.dh_nms_workspace(bs, nc, cap, MAX_NMS)
    ws = L.workspace(ws_bytes, dev)
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in xs])
    key = (("conv_nms" if conv_na else "detect_nms") + ("_half" if half_arithmetic else ""), bs, A, nc, float(conf_thres), float(iou_thres), multi_label, bool(agnostic), keep_mask is None, max_det)
    k_select = 0 if merge else _first_attempt(key, max_det, k_select)
    _remember(rows, key, k_select)
    with L.device_guard(dev):
        if conv_na:
            entry = lib.dh_yolo_conv_nms_f16 if half else lib.dh_yolo_conv_nms
        else:
            entry = lib.dh_yolo_decode_nms_f16 if half else lib.dh_yolo_decode_nms     # fp16 maps are widened on load
        if half_arithmetic:                                                            # ... or keep the fp16 roundings
            entry = lib.dh_yolo_conv_nms_half if conv_na else lib.dh_yolo_decode_nms_half
        L.check(entry(ptrs, L.int_array(ny), L.int_array(nx), nl, bs, na, no, anchor_px, strides,
                      float(conf_thres), float(iou_thres), int(multi_label), int(bool(agnostic)),
                      L.ptr(keep_mask), max_det, MAX_NMS, float(MAX_WH), cap, int(k_select), L.ptr(rows),
                      L.ptr(meta), L.c_void_p(meta.data_ptr() + 8 * bs),
                      L.c_void_p(meta.data_ptr() + 12 * bs), L.ptr(ws), ws_bytes, L.stream()),
                "dh_yolo_decode_nms")
        if merge:
            _merge(lib, rows, meta, bs, nc, cap, max_det, iou_thres, agnostic, ws, ws_bytes)
    return rows, meta, cap


def detect_nms(levels, anchors, stride, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
               labels=(), max_det=300, merge=False, half_arithmetic=False):
    """non_max_suppression(Detect(x)[0], ...) (val.py:194,203; detect.py:126,132) straight from the raw head maps
    `levels` (list of (bs,na,ny,nx,5+nc)); anchors / stride as Detect holds them.  Bit-identical to the two-call
    sequence, at about half its HBM traffic: the decoded (bs, A, 5+nc) tensor is neither written nor read.
    labels: a-priori label rows per image (general.py:678-684, the auto-labelling mode of val.py:203).  They join the
    candidates as rows of the decoded tensor, so with labels the decoded tensor is made and the two-call sequence runs.
    half_arithmetic (fp16 maps only): the detections a half-precision model yields (model.half(), val.py:129-130, Detect in
    place): sigmoid, box, conf = obj * cls and the corners rounded to fp16 where torch rounds them -- bit-identical to
    non_max_suppression(detect_decode(levels, anchors, stride, half_arithmetic=True), ...).  Without it fp16 maps are
    widened on load and everything is fp32 arithmetic."""
    if labels is not None and len(labels) and any(len(l) for l in labels):
        from .yolo import detect_decode
        return non_max_suppression(detect_decode(levels, anchors, stride, half_arithmetic=half_arithmetic), conf_thres, iou_thres,
                                   classes, agnostic, multi_label, labels, max_det, merge)
    return _until_complete(lambda cap, k_select: detect_nms_launch(levels, anchors, stride, conf_thres, iou_thres, classes, agnostic,
                                                                   multi_label, max_det, cap, k_select, merge,
                                                                   half_arithmetic=half_arithmetic),
                           levels[0].device, levels[0].shape[0])


def conv_nms(convs, na, anchors, stride, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
             labels=(), max_det=300, merge=False, half_arithmetic=False):
    """non_max_suppression(Detect(x)[0], ...) (detect.py:126,132; val.py:194,203 without compute_loss) straight from the
    outputs of Detect's 1x1 convolutions, `convs` = list of (bs, na*no, ny, nx): what Detect.forward holds before the
    layout change of yolo.py:54.  Neither the head maps nor the decoded tensor are written or read; the detections are
    bit-identical to non_max_suppression(head_maps_decode(convs, ...)[0], ...).
    labels: as for detect_nms (with labels the decoded tensor is made: head_maps_decode, then non_max_suppression).
    half_arithmetic (fp16 convolution outputs only): as for detect_nms -- the rows of non_max_suppression on the fp16 tensor a
    half-precision model decodes in place."""
    if labels is not None and len(labels) and any(len(l) for l in labels):
        from .yolo import detect_decode, head_maps, head_maps_decode
        if half_arithmetic:
            z = detect_decode(head_maps(convs, na), anchors, stride, half_arithmetic=True)
        else:
            z = head_maps_decode(convs, na, anchors, stride, lazy=True)[0]
        return non_max_suppression(z, conf_thres, iou_thres, classes, agnostic, multi_label, labels, max_det, merge)
    return _until_complete(lambda cap, k_select: detect_nms_launch(convs, anchors, stride, conf_thres, iou_thres, classes, agnostic,
                                                                   multi_label, max_det, cap, k_select, merge, conv_na=na,
                                                                   half_arithmetic=half_arithmetic),
                           convs[0].device, convs[0].shape[0])


def nms_meta_to_host(meta, pinned=None):
    """Start the one device-to-host copy of the NMS bookkeeping (3*bs int32) on the current stream.

    Returns (host tensor in pinned memory, event recorded after the copy): a caller that wants to keep the
    GPU busy can enqueue more work before waiting on the event (software pipelining of consecutive batches).
    """
    if pinned is None:
        pinned = torch.empty(meta.numel(), dtype=torch.int32, pin_memory=True)
    pinned.copy_(meta, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    return pinned, done


def nms_views(rows, host, cap):
    """Per-image (k,6) views of the packed rows from the host copy of meta.

    (None, needed_cap) when the candidate capacity overflowed, (None, -1) when some image needs the full sort,
    (None, -2) when some image needs the separate launches (see nms_launch, _next_attempt): the caller repeats the
    launch with that cap, or with the next k_select."""
    bs = len(host) // 4
    busiest = max(host[2 * bs:3 * bs])
    key = getattr(rows, "_dh_attempt", (None, 0))[0]
    if key is not None:
        if len(_SMALL_HINT) > 64:
            _SMALL_HINT.clear()
        _SMALL_HINT[key] = busiest <= SMALL_BELOW
    if busiest > cap:
        return None, busiest
    flags = host[3 * bs:]
    if any(flags):
        return None, (-1 if 1 in flags else -2)
    # images are packed in completion order: one split() call makes every per-image view
    first, count = host[0:2 * bs:2], host[1:2 * bs:2]
    order = [i for i in sorted(range(bs), key=first.__getitem__) if count[i]]
    result = [torch.zeros((0, 6), device=rows.device)] * bs
    if order:
        at, packed = 0, True
        for i in order:
            packed = packed and first[i] == at
            at += count[i]
        if packed:
            for i, piece in zip(order, rows[:at].split([count[i] for i in order])):
                result[i] = piece
        else:       # merge-NMS dropped rows of some images: each image still starts at its first row
            for i in order:
                result[i] = rows[first[i]:first[i] + count[i]]
    return result, cap


def nms_finish(rows, meta, cap):
    """Blocking variant: copy meta to the host now and build the views."""
    return nms_views(rows, meta.tolist(), cap)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, merge=False):
    """Batched NMS on (bs, A, 5+nc) predictions -> list of (k,6) [xyxy, conf, cls] per image (general.py:647-738).

    merge is the one addition to the reference's signature: there the merge-NMS step (general.py:725-731) sits behind a
    local constant (`merge = False`, general.py:668); the default reproduces that, merge=True runs the step.

    The whole batch is processed by one sequence of kernel launches and a single device-to-host copy
    of the per-image counts; the returned tensors are views of one packed buffer.
    The reference's 10 s wall-clock bail-out (general.py:734-736) is not reproduced.
    """
    return _until_complete(lambda cap, k_select: nms_launch(prediction, conf_thres, iou_thres, classes, agnostic, multi_label,
                                                            labels, max_det, cap, k_select, merge),
                           prediction.device, prediction.shape[0])


def _until_complete(launch, device, bs):
    """Runs launch(cap, k_select) -> (rows, meta, cap) and reads its bookkeeping back until it covers every image: an
    attempt that left some image to a more general one is followed by that one, a candidate list that overflowed by a
    launch with room for all of it.  Returns the per-image views."""
    cap, k_select = None, None
    while True:
        rows, meta, cap = launch(cap, k_select)
        if rows is None:
            return [torch.zeros((0, 6), device=device)] * bs
        result, need = nms_finish(rows, meta, cap)
        if result is not None:
            return result
        if need < 0:
            k_select = _next_attempt(rows, need)      # the attempt did not cover some image: the next, more general one
        else:
            cap = need        # multi-label produced more candidates than rows: rerun with room for all of them
