"""Drop-ins for the CenterNet head glue.

  object_detection/detectors/centernet/loss.py:4-28        offset_to_coord, compute_losses
  object_detection/detectors/centernet/dataset.py:96-161   target maps (batched, on the device)
  object_detection/detectors/centernet/dataset.py:164-297  pool_nms, decode_bbox, postprocess, postprocess_predictions
"""
import numpy as np
import torch

from . import _lib as L
from .loss import IoULossWithMask, masked_ciou_loss, masked_ciou_loss_forward_backward


def pool_nms(heat, kernel=3):
    """heat (B,C,W,H) -> heat * (maxpool_kxk(heat) == heat)  (dataset.py:164-169)."""
    h = L.dev_f32(heat, "heat").contiguous()
    if h.dim() != 4:
        raise RuntimeError(f"pool_nms: expected (B,C,W,H), got {tuple(heat.shape)}")
    out = torch.empty_like(h)
    with L.device_guard(h.device):
        L.check(L.lib().dh_centernet_pool_nms(L.ptr(h), L.ptr(out), h.shape[0] * h.shape[1], h.shape[2], h.shape[3],
                                              int(kernel), L.stream()), "dh_centernet_pool_nms")
    return out


class _Decoded(list):
    """decode_bbox's per-image list, which also remembers the packed device buffer the entries are views of, so that
    postprocess can hand the whole batch to one kernel sequence without concatenating them again."""
    buffer = None        # (rows of the whole batch, 6), image after image
    first = None         # first row of every image in it
    num_classes = 0
    version = -1         # autograd version counter of the buffer when it was written


def decode_bbox(pred_cls, pred_size, pred_offset, confidence, device=None):
    """Per image: [] or (k,6) [x1,y1,x2,y2,conf,cls] normalised boxes in cell order (dataset.py:172-216).

    Peak filter, tanh, best class and thresholding run in three launches for the whole batch, which leave the rows of all
    images packed in one buffer; one host copy of the per-image counts and one split of that buffer replace the
    reference's per-image loop.
    """
    heat = L.dev_f32(pred_cls, "pred_cls").contiguous()
    size = L.dev_f32(pred_size, "pred_size").contiguous()
    off = L.dev_f32(pred_offset, "pred_offset").contiguous()
    B, C, W, H = heat.shape
    if tuple(size.shape) != (B, 2, W, H) or tuple(off.shape) != (B, 2, W, H):
        raise RuntimeError("decode_bbox: size/offset maps must be (B,2,W,H) matching the class map")
    lib = L.lib()
    ws_bytes = lib.dh_centernet_decode_workspace(B, W, H)
    ws = L.workspace(ws_bytes, heat.device)
    det = torch.empty((B * W * H, 6), dtype=torch.float32, device=heat.device)
    counts = torch.empty(B, dtype=torch.int32, device=heat.device)
    with L.device_guard(heat.device):
        L.check(lib.dh_centernet_decode(L.ptr(heat), L.ptr(size), L.ptr(off), B, C, W, H, float(confidence), L.ptr(det),
                                        L.ptr(counts), L.ptr(ws), ws_bytes, L.stream()), "dh_centernet_decode")
    sizes = counts.tolist()
    total = sum(sizes)
    out = _Decoded(d if k else [] for d, k in zip(det[:total].split(sizes), sizes)) if B else _Decoded()
    first = [0] * B
    for i in range(1, B):
        first[i] = first[i - 1] + sizes[i - 1]
    out.buffer, out.first, out.num_classes, out.version = det, first, int(C), det._version
    return out


def centernet_correct_boxes(box_xy, box_wh, input_shape, image_shape, letterbox_image):
    """dataset.py:219-243.  numpy in, numpy out: the reference does this step on the host after .cpu()."""
    net, img = np.array(input_shape), np.array(image_shape)
    if letterbox_image:      # undo the letterbox: the image sits centred in the network input, scaled to fit
        fitted = np.round(img * np.min(net / img))
        box_xy = (box_xy - (net - fitted) / 2.0 / net) * (net / fitted)
        box_wh *= net / fitted                      # in place: the caller's array changes, as it does in the reference
    half = box_wh / 2.0
    corners = np.concatenate([(box_xy - half)[..., :2], (box_xy + half)[..., :2]], axis=-1)
    corners *= img                                  # normalised -> pixels
    return corners


def _group_by_class(prediction, need_nms, nms_thres):
    """Device part of postprocess for the whole batch: rows of every image regrouped by ascending class, each class
    through NMS when asked.  Returns (rows on the host (n,6) float32, [first row, row count] per image)."""
    sizes = [len(d) for d in prediction]
    B = len(sizes)
    out_first = [0] * B
    for i in range(1, B):
        out_first[i] = out_first[i - 1] + sizes[i - 1]
    total = out_first[-1] + sizes[-1]
    if isinstance(prediction, _Decoded) and prediction.buffer is not None and len(prediction.first) == B and all(
            len(d) == 0 or (d.data_ptr() == prediction.buffer.data_ptr() + 24 * prediction.first[i] and d._version == prediction.version)
            for i, d in enumerate(prediction)):      # still the untouched views decode_bbox made
        rows = prediction.buffer
        first = prediction.first
        nc = prediction.num_classes
    else:
        rows = torch.cat([L.dev_f32(d, f"prediction[{i}]").reshape(-1, 6) for i, d in enumerate(prediction) if len(d)])
        first = out_first
        nc = int(rows[:, 5].max()) + 1          # the only place that has to look at the data before launching
        if nc < 1 or int(rows[:, 5].min()) < 0:
            raise RuntimeError("postprocess: class ids must be non-negative")
    dev = rows.device
    index = torch.tensor([first, sizes, out_first], dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
    out = torch.empty((total, 6), dtype=torch.float32, device=dev)
    out_count = torch.empty(B, dtype=torch.int32, device=dev)
    lib = L.lib()
    ws_bytes = lib.dh_group_nms_workspace(B, max(sizes), nc)
    ws = L.workspace(ws_bytes, dev)
    with L.device_guard(dev):
        L.check(lib.dh_group_nms(L.ptr(rows), L.ptr(index[0]), L.ptr(index[1]), L.ptr(index[2]), B, max(sizes), nc,
                                 int(bool(need_nms)), float(nms_thres), L.ptr(out), L.ptr(out_count), L.ptr(ws), ws_bytes,
                                 L.stream()), "dh_group_nms")
    return out.cpu().numpy(), list(zip(out_first, out_count.tolist()))


def postprocess(prediction, need_nms, image_shape, input_shape, letterbox_image, nms_thres=0.4):
    """dataset.py:246-284 -> list of numpy (k,6) arrays or None; rows grouped by ascending class,
    inside a class in score order when need_nms else in input order.

    The class grouping and the per-class NMS of every image run as one sequence of launches (the reference loops
    over images and classes with a host round trip each); the box rescaling stays the reference's numpy arithmetic on
    the host, where the result has to end up anyway."""
    output = [None for _ in range(len(prediction))]
    if not any(len(d) for d in prediction):
        return output
    rows, spans = _group_by_class(prediction, need_nms, nms_thres)
    for i, (start, k) in enumerate(spans):
        if len(prediction[i]) == 0:
            continue
        arr = rows[start:start + k].copy()
        box_xy, box_wh = (arr[:, 0:2] + arr[:, 2:4]) / 2, arr[:, 2:4] - arr[:, 0:2]
        arr[:, :4] = centernet_correct_boxes(box_xy, box_wh, input_shape, image_shape, letterbox_image)
        output[i] = arr
    return output


def postprocess_predictions(pred, image_size, device=None):
    """dataset.py:287-297: confidence 0.9, no NMS, input_shape = image_size / 4."""
    outputs = decode_bbox(pred[0], pred[1], pred[2], confidence=0.9, device=device)
    return postprocess(outputs, need_nms=False, image_shape=image_size, input_shape=image_size / 4,
                       letterbox_image=False, nms_thres=0.4)


def centernet_targets(labels, num_classes, input_shape, output_shape, device=None):
    """Target maps of CenternetDataset.__getitem__ (dataset.py:103-150) for a whole batch on the device.

    labels: list of (n_i,5) arrays/tensors [x1,y1,x2,y2,cls] in input pixels.
    Returns (cls (B,C,W,H), size (B,2,W,H), offset (B,2,W,H), mask (B,W,H)), what the reference's
    DataLoader workers build one sample at a time in numpy.
    """
    B = len(labels)
    dev = torch.device(device) if device is not None else next(
        (l.device for l in labels if isinstance(l, torch.Tensor) and l.is_cuda), torch.device("cuda"))
    W, H = int(output_shape[0]), int(output_shape[1])
    lmax = max((len(l) for l in labels), default=0)
    packed = torch.zeros((B, max(lmax, 1), 5), dtype=torch.float32)
    for i, l in enumerate(labels):
        if len(l):
            packed[i, :len(l)] = torch.as_tensor(np.asarray(l.cpu() if isinstance(l, torch.Tensor) else l)[:, :5],
                                                 dtype=torch.float32)
    nl = torch.tensor([len(l) for l in labels], dtype=torch.int32)
    packed, nl = packed.to(dev), nl.to(dev)
    t_cls = torch.empty((B, num_classes, W, H), dtype=torch.float32, device=dev)
    t_size = torch.empty((B, 2, W, H), dtype=torch.float32, device=dev)
    t_off = torch.empty((B, 2, W, H), dtype=torch.float32, device=dev)
    t_mask = torch.empty((B, W, H), dtype=torch.float32, device=dev)
    owner = torch.empty((B, W, H), dtype=torch.int32, device=dev)
    with L.device_guard(dev):
        L.check(L.lib().dh_centernet_targets(L.ptr(packed), L.ptr(nl), B, lmax, float(input_shape[0]), float(input_shape[1]),
                                             int(num_classes), W, H, L.ptr(t_cls), L.ptr(t_size), L.ptr(t_off),
                                             L.ptr(t_mask), L.ptr(owner), L.stream()), "dh_centernet_targets")
    return t_cls, t_size, t_off, t_mask


def offset_to_coord(pred):
    """pred (B,2,W,H): plane 0 += index along dim 2, plane 1 += index along dim 3 (loss.py:4-13)."""
    p = L.dev_f32(pred, "pred")
    _, _, dim_w, dim_h = p.shape
    ix = torch.arange(dim_w, device=p.device, dtype=torch.float32).view(1, 1, dim_w, 1)
    iy = torch.arange(dim_h, device=p.device, dtype=torch.float32).view(1, 1, 1, dim_h)
    return p + torch.cat([ix.expand(1, 1, dim_w, dim_h), iy.expand(1, 1, dim_w, dim_h)], dim=1)


def box_loss_forward_backward(pred_offset, pred_size, label_offset, label_size, mask, grad_loss=None):
    """The box branch of compute_losses (object_detection/detectors/centernet/loss.py:22-28 with criterion["box"] =
    IoULossWithMask) and its gradients in one call of two launches, without the autograd engine:
    (loss, d loss / d pred_offset, d loss / d pred_size).  The coordinate add of offset_to_coord, the mask multiply, CIoU,
    the reduction and the gradient never leave the kernels; values equal compute_losses + backward() bit for bit."""
    return masked_ciou_loss_forward_backward(pred_offset, pred_size, label_offset, label_size, mask, add_coords=True,
                                             grad_loss=grad_loss)


def compute_losses(pred, labels, criterion, losses):
    """loss.py:16-28; fills the `losses` dict in place.

    When the box criterion is this package's IoULossWithMask the coordinate add, the channel
    concat, the mask multiply, CIoU and the reduction run as one kernel; any other criterion gets
    the tensors the reference would have built.
    """
    losses["cls"] = criterion["cls"](pred[0], labels[0])
    if criterion["box"] is None:
        losses["size"] = criterion["size"](pred[1], labels[1], labels[3])
        losses["offset"] = criterion["offset"](pred[2], labels[2], labels[3])
    elif isinstance(criterion["box"], IoULossWithMask):
        losses["box"] = masked_ciou_loss(pred[2], pred[1], labels[2], labels[1], labels[3], add_coords=True)
    else:
        losses["box"] = criterion["box"](torch.cat([offset_to_coord(pred[2]), pred[1]], dim=1),
                                         torch.cat([offset_to_coord(labels[2]), labels[1]], dim=1), labels[3])
