"""Drop-in for torchvision.ops.nms as the reference calls it
(object_detection/yolov5/utils/general.py:722, object_detection/detectors/centernet/dataset.py:263)."""
import torch

from . import _lib as L


def nms(boxes, scores, iou_threshold):
    """Greedy NMS of xyxy boxes -> int64 indices of kept boxes, highest score first.

    Suppression is strict (IoU > threshold, compared in double like the CPU kernel of torchvision
    0.11.1); equal scores are ordered by ascending index (the stable choice among the orders the
    reference's unstable sort may produce).
    """
    b = L.dev_f32(boxes, "boxes").contiguous()
    s = L.dev_f32(scores, "scores").contiguous()
    n = b.shape[0]
    if b.dim() != 2 or b.shape[1] != 4 or s.dim() != 1 or s.shape[0] != n:
        raise RuntimeError(f"nms: boxes must be (N,4) and scores (N,), got {tuple(boxes.shape)} and {tuple(scores.shape)}")
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=b.device)
    lib = L.lib()
    ws_bytes = lib.dh_nms_plain_workspace(n)
    ws = L.workspace(ws_bytes, b.device)
    keep = torch.empty(n, dtype=torch.int64, device=b.device)
    count = torch.empty(1, dtype=torch.int32, device=b.device)
    with L.device_guard(b.device):
        L.check(lib.dh_nms_plain(L.ptr(b), L.ptr(s), n, float(iou_threshold), L.ptr(keep), L.ptr(count), L.ptr(ws),
                                 ws_bytes, L.stream()), "dh_nms_plain")
    return keep[:int(count.item())]
