"""Drop-in for the detection/label matching of validation (object_detection/yolov5/val.py:71-92)."""
import torch

from . import _lib as L


def _rows(t, width, name):
    t = L.dev_f32(t, name)
    if t.dim() != 2 or t.shape[1] != width:
        raise RuntimeError(f"{name} must be (n, {width}), got {tuple(t.shape)}")
    return t.contiguous()


def _offsets(sizes, device):
    """Row ranges of the images inside a concatenated tensor, as n_img+1 int32 on the device (one small async copy)."""
    first = [0]
    for n in sizes:
        first.append(first[-1] + int(n))
    return torch.tensor(first, dtype=torch.int32).pin_memory().to(device, non_blocking=True), first


def _match(det, labels, det_first, label_first, n_img, iouv):
    iouv = L.dev_f32(iouv, "iouv").contiguous()
    n_det, n_lab, n_iou = int(det.shape[0]), int(labels.shape[0]), int(iouv.numel())
    correct = torch.empty((n_det, n_iou), dtype=torch.bool, device=det.device)
    if n_det == 0:
        return correct
    lib = L.lib()
    ws_bytes = lib.dh_eval_match_workspace(n_det, n_lab)
    ws = L.workspace(ws_bytes, det.device)
    with L.device_guard(det.device):
        L.check(lib.dh_eval_match(L.ptr(det), L.ptr(labels), L.ptr(det_first), L.ptr(label_first), n_img, n_det, n_lab,
                                  L.ptr(iouv), n_iou, L.ptr(correct), L.ptr(ws), ws_bytes, L.stream()), "dh_eval_match")
    return correct


def process_batch(detections, labels, iouv):
    """Correct-prediction matrix of ONE image: detections (N,6) [x1,y1,x2,y2,conf,cls], labels (M,5) [cls,x1,y1,x2,y2],
    iouv (n_iou,) -> bool (N, n_iou) on iouv's device (val.py:71-92).  One kernel launch, no device-to-host copy
    (the reference moves the candidate pairs to the host and back)."""
    det = _rows(detections, 6, "detections")
    lab = _rows(labels, 5, "labels")
    return _match(det, lab, None, None, 1, iouv)


def process_batches(detections, labels, iouv):
    """process_batch for every image of a batch in one launch: lists of per-image (N_i,6) and (M_i,5) tensors
    -> list of bool (N_i, n_iou)."""
    if len(detections) != len(labels):
        raise RuntimeError("process_batches needs one label tensor per detection tensor")
    if not detections:
        return []
    dets = [_rows(d, 6, f"detections[{i}]") for i, d in enumerate(detections)]
    labs = [_rows(t, 5, f"labels[{i}]") for i, t in enumerate(labels)]
    det_first, first = _offsets([d.shape[0] for d in dets], dets[0].device)
    label_first, _ = _offsets([t.shape[0] for t in labs], dets[0].device)
    correct = _match(torch.cat(dets), torch.cat(labs), det_first, label_first, len(dets), iouv)
    return list(correct.split([b - a for a, b in zip(first[:-1], first[1:])]))
