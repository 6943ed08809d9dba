"""Oracle: greedy NMS and the YOLOv5 ``non_max_suppression`` wrapper (CPU, test-only).

Restates object_detection/yolov5/utils/general.py:647-738.  The greedy kernel itself
(torchvision.ops.nms, absent from the reference tree) is ``csrc/oracle_nms.c``.
Sort order is the stable refinement of the reference's unstable descending sorts:
equal scores keep ascending original index (general.py:717 and inside torchvision).
The 10 s wall-clock bail-out of general.py:734-736 is deliberately not restated.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

from .boxes import box_iou, xywh2xyxy

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_WH = 7680      # general.py:663
MAX_NMS = 30000    # general.py:664


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        lib = ctypes.CDLL(path)
        lib.oracle_nms.restype = ctypes.c_int64
        lib.oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                   ctypes.c_double, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms(boxes[N,4] xyxy, scores[N], thr) -> int64[K], score-descending."""
    b = np.ascontiguousarray(boxes.detach().cpu().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
    n = b.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    k = _lib().oracle_nms(b.ctypes.data, s.ctypes.data, n, float(iou_threshold), keep.ctypes.data)
    return torch.from_numpy(keep[:k].copy())


def nms_py(boxes, scores, iou_threshold):
    """Pure-numpy twin of csrc/oracle_nms.c for tiny cases (cross-check of the C code)."""
    b = np.asarray(boxes, dtype=np.float32)
    s = np.asarray(scores, dtype=np.float32)
    n = len(s)
    order = sorted(range(n), key=lambda k: (-float(s[k]), k))
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = [False] * n
    keep = []
    zero = np.float32(0)
    for p, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        for j in order[p + 1:]:
            if dead[j]:
                continue
            w = max(zero, min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]))
            h = max(zero, min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]))
            inter = np.float32(w * h)
            with np.errstate(invalid="ignore", divide="ignore"):
                ovr = inter / np.float32(np.float32(area[i] + area[j]) - inter)
            if float(ovr) > float(iou_threshold):
                dead[j] = True
    return np.asarray(keep, dtype=np.int64)


def stable_desc_order(scores):
    """Indices sorting ``scores`` descending with ties in ascending index order."""
    return torch.sort(scores, descending=True, stable=True)[1]


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, labels=(), max_det=300, nms_fn=None, merge=False):
    """general.py:647-738 -> list of (k,6) tensors [x1,y1,x2,y2,conf,cls] per image.

    merge: the reference's local switch of general.py:668 (always False there) made a parameter; True runs the
    merge-NMS step of general.py:725-731 with redundant = True (general.py:666)."""
    nms_fn = nms if nms_fn is None else nms_fn
    assert 0 <= conf_thres <= 1, f'Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0'
    assert 0 <= iou_thres <= 1, f'Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0'
    nc = prediction.shape[2] - 5
    multi_label = multi_label and nc > 1
    results = [torch.zeros((0, 6), device=prediction.device)] * prediction.shape[0]

    for img, rows in enumerate(prediction):
        rows = rows[rows[:, 4] > conf_thres]                      # :656,675
        if labels and len(labels[img]):                            # :678-684
            lab = labels[img]
            extra = torch.zeros((len(lab), nc + 5), device=rows.device)
            extra[:, :4] = lab[:, 1:5]
            extra[:, 4] = 1.0
            extra[range(len(lab)), lab[:, 0].long() + 5] = 1.0
            rows = torch.cat((rows, extra), 0)
        if rows.shape[0] == 0:
            continue
        rows[:, 5:] *= rows[:, 4:5]                                # :691 (rows is a fresh copy)
        corners = xywh2xyxy(rows[:, :4])                           # :694
        if multi_label:                                            # :697-699
            ai, ci = (rows[:, 5:] > conf_thres).nonzero(as_tuple=False).T
            det = torch.cat((corners[ai], rows[ai, ci + 5, None], ci[:, None].float()), 1)
        else:                                                      # :701-702 (ties -> first class)
            best, ci = rows[:, 5:].max(1, keepdim=True)
            det = torch.cat((corners, best, ci.float()), 1)[best.view(-1) > conf_thres]
        if classes is not None:                                    # :705-706
            det = det[(det[:, 5:6] == torch.tensor(classes, device=det.device)).any(1)]
        n = det.shape[0]
        if n == 0:
            continue
        if n > MAX_NMS:                                            # :716-717
            det = det[stable_desc_order(det[:, 4])[:MAX_NMS]]
        shift = det[:, 5:6] * (0 if agnostic else MAX_WH)          # :720
        shifted, scores = det[:, :4] + shift, det[:, 4]            # :721
        keep = nms_fn(shifted, scores, iou_thres)                  # :722
        if keep.shape[0] > max_det:                                # :723-724
            keep = keep[:max_det]
        if merge and (1 < n < 3E3):                                # :725 (n is the count before the max_nms cut)
            hit = box_iou(shifted[keep], shifted) > iou_thres      # :727
            weights = hit * scores[None]                           # :728
            det[keep, :4] = torch.mm(weights, det[:, :4]).float() / weights.sum(1, keepdim=True)   # :729
            keep = keep[hit.sum(1) > 1]                            # :730-731 (redundant = True, :666)
        results[img] = det[keep]
    return results
