"""Oracle: matching of detections to labels in evaluation (CPU, test-only).

Restates
  object_detection/yolov5/val.py:71-92            (process_batch -> correct[N, n_iou])
  object_detection/yolov5/utils/metrics.py:116-171 (ConfusionMatrix.__init__/process_batch/tp_fp)
Operation order is the reference's.  Its two numpy sorts (`argsort()[::-1]`, default kind = an unstable
introsort) leave the order of equal IoUs to the implementation; here they are pinned to the stable variant
(`argsort(kind="stable")[::-1]`: equal IoUs in descending pair position), which is one of the orders the
reference can produce.  Inputs without equal IoUs are insensitive to that choice; the golden fixtures hold both kinds.
"""
import numpy as np
import torch

from .boxes import box_iou


def _desc(matches):
    return matches[matches[:, 2].argsort(kind="stable")[::-1]]


def process_batch(detections, labels, iouv):
    """val.py:71-92.  detections (N,6) [xyxy, conf, cls]; labels (M,5) [cls, xyxy]; iouv (n_iou,) -> bool (N, n_iou)."""
    correct = torch.zeros(detections.shape[0], iouv.shape[0], dtype=torch.bool)
    iou = box_iou(labels[:, 1:], detections[:, :4])
    x = torch.where((iou >= iouv[0]) & (labels[:, 0:1] == detections[:, 5]))
    if x[0].shape[0]:
        matches = torch.cat((torch.stack(x, 1), iou[x[0], x[1]][:, None]), 1).cpu().numpy()     # [label, detection, iou] float32
        if x[0].shape[0] > 1:
            matches = _desc(matches)
            matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
            matches = matches[np.unique(matches[:, 0], return_index=True)[1]]                  # no re-sort here (val.py:87)
        matches = torch.Tensor(matches)
        correct[matches[:, 1].long()] = matches[:, 2:3] >= iouv
    return correct


class ConfusionMatrix:
    """metrics.py:116-171 without the plotting / printing."""

    def __init__(self, nc, conf=0.25, iou_thres=0.45):
        self.matrix = np.zeros((nc + 1, nc + 1))
        self.nc = nc
        self.conf = conf
        self.iou_thres = iou_thres

    def process_batch(self, detections, labels):
        detections = detections[detections[:, 4] > self.conf]
        gt_classes = labels[:, 0].int()
        detection_classes = detections[:, 5].int()
        iou = box_iou(labels[:, 1:], detections[:, :4])
        x = torch.where(iou > self.iou_thres)
        if x[0].shape[0]:
            matches = torch.cat((torch.stack(x, 1), iou[x[0], x[1]][:, None]), 1).cpu().numpy()
            if x[0].shape[0] > 1:
                matches = _desc(matches)
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = _desc(matches)
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
        else:
            matches = np.zeros((0, 3))
        n = matches.shape[0] > 0
        m0, m1, _ = matches.transpose().astype(np.int16)
        for i, gc in enumerate(gt_classes):
            j = m0 == i
            if n and sum(j) == 1:
                self.matrix[detection_classes[m1[j]], gc] += 1      # correct
            else:
                self.matrix[self.nc, gc] += 1                       # background FP
        if n:
            for i, dc in enumerate(detection_classes):
                if not any(m1 == i):
                    self.matrix[dc, self.nc] += 1                   # background FN

    def tp_fp(self):
        tp = self.matrix.diagonal()
        fp = self.matrix.sum(1) - tp
        return tp[:-1], fp[:-1]
