"""Oracle: IoU family, pairwise IoU and box-format helpers (CPU, test-only).

Restates
  object_detection/yolov5/utils/metrics.py:201-296  (bbox_iou, box_iou, bbox_ioa, wh_iou)
  object_detection/loss.py:102-166                  (IoULossWithMask, bbox_iou on [...,4])
  object_detection/yolov5/utils/general.py:543-560,619-644 (xyxy2xywh, xywh2xyxy, scale_coords, clip_coords)
Operation order is kept identical to the reference so fp32 CPU results match bitwise.
"""
import math

import numpy as np
import torch

MODES = ("iou", "giou", "diou", "ciou")


def mode_of(GIoU=False, DIoU=False, CIoU=False):
    """Flag precedence of the reference: CIoU, then DIoU, then GIoU (metrics.py:225-239)."""
    if CIoU:
        return "ciou"
    if DIoU:
        return "diou"
    if GIoU:
        return "giou"
    return "iou"


def _to_corners(c, corners_given):
    """c = (c0,c1,c2,c3) component tensors. xywh -> x1,y1,x2,y2 as metrics.py:210-213."""
    if corners_given:
        return c[0], c[1], c[2], c[3]
    x1, x2 = c[0] - c[2] / 2, c[0] + c[2] / 2
    y1, y2 = c[1] - c[3] / 2, c[1] + c[3] / 2
    return x1, y1, x2, y2


def iou_components(p, t, corners_given, mode, eps=1e-7):
    """Element-wise IoU-family value between component tuples p and t.

    metrics.py:216-240 == loss.py:138-166.  eps placement: h=y2-y1+eps, union+eps,
    c2+eps, c_area+eps; CIoU's alpha is a constant for autograd (no_grad block).
    """
    ax1, ay1, ax2, ay2 = _to_corners(p, corners_given)
    bx1, by1, bx2, by2 = _to_corners(t, corners_given)

    iw = (torch.min(ax2, bx2) - torch.max(ax1, bx1)).clamp(0)
    ih = (torch.min(ay2, by2) - torch.max(ay1, by1)).clamp(0)
    inter = iw * ih

    w1, h1 = ax2 - ax1, ay2 - ay1 + eps
    w2, h2 = bx2 - bx1, by2 - by1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if mode == "iou":
        return iou

    cw = torch.max(ax2, bx2) - torch.min(ax1, bx1)
    ch = torch.max(ay2, by2) - torch.min(ay1, by1)
    if mode == "giou":
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area

    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((bx1 + bx2 - ax1 - ax2) ** 2 + (by1 + by2 - ay1 - ay2) ** 2) / 4
    if mode == "diou":
        return iou - rho2 / c2

    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def bbox_iou_yolo(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """metrics.py:201-240: box1 is (4,n), box2 is (n,4)."""
    b2 = box2.T
    return iou_components(tuple(box1[k] for k in range(4)), tuple(b2[k] for k in range(4)),
                          x1y1x2y2, mode_of(GIoU, DIoU, CIoU), eps)


def bbox_iou_lastdim(box1, box2, x1y1x2y2=False, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """object_detection/loss.py:118-166: boxes on the last dimension, xywh by default."""
    return iou_components(tuple(box1[..., k] for k in range(4)), tuple(box2[..., k] for k in range(4)),
                          x1y1x2y2, mode_of(GIoU, DIoU, CIoU), eps)


def iou_loss_with_mask(pred, target, mask):
    """object_detection/loss.py:109-115: always CIoU; divisor is sum of the 4x repeated mask + 1e-4."""
    p = pred.permute(0, 2, 3, 1)
    t = target.permute(0, 2, 3, 1)
    m4 = mask.unsqueeze(-1).repeat(1, 1, 1, 4)
    val = bbox_iou_lastdim(p * m4, t * m4, CIoU=True)
    return 1 - val.sum() / (m4.sum() + 1e-4)


def box_iou(box1, box2):
    """metrics.py:242-264: N x M IoU of xyxy boxes, no eps."""
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    rb = torch.min(box1[:, None, 2:], box2[:, 2:])
    lt = torch.max(box1[:, None, :2], box2[:, :2])
    inter = (rb - lt).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def wh_iou(wh1, wh2):
    """metrics.py:291-296."""
    a = wh1[:, None]
    b = wh2[None]
    inter = torch.min(a, b).prod(2)
    return inter / (a.prod(2) + b.prod(2) - inter)


def bbox_ioa(box1, box2, eps=1e-7):
    """metrics.py:267-288 (numpy): intersection over area of box2."""
    b2 = box2.transpose()
    iw = (np.minimum(box1[2], b2[2]) - np.maximum(box1[0], b2[0])).clip(0)
    ih = (np.minimum(box1[3], b2[3]) - np.maximum(box1[1], b2[1])).clip(0)
    area2 = (b2[2] - b2[0]) * (b2[3] - b2[1]) + eps
    return iw * ih / area2


def _dup(x):
    return x.clone() if isinstance(x, torch.Tensor) else np.copy(x)


def xyxy2xywh(x):
    """general.py:543-550."""
    y = _dup(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def xywh2xyxy(x):
    """general.py:553-560."""
    y = _dup(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def clip_coords(boxes, shape):
    """general.py:635-644: in place, shape=(h,w)."""
    if isinstance(boxes, torch.Tensor):
        for col, hi in ((0, shape[1]), (1, shape[0]), (2, shape[1]), (3, shape[0])):
            boxes[:, col].clamp_(0, hi)
    else:
        boxes[:, [0, 2]] = boxes[:, [0, 2]].clip(0, shape[1])
        boxes[:, [1, 3]] = boxes[:, [1, 3]].clip(0, shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """general.py:619-632: undo letterbox pad + gain in place, clip, return coords."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords
