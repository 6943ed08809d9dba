"""Oracle: CenterNet decode / post-process / target drawing / box-loss glue (CPU, test-only).

Restates
  object_detection/detectors/centernet/dataset.py:96-161  (target maps; first spatial axis is x)
  object_detection/detectors/centernet/dataset.py:164-297 (pool_nms, decode_bbox, postprocess...)
  object_detection/detectors/centernet/loss.py:4-28       (offset_to_coord, compute_losses)
"""
import numpy as np
import torch
import torch.nn.functional as F

from .boxes import iou_loss_with_mask
from .nms import nms as _nms


def pool_nms(heat, kernel=3):
    """dataset.py:164-169: keep a cell only where it equals its kxk neighbourhood max."""
    peak = F.max_pool2d(heat, (kernel, kernel), stride=1, padding=(kernel - 1) // 2)
    return heat * (peak == heat).float()


def decode_bbox(pred_cls, pred_size, pred_offset, confidence, device=None):
    """dataset.py:172-216 -> per image either [] or (k,6) [x1,y1,x2,y2,conf,cls] normalised."""
    heat = pool_nms(pred_cls)
    nb, nc, dim_w, dim_h = heat.shape
    ix, iy = torch.meshgrid(torch.arange(0, dim_w), torch.arange(0, dim_h), indexing="ij")
    ix, iy = ix.flatten().float(), iy.flatten().float()
    found = []
    for k in range(nb):
        score = torch.tanh(heat[k]).permute(1, 2, 0).reshape(-1, nc)
        wh = pred_size[k].permute(1, 2, 0).reshape(-1, 2)
        off = pred_offset[k].permute(1, 2, 0).reshape(-1, 2)
        conf, cls = torch.max(score, dim=-1)
        sel = conf > confidence
        if int(sel.sum()) == 0:
            found.append([])
            continue
        cx = (ix[sel] + off[sel][:, 0]).unsqueeze(-1)
        cy = (iy[sel] + off[sel][:, 1]).unsqueeze(-1)
        hw, hh = wh[sel][:, 0:1] / 2, wh[sel][:, 1:2] / 2
        box = torch.cat([cx - hw, cy - hh, cx + hw, cy + hh], dim=1)
        box[:, [0, 2]] /= dim_w
        box[:, [1, 3]] /= dim_h
        found.append(torch.cat([box, conf[sel].unsqueeze(-1), cls[sel].unsqueeze(-1).float()], dim=-1))
    return found


def correct_boxes(box_xy, box_wh, input_shape, image_shape, letterbox_image):
    """dataset.py:219-243 (numpy)."""
    input_shape = np.array(input_shape)
    image_shape = np.array(image_shape)
    if letterbox_image:
        new_shape = np.round(image_shape * np.min(input_shape / image_shape))
        offset = (input_shape - new_shape) / 2.0 / input_shape
        scale = input_shape / new_shape
        box_xy = (box_xy - offset) * scale
        box_wh *= scale
    lo = box_xy - (box_wh / 2.0)
    hi = box_xy + (box_wh / 2.0)
    boxes = np.concatenate([lo[..., 0:1], lo[..., 1:2], hi[..., 0:1], hi[..., 1:2]], axis=-1)
    boxes *= image_shape
    return boxes


def postprocess(prediction, need_nms, image_shape, input_shape, letterbox_image, nms_thres=0.4):
    """dataset.py:246-284: per image, per ascending unique class, optional NMS, concat, rescale."""
    out = [None] * len(prediction)
    for k, det in enumerate(prediction):
        if len(det) == 0:
            continue
        for c in det[:, -1].unique():
            same = det[det[:, -1] == c]
            if need_nms:
                same = same[_nms(same[:, :4], same[:, 4], nms_thres)]
            out[k] = same if out[k] is None else torch.cat((out[k], same))
        if out[k] is not None:
            arr = out[k].detach().numpy()
            xy, wh = (arr[:, 0:2] + arr[:, 2:4]) / 2, arr[:, 2:4] - arr[:, 0:2]
            arr[:, :4] = correct_boxes(xy, wh, input_shape, image_shape, letterbox_image)
            out[k] = arr
    return out


def postprocess_predictions(pred, image_size, device=None):
    """dataset.py:287-297: confidence 0.9, no NMS, input_shape = image_size/4."""
    found = decode_bbox(pred[0], pred[1], pred[2], confidence=0.9)
    return postprocess(found, need_nms=False, image_shape=image_size, input_shape=image_size / 4,
                       letterbox_image=False, nms_thres=0.4)


def draw_targets(labels, num_classes, input_shape, output_shape):
    """dataset.py:103-150 for one image.

    labels: (n,5) [x1,y1,x2,y2,cls] in input pixels.  Returns CHW fp32 maps
    (cls (C,W,H), size (2,W,H), offset (2,W,H), mask (W,H)); axis 0 of the maps is x.
    Later labels overwrite earlier ones on the same cell; w<=0 or h<=0 labels are skipped.
    """
    ow, oh = output_shape
    t_cls = np.zeros((ow, oh, num_classes), dtype=np.float32)
    t_size = np.zeros((ow, oh, 2), dtype=np.float32)
    t_off = np.zeros((ow, oh, 2), dtype=np.float32)
    t_mask = np.zeros((ow, oh), dtype=np.float32)
    labels = np.asarray(labels)
    if len(labels):
        box = np.array(labels[:, :4], dtype=np.float32)
        box[:, [0, 2]] = np.clip(box[:, [0, 2]] / input_shape[0] * ow, 0, ow - 1)
        box[:, [1, 3]] = np.clip(box[:, [1, 3]] / input_shape[1] * oh, 0, oh - 1)
        for k in range(len(labels)):
            x1, y1, x2, y2 = box[k]
            w, h = x2 - x1, y2 - y1
            if not (w > 0 and h > 0):
                continue
            ct = np.array([(x1 + x2) / 2, (y1 + y2) / 2], dtype=np.float32)
            ci = ct.astype(np.int32)
            t_cls[ci[0], ci[1], int(labels[k, -1])] = 1          # dataset.py:57-60 live effect
            t_size[ci[0], ci[1]] = 1.0 * w, 1.0 * h
            t_off[ci[0], ci[1]] = ct - ci
            t_mask[ci[0], ci[1]] = 1
    return (np.transpose(t_cls, (2, 0, 1)), np.transpose(t_size, (2, 0, 1)),
            np.transpose(t_off, (2, 0, 1)), t_mask)


def offset_to_coord(pred):
    """loss.py:4-13: channel 0 += index along dim 2, channel 1 += index along dim 3."""
    b, _, dim_w, dim_h = pred.shape
    ix, iy = torch.meshgrid(torch.arange(0, dim_w), torch.arange(0, dim_h), indexing="ij")
    return pred + torch.stack([ix, iy]).repeat(b, 1, 1, 1)


def box_loss(pred, labels):
    """The criterion["box"] branch of compute_losses (loss.py:22-28) with IoULossWithMask."""
    return iou_loss_with_mask(torch.cat([offset_to_coord(pred[2]), pred[1]], dim=1),
                              torch.cat([offset_to_coord(labels[2]), labels[1]], dim=1), labels[3])
