"""Drop-ins for the CenterNet head glue.

  object_detection/detectors/centernet/loss.py:4-28        offset_to_coord, compute_losses
"""
import torch

from . import _lib as L
from .loss import IoULossWithMask, masked_ciou_loss


def offset_to_coord(pred):
    """pred (B,2,W,H): plane 0 += index along dim 2, plane 1 += index along dim 3 (loss.py:4-13)."""
    p = L.dev_f32(pred, "pred")
    _, _, dim_w, dim_h = p.shape
    ix = torch.arange(dim_w, device=p.device, dtype=torch.float32).view(1, 1, dim_w, 1)
    iy = torch.arange(dim_h, device=p.device, dtype=torch.float32).view(1, 1, 1, dim_h)
    return p + torch.cat([ix.expand(1, 1, dim_w, dim_h), iy.expand(1, 1, dim_w, dim_h)], dim=1)


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
