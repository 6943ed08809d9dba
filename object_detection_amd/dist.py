"""The one collective of the data-parallel path: a single small all-reduce per training step.

Batches shard by image (one process per GPU, contiguous bs/W images each, as DistributedSampler
does for the reference, object_detection/yolov5/utils/datasets.py:114).  The reference does not
exchange loss values between ranks: each rank normalises by its local positives and multiplies
the loss by WORLD_SIZE (object_detection/yolov5/train.py:331-332) while DDP averages gradients.
That behaviour is kept for the gradients; this helper only makes the *reported* loss terms and the
assigned-positive count global, with one all_reduce(SUM) over RCCL ("nccl" backend on ROCm).
"""
import torch
import torch.distributed as dist


def pack_loss_stats(loss_items, positive_counts):
    """[lbox, lobj, lcls, n_pos] as one fp32 tensor on the device of loss_items (no host copy)."""
    n_pos = positive_counts.sum().to(torch.float32).reshape(1)
    return torch.cat((loss_items.detach().reshape(3).float(), n_pos.to(loss_items.device)))


def all_reduce_loss_stats(loss_items, positive_counts=None, group=None):
    """Global means of (lbox, lobj, lcls) and the global number of assigned positives.

    loss_items: the 3 loss terms with positive_counts (per-level rows), or the ready-made 4-element
    [lbox, lobj, lcls, n_pos] that ComputeLoss leaves in `.last_stats` (then positive_counts is None).
    Returns a 4-element tensor [mean lbox, mean lobj, mean lcls, sum n_pos]; with no initialised
    process group (single process) the local values are returned unchanged.
    """
    stats = loss_items.detach().clone() if positive_counts is None else pack_loss_stats(loss_items, positive_counts)
    if not (dist.is_available() and dist.is_initialized()):
        return stats
    world = dist.get_world_size(group)
    dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    stats[:3] /= world
    return stats


def shard_targets(targets, rank, world_size, batch_size):
    """Rows of a global (nt,6) target tensor that belong to this rank's contiguous image shard,
    with the image index re-based to the shard (what a per-rank collate_fn would have produced,
    object_detection/yolov5/utils/datasets.py:627-632)."""
    if batch_size % world_size:
        raise ValueError(f"batch size {batch_size} is not divisible by world size {world_size}")
    per = batch_size // world_size
    lo, hi = rank * per, (rank + 1) * per
    img = targets[:, 0]
    mine = targets[(img >= lo) & (img < hi)].clone()
    mine[:, 0] -= lo
    return mine
