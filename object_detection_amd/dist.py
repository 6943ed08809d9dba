"""The one collective of the data-parallel path: a single small all-reduce per training step.

Batches shard by image (one process per GPU, contiguous bs/W images each, as DistributedSampler
does for the reference, object_detection/yolov5/utils/datasets.py:114).  The reference does not
exchange loss values between ranks: each rank normalises by its local positives and multiplies
the loss by WORLD_SIZE (object_detection/yolov5/train.py:331-332) while DDP averages gradients.
That behaviour is kept for the gradients by default.  What travels between ranks is one all_reduce(SUM) over RCCL ("nccl"
backend on ROCm) of the loss kernels' un-normalised sums and their divisors (`ComputeLoss.last_sums`, 5 doubles per level):
  global_loss_stats       the loss terms of the WHOLE batch, for reporting -- sums divided by summed divisors, which a mean
                          of per-rank means only equals when every rank holds the same number of assigned rows;
  all_reduce_loss_stats   the older form, the per-rank means averaged over the ranks (what the reference's ranks would log,
                          averaged), kept for callers that hold only the loss items;
  ComputeLoss.forward_backward_global (yolo_loss.py) uses the same sums between forward and backward to normalise the
                          gradients over the whole batch as well (opt-in).
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


def all_reduce_loss_sums(sums, group=None):
    """Sum over the ranks of a (nl, 5) float64 block of loss sums [sum(1-iou), sum class BCE, sum objectness BCE, rows,
    cells per level]; the input is left as it is.  A host-side backend (gloo rehearsal) takes the block through the host."""
    total = sums.detach().clone()
    if not (dist.is_available() and dist.is_initialized()):
        return total
    if total.is_cuda and dist.get_backend(group) == "gloo":
        host = total.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        return host.to(total.device)
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total


def global_loss_stats(crit, group=None):
    """[lbox, lobj, lcls, n_pos] of the whole data-parallel batch from the last call of `crit` on every rank: one
    all-reduce of crit.last_sums, then the divisions (ComputeLoss.items_from_sums)."""
    return crit.items_from_sums(all_reduce_loss_sums(crit.last_sums, group))


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
