"""One rank of the two-rank rehearsal on a one-GPU box (started by tests/test_gpu_dist.py as a fresh process): both ranks
share cuda:0 and talk over gloo; each evaluates ComputeLoss.forward_backward_global on its image shard of the batch of
tests/test_dist_cpu.py::_unequal_batch and saves what it got.

  python -m tests._dist_rank <rank> <world> <port> <out_dir>
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from object_detection_amd.dist import global_loss_stats, shard_targets
    from oracle import yolo as oy
    from tests.test_dist_cpu import _unequal_batch
    from tests.test_gpu_yolo_loss import make_loss
    levels, targets, bs = _unequal_batch()
    per = bs // world
    mine = shard_targets(torch.from_numpy(targets), rank, world, bs).cuda()
    d = [torch.from_numpy(np.ascontiguousarray(x[rank * per:(rank + 1) * per])).cuda() for x in levels]
    crit = make_loss(oy.grid_anchors(), 20, dict(oy.HYP_SCRATCH))
    loss, items, grads = crit.forward_backward_global(d, mine)
    # the reference's normalisation on the same shard, and the reporting path on top of it
    l_loc, i_loc, g_loc = crit.forward_backward(d, mine)
    reported = global_loss_stats(crit)
    # the same on the convolution layout (lazy maps)
    convs = [x.permute(0, 1, 4, 2, 3).reshape(x.shape[0], -1, x.shape[2], x.shape[3]).contiguous() for x in d]
    views = [c.view(c.shape[0], 3, 25, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2) for c in convs]
    _, items_c, grads_c = crit.forward_backward_global(views, mine)
    torch.cuda.synchronize()
    torch.save({"loss": loss.cpu(), "items": items.cpu(), "grads": [g.cpu() for g in grads], "local_items": i_loc.cpu(),
                "local_grads": [g.cpu() for g in g_loc], "reported": reported.cpu(), "items_conv": items_c.cpu(),
                "grads_conv": [g.contiguous().cpu() for g in grads_c], "rows": float(crit.last_sums[:, 3].sum())},
               os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
