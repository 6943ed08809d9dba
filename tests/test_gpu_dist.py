"""The DDP train-step configuration on the device: the loss kernels' own statistics vector goes through an RCCL
all-reduce on a side stream (one-rank group on the one-GPU box: the process group, the device tensor, the stream
ordering and the collective are the real ones; what a second rank would add is covered on gloo in test_dist_cpu.py).
Launch being mirrored: object_detection/yolov5/train.py:522-530 (process group), :254-255 (DDP), :331-332 (loss scaling)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from oracle import yolo as oy
from tests._synth import synth_head, synth_targets
from tests.test_gpu_yolo_loss import T, close, make_loss

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))      # "nccl" is RCCL on ROCm
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_loss_statistics_all_reduce_on_a_side_stream(rccl_group):
    from object_detection_amd.dist import all_reduce_loss_stats, shard_targets
    rs = np.random.RandomState(4004)
    bs, nc, world = 16, 80, 2
    anchors = oy.grid_anchors()
    levels = synth_head(rs, bs, 320, nc)
    targets = synth_targets(rs, 160, bs)
    hyp = dict(oy.HYP_SCRATCH)
    # this rank's shard of a global batch (second half: image indices are re-based)
    mine = shard_targets(T(targets), 1, world, bs)
    half = [x[bs // 2:] for x in levels]
    want_fn = oy.YoloLossOracle(anchors, hyp, nc=nc)
    wl, wi = want_fn([T(x) for x in half], mine)
    n_pos = sum(len(t["b"]) for t in want_fn.targets_for([T(x) for x in half], mine))

    crit = make_loss(anchors, nc, hyp)
    d = [T(x).cuda() for x in half]
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    for step in range(3):         # the bench's ordering: loss on the main stream, the collective on the side stream
        loss, items, grads = crit.forward_backward(d, mine.cuda())
        stats = crit.last_stats
        side.wait_stream(main)
        with torch.cuda.stream(side):
            stats.record_stream(side)
            reduced = all_reduce_loss_stats(stats)
        main.wait_stream(side)
    torch.cuda.synchronize()
    assert reduced.is_cuda and reduced.shape == (4,)
    close(reduced[:3], wi)                                               # mean over one rank = its own loss terms
    assert float(reduced[3]) == float(n_pos) == float(crit.last_positive_counts.sum())
    close(loss, wl.detach())
    # the packed form (loss items + per-level counts) gives the same vector
    torch.testing.assert_close(all_reduce_loss_stats(items, crit.last_positive_counts), reduced, rtol=0, atol=0)
    # the reference's scaling of the loss by the world size is the caller's (train.py:331-332): gradients stay per-rank
    assert all(g.shape == x.shape for g, x in zip(grads, d))
