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
from tests._checks import all_grads_close, loss_grads64
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
    p = [T(x).requires_grad_(True) for x in half]
    wl2, _ = want_fn(p, mine)
    all_grads_close(grads, torch.autograd.grad(wl2, p), loss_grads64(want_fn, half, mine))
    # the sum form: per-level sums and divisors, reduced, give the same items (one rank: its own)
    from object_detection_amd.dist import global_loss_stats
    torch.testing.assert_close(crit.last_sums.cpu(), want_fn.last_sums, rtol=1e-5, atol=1e-9)
    g = global_loss_stats(crit)
    close(g[:3], wi)
    assert float(g[3]) == float(n_pos)


def test_global_normalisation_without_a_group_is_the_plain_call():
    """forward_backward_global in a single process: every factor is 1, so loss, items and gradients are forward_backward's
    (the gradients bit for bit: a multiplication by 1.0f)."""
    rs = np.random.RandomState(4005)
    anchors = oy.grid_anchors()
    d = [T(x).cuda() for x in synth_head(rs, 4, 320, 80)]
    tg = T(synth_targets(rs, 40, 4)).cuda()
    crit = make_loss(anchors, 80, dict(oy.HYP_SCRATCH))
    l0, i0, g0 = crit.forward_backward(d, tg)
    l1, i1, g1 = crit.forward_backward_global(d, tg)
    close(l1, l0)
    close(i1[:3], i0)
    for a, b in zip(g1, g0):
        assert torch.equal(a, b)


def _spawn(cmds, env=None, timeout=240):
    import subprocess
    procs = [subprocess.Popen(c, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for c in cmds]
    outs = []
    try:
        for pr in procs:
            out, err = pr.communicate(timeout=timeout)
            outs.append((pr.returncode, out, err))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return outs


def test_two_ranks_sharing_the_device_global_normalisation(tmp_path):
    """Two fresh rank processes on this one GPU (gloo between them), very unequal numbers of assigned rows: with
    forward_backward_global every rank's gradient is the gradient of the loss of the WHOLE batch with respect to its own
    maps, its items are the whole batch's; the default call keeps the reference's per-rank normalisation."""
    import sys
    from tests.test_dist_cpu import _free_port, _unequal_batch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(_free_port())
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    outs = _spawn([[sys.executable, "-m", "tests._dist_rank", str(r), "2", port, str(tmp_path)] for r in range(2)], env)
    for rc, out, err in outs:
        assert rc == 0, err[-3000:]
    r = [torch.load(tmp_path / f"g{k}.pt") for k in range(2)]
    levels, targets, bs = _unequal_batch()
    whole = oy.YoloLossOracle(oy.grid_anchors(), dict(oy.HYP_SCRATCH), nc=20)
    p = [T(x).requires_grad_(True) for x in levels]
    wl, wi = whole(p, T(targets))
    wg = torch.autograd.grad(wl, p)
    assert r[0]["rows"] > 4 * r[1]["rows"] > 0
    for k in range(2):
        close(r[k]["items"][:3], wi, rtol=1e-5)
        close(r[k]["items_conv"][:3], wi, rtol=1e-5)
        close(r[k]["reported"][:3], wi, rtol=1e-5)
        assert float(r[k]["items"][3]) == float(r[k]["reported"][3]) == r[0]["rows"] + r[1]["rows"]
        close(r[k]["loss"], wl.detach() / 2, rtol=1e-5)               # the reference's `loss *= WORLD_SIZE` gives the whole loss
    w64 = loss_grads64(whole, levels, targets)
    all_grads_close([torch.cat((a, b)) for a, b in zip(r[0]["grads"], r[1]["grads"])], wg, w64)
    all_grads_close([torch.cat((a, b)) for a, b in zip(r[0]["grads_conv"], r[1]["grads_conv"])], wg, w64)
    # default call: per-rank normalisation, i.e. the oracle on the shard alone
    per = bs // 2
    from object_detection_amd.dist import shard_targets
    for k in range(2):
        shard = [x[k * per:(k + 1) * per] for x in levels]
        mine = shard_targets(T(targets), k, 2, bs)
        o = oy.YoloLossOracle(oy.grid_anchors(), dict(oy.HYP_SCRATCH), nc=20)
        q = [T(np.ascontiguousarray(x)).requires_grad_(True) for x in shard]
        ll, li = o(q, mine)
        close(r[k]["local_items"], li)
        all_grads_close(r[k]["local_grads"], torch.autograd.grad(ll, q), loss_grads64(o, shard, mine))


def test_bench_two_ranks_sharing_the_device():
    """`python bench.py --gpus 2` end to end on the one-GPU box: two fresh rank processes on the same device, gloo between
    them (RCCL refuses two ranks on one GPU); the launch path, per-rank seeds, barrier + max-over-ranks timing and the
    per-step collective are the ones of a real multi-GPU run."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DH_BENCH_SHARE_DEVICE="1", DH_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    (rc, out, err), = _spawn([[sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                               "--batch", "8", "--pre-roll", "0", "--post-roll", "0"]], env)
    assert rc == 0, err[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5
    assert len(line["ms_per_step_per_rank"]) == 2 and line["ms_per_step"] == pytest.approx(max(line["ms_per_step_per_rank"]), rel=1e-3)
    cc = line["collective_check"]
    assert cc["ok"] and len(cc["reduced"]) == 15
    assert cc["global_lbox_lobj_lcls_npos"][3] > cc["this_rank_lbox_lobj_lcls_npos"][3] > 0
    assert line["value"] == pytest.approx(2 * 8 * 5 / (line["ms_per_step"] * 5e-3), rel=1e-3)
