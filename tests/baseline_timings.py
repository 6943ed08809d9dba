#!/usr/bin/env python3
"""CPU / torch-operation baselines for rows that bench.py does not time, kept under tests/ because they run the oracle
(test infrastructure: the restatement of the reference's algorithm).  Not collected by pytest.

  python tests/baseline_timings.py ap      ap_per_class of a 5000-image evaluation run (1.5 M detections x 10 IoU thresholds, 80
                                           classes): the numpy restatement on the host against the device path, results compared
  python tests/baseline_timings.py focal   LabelSmoothingFocalLoss forward + backward at bs 64 x 80 x 160 x 160 as the reference's
                                           torch operations on the GPU against the kernels"""
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def ap(dev):
    from object_detection_amd.metrics import ap_per_class
    from oracle import ap as oa
    rs = np.random.RandomState(5)
    n, nc, m, niou = 1_500_000, 80, 36_000, 10
    target_cls = rs.randint(0, nc, m).astype(np.float64)
    pred_cls = rs.randint(0, nc, n).astype(np.float32)
    conf = rs.uniform(0.001, 1, n).astype(np.float32)
    depth = np.where(rs.rand(n) < 0.02 * (0.2 + conf), rs.randint(1, niou + 1, n), 0)
    tp = np.arange(niou)[None, :] < depth[:, None]
    host = (tp, conf, pred_cls, target_cls)
    devs = [torch.from_numpy(a).to(dev) for a in host]
    got = ap_per_class(*devs)
    ms_d = timed(lambda: ap_per_class(*devs), 5)
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oa.ap_per_class(*host)
    ms_c = (time.perf_counter() - t0) * 1e3
    same = all(np.array_equal(g, w) for g, w in zip(got, want))
    print(f"ap n {n} x {niou}, {nc} classes: device {ms_d:8.3f} ms, numpy restatement on the host {ms_c:8.1f} ms; "
          f"results {'identical' if same else 'DIFFER'}", flush=True)


def focal(dev):
    from object_detection_amd.loss import LabelSmoothingFocalLoss
    from oracle import focal as of
    g = torch.Generator(device=dev).manual_seed(5)
    logits = torch.randn((64, 80, 160, 160), generator=g, device=dev).requires_grad_(True)
    label = torch.zeros_like(logits)
    idx = [torch.randint(0, d, (200 * 64,), generator=g, device=dev) for d in logits.shape]
    label[idx[0], idx[1], idx[2], idx[3]] = 1.0
    kw = dict(num_classes=80, one_hot_label_format=True, gamma=8.0, smoothing=0.1)
    mine, ref = LabelSmoothingFocalLoss(**kw), of.LabelSmoothingFocalLoss(**kw)
    ms_k = timed(lambda: torch.autograd.grad(mine(logits, label), logits), 20)
    ms_r = timed(lambda: torch.autograd.grad(ref(logits, label), logits), 5)
    print(f"class loss forward + backward: kernels {ms_k:8.3f} ms, the reference's torch operations on the same device {ms_r:8.3f} ms",
          flush=True)


if __name__ == "__main__":
    device = torch.device("cuda", 0)
    for which in sys.argv[1:] or ["ap", "focal"]:
        {"ap": ap, "focal": focal}[which](device)
