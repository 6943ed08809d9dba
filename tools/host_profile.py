#!/usr/bin/env python3
"""cProfile of the host-side enqueue cost of one bench step (GPU work is asynchronous)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
from object_detection_amd.general import nms_launch, nms_meta_to_host, nms_views  # noqa: E402
from object_detection_amd.yolo import detect_decode  # noqa: E402
from object_detection_amd.yolo_loss import ComputeLoss  # noqa: E402

bs = 64
anchors = bench.grid_anchors(dev)
crit = ComputeLoss(bench._Model(anchors, 80))
levels, targets = bench.synth_batch(bs, 640, 80, 10 * bs, 1000, dev)
bench.plant_objects(levels, targets, crit)
strides = torch.tensor([8., 16., 32.])
p = [x.requires_grad_(True) for x in levels]


def step():
    with torch.no_grad():
        z = detect_decode(levels, anchors, strides)
    loss, items = crit(p, targets)
    grads = torch.autograd.grad(loss, p)
    rows, meta, cap = nms_launch(z, 0.25, 0.45, max_det=300, cap=25200)
    host, ev = nms_meta_to_host(meta)
    ev.synchronize()
    return nms_views(rows, host.tolist(), cap)


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats("object_detection_amd|bench|host_profile")
