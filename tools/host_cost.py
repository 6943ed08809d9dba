#!/usr/bin/env python3
"""Host-side cost (microseconds per call) of every piece of one step, measured with a tiny batch so that the
GPU never back-pressures the host: what the Python + HIP-runtime enqueue path costs by itself."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
from object_detection_amd import _lib as L  # noqa: E402
from object_detection_amd import yolo_loss as YL  # noqa: E402
from object_detection_amd.general import nms_launch, nms_meta_to_host, nms_views  # noqa: E402
from object_detection_amd.yolo import detect_decode  # noqa: E402
from object_detection_amd.yolo_loss import ComputeLoss  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = 300
anchors = bench.grid_anchors(dev)
crit = ComputeLoss(bench._Model(anchors, 80))
levels, targets = bench.synth_batch(bs, 640, 80, 10 * bs, 1000, dev)
bench.plant_objects(levels, targets, crit)
strides = torch.tensor([8., 16., 32.])
p = [x.requires_grad_(True) for x in levels]
if len(sys.argv) > 2:
    torch.cuda.set_stream(torch.cuda.Stream())


def cost(name, fn, n=N):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f"{name:34s} {dt:8.2f} us", flush=True)
    return dt


with torch.no_grad():
    z = detect_decode(levels, anchors, strides)
shapes = [(x.shape[2], x.shape[3]) for x in p]
hyp = crit._hyp_struct()
maps = [x.detach() for x in p]
assign = YL._assign(shapes, targets, crit.anchors, 4.0)
out, ws = YL._loss_fwd(assign, hyp, 80, maps)
one = torch.ones(1, device=dev)
rows, meta, cap = nms_launch(z, 0.25, 0.45, max_det=300, cap=25200)
host, evd = nms_meta_to_host(meta)
torch.cuda.synchronize()
hl = host.tolist()
s2 = torch.cuda.Stream()
e = torch.cuda.Event()

cost("torch.empty(8)", lambda: torch.empty(8, device=dev))
cost("torch.empty_like(level0)", lambda: torch.empty_like(levels[0]))
cost("Event()+record", lambda: torch.cuda.Event().record())
cost("Event(timing)+record", lambda: torch.cuda.Event(enable_timing=True).record())
cost("stream.wait_event", lambda: s2.wait_event(e))
cost("set_stream x2", lambda: (torch.cuda.set_stream(s2), torch.cuda.set_stream(torch.cuda.default_stream())))
cost("L.stream()", L.stream)
cost("dh_nms_workspace (ctypes, no GPU)", lambda: L.lib().dh_nms_workspace(bs, 80, 25200, 30000))
cost("detect_decode", lambda: detect_decode(levels, anchors, strides))
cost("_assign", lambda: YL._assign(shapes, targets, crit.anchors, 4.0))
cost("_hyp_struct", crit._hyp_struct)
cost("_loss_fwd", lambda: YL._loss_fwd(assign, hyp, 80, maps))
cost("_loss_bwd", lambda: YL._loss_bwd(assign, hyp, 80, maps, ws, one))
cost("crit(p,t)", lambda: crit(p, targets))


def fb():
    loss, _ = crit(p, targets)
    torch.autograd.grad(loss, p)


cost("crit + autograd.grad", fb)
cost("crit.forward_backward", lambda: crit.forward_backward(p, targets))
cost("nms_launch", lambda: nms_launch(z, 0.25, 0.45, max_det=300, cap=25200))
cost("nms_meta_to_host", lambda: nms_meta_to_host(meta))
cost("nms_views", lambda: nms_views(rows, hl, cap))
