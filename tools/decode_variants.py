#!/usr/bin/env python3
"""Isolated decode timings through the C ABI: no planes / objectness plane / objectness + logit planes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from object_detection_amd import _lib as L  # noqa: E402
from tools.time_kernels import timed  # noqa: E402

dev = torch.device("cuda", 0)
bs, A, no = 64, 25200, 85
levels, _ = bench.synth_batch(bs, 640, 80, 640, 1000, dev)
out = torch.empty((bs, A, no), device=dev)
planes = torch.empty((2, bs, A), device=dev)
ptrs = (L.c_void_p * 3)(*[x.data_ptr() for x in levels])
ny = L.int_array([80, 40, 20])
apx = L.float_array([v for row in bench.ANCHORS_PX for v in row])
st = L.float_array([8, 16, 32])


def run(p0, p1):
    L.check(L.lib().dh_yolo_decode(ptrs, ny, ny, 3, bs, 3, no, apx, st, L.ptr(out), p0, p1, L.stream()), "decode")


null = L.c_void_p(0)
for name, a, b in (("no planes", null, null), ("obj plane", L.ptr(planes[0]), null), ("obj + logit planes", L.ptr(planes[0]), L.ptr(planes[1]))):
    us = timed(lambda: run(a, b), 50)
    print(f"{name:22s} {us:8.1f} us  {680.0 * A * bs / us / 1e3:8.1f} GB/s", flush=True)
src = torch.empty(bs * A * no, device=dev)
us = timed(lambda: out.view(-1).copy_(src), 50)
print(f"{'torch copy':22s} {us:8.1f} us  {680.0 * A * bs / us / 1e3:8.1f} GB/s")
