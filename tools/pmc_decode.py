#!/usr/bin/env python3
"""A few decode launches (fp32 maps, then fp16 maps) and nothing else: the program to put behind
`rocprofv3 --pmc ... --` when the decode kernel's issue / wait counters are wanted."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from object_detection_amd.yolo import detect_decode  # noqa: E402

dev = torch.device("cuda", 0)
anchors = bench.grid_anchors(dev)
levels, _ = bench.synth_batch(64, 640, 80, 640, 1000, dev)
strides = torch.tensor([8., 16., 32.])
half = [x.half() for x in levels]
for _ in range(5):
    detect_decode(levels, anchors, strides)
for _ in range(5):
    detect_decode(half, anchors, strides)
torch.cuda.synchronize()
