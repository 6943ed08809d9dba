#!/usr/bin/env python3
"""A few validation-threshold NMS calls (conf 0.001, IoU 0.6, multi-label; bs 64 @ 640 px) on a decoded tensor, on the
raw maps and on the convolution outputs, and nothing else: the program to put behind `rocprofv3 --pmc ... --` when the
counters of the candidate-search kernels are wanted."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from object_detection_amd.general import conv_nms, detect_nms, non_max_suppression  # noqa: E402
from object_detection_amd.yolo import detect_decode  # noqa: E402
from object_detection_amd.yolo_loss import ComputeLoss  # noqa: E402

dev = torch.device("cuda", 0)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
anchors = bench.grid_anchors(dev)
crit = ComputeLoss(bench._Model(anchors, 80))
levels, targets = bench.synth_batch(bs, 640, 80, 10 * bs, 1000, dev)
bench.plant_objects(levels, targets, crit)
strides = torch.tensor([8., 16., 32.])
z = detect_decode(levels, anchors, strides)
convs = [x.permute(0, 1, 4, 2, 3).reshape(bs, 255, x.shape[2], x.shape[3]).contiguous() for x in levels]
kw = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)
for _ in range(4):
    non_max_suppression(z, **kw)
    detect_nms(levels, anchors, strides, **kw)
    conv_nms(convs, 3, anchors, strides, **kw)
torch.cuda.synchronize()
