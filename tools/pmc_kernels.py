#!/usr/bin/env python3
"""A few launches of the kernels whose counters are wanted, at the sizes the design document quotes, and nothing else: the
program to put behind `rocprofv3 --kernel-trace ... --` and `rocprofv3 --pmc ... --`.

  python tools/pmc_kernels.py [pairwise,cnloss,cndecode,head16,convval,iou] [launches]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

which = set((sys.argv[1] if len(sys.argv) > 1 else "pairwise,cnloss,cndecode,head16,convval,iou").split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
bs = 64

if "pairwise" in which:        # metrics.py:242-264 at the validation size, the merge-NMS size and two write-bound sizes
    from object_detection_amd.metrics import box_iou
    g = torch.Generator(device=dev).manual_seed(6)
    for n, m in ((100, 300), (300, 3000), (8192, 8192), (30000, 3000)):
        xy1, xy2 = torch.rand((n, 2), generator=g, device=dev) * 600, torch.rand((m, 2), generator=g, device=dev) * 600
        b1 = torch.cat((xy1, xy1 + torch.rand((n, 2), generator=g, device=dev) * 200 + 1), 1)
        b2 = torch.cat((xy2, xy2 + torch.rand((m, 2), generator=g, device=dev) * 200 + 1), 1)
        out = torch.empty((n, m), device=dev)
        for _ in range(reps):
            box_iou(b1, b2, out=out)
        del out
if "iou" in which:             # element-wise IoU family, one size per run of the loop: 1 M and 16.8 M pairs
    from object_detection_amd.metrics import bbox_iou
    for n in (1 << 20, 1 << 24):
        g = torch.Generator(device=dev).manual_seed(5)
        b1 = (torch.rand((4, n), generator=g, device=dev) + 0.1).requires_grad_(True)
        b2 = torch.rand((n, 4), generator=g, device=dev) + 0.1
        for kw in (dict(), dict(CIoU=True)):
            for _ in range(reps):
                v = bbox_iou(b1, b2, x1y1x2y2=False, **kw)
                torch.autograd.grad(v.sum(), b1)
        del b1, b2
if "cnloss" in which or "cndecode" in which:
    from object_detection_amd import centernet as CN
    from object_detection_amd.loss import masked_ciou_loss
    B, W, H = bs, 160, 160
    g = torch.Generator(device=dev).manual_seed(3)
    if "cnloss" in which:
        mask = (torch.rand((B, W, H), generator=g, device=dev) < 0.005).float()
        ls = torch.rand((B, 2, W, H), generator=g, device=dev) * 20 * mask[:, None]
        lo = torch.rand((B, 2, W, H), generator=g, device=dev) * mask[:, None]
        ps = (torch.rand((B, 2, W, H), generator=g, device=dev) * 20).requires_grad_(True)
        po = torch.rand((B, 2, W, H), generator=g, device=dev).requires_grad_(True)
        for _ in range(reps):
            v = masked_ciou_loss(po, ps, lo, ls, mask, add_coords=True)
            torch.autograd.grad(v, [ps, po])
        if hasattr(CN, "box_loss_forward_backward"):
            for _ in range(reps):
                CN.box_loss_forward_backward(po.detach(), ps.detach(), lo, ls, mask)
        del mask, ls, lo, ps, po
    if "cndecode" in which:
        heat = torch.randn((B, 80, W, H), generator=g, device=dev) - 1.5
        size = torch.rand((B, 2, W, H), generator=g, device=dev) * 10
        off = torch.rand((B, 2, W, H), generator=g, device=dev)
        for _ in range(reps):
            CN.decode_bbox(heat, size, off, 0.9)
        del heat, size, off
if "head16" in which or "convval" in which:
    from object_detection_amd import yolo
    from object_detection_amd.general import conv_nms
    from object_detection_amd.yolo_loss import ComputeLoss
    anchors = bench.grid_anchors(dev)
    crit = ComputeLoss(bench._Model(anchors, 80))
    levels, targets = bench.synth_batch(bs, 640, 80, 10 * bs, 1000, dev)
    bench.plant_objects(levels, targets, crit)
    strides = torch.tensor([8., 16., 32.])
    convs = [x.permute(0, 1, 4, 2, 3).reshape(bs, 255, x.shape[2], x.shape[3]).contiguous() for x in levels]
    del levels
    if "convval" in which:     # candidate search on the convolution layout at validation thresholds
        for _ in range(reps):
            conv_nms(convs, 3, anchors, strides, conf_thres=0.001, iou_thres=0.6, multi_label=True)
    if "head16" in which:      # the fp16 head: layout change (+ decode) from fp16 convolution outputs
        half = [c.half() for c in convs]
        for _ in range(reps):
            yolo.head_maps(half, 3)
            yolo.head_maps_decode(half, 3, anchors, strides)
torch.cuda.synchronize()
