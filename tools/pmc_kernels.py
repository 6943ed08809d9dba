#!/usr/bin/env python3
"""A few launches of the kernels whose timings and counters the design document quotes, at the sizes it quotes them for, and
nothing else: the program to put behind `rocprofv3 --kernel-trace ... --` and `rocprofv3 --pmc ... --`.

  python tools/pmc_kernels.py [pairwise,iou,cnloss,cndecode,convval,head] [launches]

Every group of launches is preceded by a marker: a dh_stream_copy whose grid size is the group's number in PHASES below.
tools/kernel_trace_by_grid.py and tools/pmc_by_grid.py read the markers and report each group on its own, so that a kernel
that runs at several sizes or in several modes is never averaged over them.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = {
    1: "box_iou 100 x 300 (labels x detections of an image, val.py:81)",
    2: "box_iou 300 x 3000 (merge-NMS, general.py:727)",
    3: "box_iou 8192 x 8192 (write-bound, rows on 128-byte lines): 268.7 MB",
    4: "box_iou 30000 x 3000 (write-bound, rows off 128-byte lines): 360.5 MB",
    10: "bbox_iou IoU fwd+bwd, 1 Mi pairs: 37.7 MB fwd, 54.5 MB bwd",
    11: "bbox_iou CIoU fwd+bwd, 1 Mi pairs",
    12: "bbox_iou IoU fwd+bwd, 16 Mi pairs: 604 MB fwd, 872 MB bwd",
    13: "bbox_iou CIoU fwd+bwd, 16 Mi pairs",
    20: "CenterNet box loss through autograd, bs 64 x 160 x 160: fwd 59 MB, reduction, bwd 85 MB",
    21: "CenterNet box loss, fused forward+backward entry (two launches)",
    30: "CenterNet decode_bbox, bs 64 x 80 x 160 x 160 (524 MB of class maps)",
    40: "conv_nms at validation thresholds (conf 0.001, multi-label), bs 64 @ 640: 548 MB of convolution outputs",
    50: "head maps from fp16 convolution outputs, layout change only: 274 MB read + 274 MB written",
    51: "head maps + decode from fp16 convolution outputs: 274 MB read + 274 MB + 548 MB written",
    52: "head maps from fp32 convolution outputs, layout change only: 548 MB read + 548 MB written",
    53: "head maps + decode from fp32 convolution outputs: 548 MB read + 548 MB + 548 MB written",
    54: "head maps backward (gradient back to the convolution layout), fp32: 548 MB read + 548 MB written",
    99: "set-up of the next group (not a measurement)",
}


def main():
    import bench
    from object_detection_amd import _lib as L
    which = set((sys.argv[1] if len(sys.argv) > 1 else "pairwise,iou,cnloss,cndecode,convval,head").split(","))
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    bs = 64
    mark_src = torch.zeros(1 << 20, device=dev)
    mark_dst = torch.empty_like(mark_src)

    def phase(k):
        assert k in PHASES and 4096 * k <= mark_src.numel() * 4
        L.check(L.lib().dh_stream_copy(L.ptr(mark_src), L.ptr(mark_dst), 4096 * k, L.stream()), "dh_stream_copy")

    if "pairwise" in which:
        from object_detection_amd.metrics import box_iou
        g = torch.Generator(device=dev).manual_seed(6)
        for k, (n, m) in zip((1, 2, 3, 4), ((100, 300), (300, 3000), (8192, 8192), (30000, 3000))):
            xy1, xy2 = torch.rand((n, 2), generator=g, device=dev) * 600, torch.rand((m, 2), generator=g, device=dev) * 600
            b1 = torch.cat((xy1, xy1 + torch.rand((n, 2), generator=g, device=dev) * 200 + 1), 1)
            b2 = torch.cat((xy2, xy2 + torch.rand((m, 2), generator=g, device=dev) * 200 + 1), 1)
            out = torch.empty((n, m), device=dev)
            phase(k)
            for _ in range(reps):
                box_iou(b1, b2, out=out)
            del out
    if "iou" in which:
        from object_detection_amd.metrics import bbox_iou
        k = 10
        for n in (1 << 20, 1 << 24):
            g = torch.Generator(device=dev).manual_seed(5)
            b1 = (torch.rand((4, n), generator=g, device=dev) + 0.1).requires_grad_(True)
            b2 = torch.rand((n, 4), generator=g, device=dev) + 0.1
            for kw in (dict(), dict(CIoU=True)):
                phase(k)
                k += 1
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
            phase(20)
            for _ in range(reps):
                v = masked_ciou_loss(po, ps, lo, ls, mask, add_coords=True)
                torch.autograd.grad(v, [ps, po])
            phase(21)
            for _ in range(reps):
                CN.box_loss_forward_backward(po.detach(), ps.detach(), lo, ls, mask)
            del mask, ls, lo, ps, po
        if "cndecode" in which:
            heat = torch.randn((B, 80, W, H), generator=g, device=dev) - 1.5
            size = torch.rand((B, 2, W, H), generator=g, device=dev) * 10
            off = torch.rand((B, 2, W, H), generator=g, device=dev)
            phase(30)
            for _ in range(reps):
                CN.decode_bbox(heat, size, off, 0.9)
            del heat, size, off
    if "head" in which or "convval" in which:
        phase(99)
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
        if "convval" in which:
            phase(40)
            for _ in range(reps):
                conv_nms(convs, 3, anchors, strides, conf_thres=0.001, iou_thres=0.6, multi_label=True)
        if "head" in which:
            half = [c.half() for c in convs]
            phase(50)
            for _ in range(reps):
                yolo.head_maps(half, 3)
            phase(51)
            for _ in range(reps):
                yolo.head_maps_decode(half, 3, anchors, strides)
            del half
            phase(52)
            for _ in range(reps):
                yolo.head_maps(convs, 3)
            phase(53)
            for _ in range(reps):
                yolo.head_maps_decode(convs, 3, anchors, strides)
            phase(99)
            leaves = [c.clone().requires_grad_(True) for c in convs]
            maps = yolo.head_maps(leaves, 3)
            gr = [torch.ones_like(m) for m in maps]
            phase(54)
            for _ in range(reps):
                torch.autograd.grad(maps, leaves, gr, retain_graph=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
