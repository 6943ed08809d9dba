#!/usr/bin/env python3
"""Times the Detect head after its convolutions at bs 64 @ 640 px (YOLOv5s, nc 80) on one GPU:
torch's permute + contiguous (yolo.py:54), that + detect_decode, the fused launch, and the two backward forms.

  python tools/time_head_maps.py [bs]        (DH_HEAD_TILE=128 selects the 128-position tile)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import yolo  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
na, no = 3, 85
dev = torch.device("cuda:0")
convs = [torch.randn(bs, na * no, n, n, device=dev) for n in (80, 40, 20)]
anchors = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]],
                       dtype=torch.float32, device=dev).view(3, 3, 2) / torch.tensor([8.0, 16.0, 32.0], device=dev).view(3, 1, 1)
stride = torch.tensor([8.0, 16.0, 32.0])
elems = sum(c.numel() for c in convs)


def torch_permute():
    return [c.view(bs, na, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for c in convs]


def timed(fn, reps=30):
    import gc
    for _ in range(5):
        fn()
    gc.collect()                # a full collection inside the timed calls would add tens of milliseconds to them
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


maps = torch_permute()
grads = [torch.randn_like(m) for m in maps]


def torch_bwd():      # what autograd + the convolution's backward do with a map-order gradient
    return [g.permute(0, 1, 4, 2, 3).contiguous() for g in grads]


def ours_bwd():
    return yolo._HeadMaps.backward(ctx, *grads)


class ctx:
    geometry = (na, no, [80, 40, 20], [80, 40, 20], bs, torch.float32)


rows = [
    ("torch permute+contiguous (yolo.py:54)", torch_permute, 2 * elems * 4),
    ("torch permute + detect_decode", lambda: yolo.detect_decode(torch_permute(), anchors, stride), 4 * elems * 4),
    ("detect_decode alone", lambda: yolo.detect_decode(maps, anchors, stride), 2 * elems * 4),
    ("head_maps (layout change only)", lambda: yolo.head_maps(convs, na), 2 * elems * 4),
    ("head_maps_decode (fused)", lambda: yolo.head_maps_decode(convs, na, anchors, stride), 3 * elems * 4),
    ("head_maps_decode, lazy maps (decoded tensor only)", lambda: yolo.head_maps_decode(convs, na, anchors, stride, lazy=True), 2 * elems * 4),
    ("torch backward copy", torch_bwd, 2 * elems * 4),
    ("head_maps backward", ours_bwd, 2 * elems * 4),
]
h_convs = [c.half() for c in convs]
h_maps = [m.half() for m in maps]


def torch_permute_h():
    return [c.view(bs, na, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for c in h_convs]


rows += [
    ("fp16: torch permute + detect_decode", lambda: yolo.detect_decode(torch_permute_h(), anchors, stride), elems * (2 + 2 + 2 + 4)),
    ("fp16: detect_decode alone", lambda: yolo.detect_decode(h_maps, anchors, stride), elems * (2 + 4)),
    ("fp16: head_maps (layout change only)", lambda: yolo.head_maps(h_convs, na), elems * 4),
    ("fp16: head_maps_decode (fused)", lambda: yolo.head_maps_decode(h_convs, na, anchors, stride), elems * (2 + 2 + 4)),
    ("fp16, half model's arithmetic: head_maps + detect_decode(half_arithmetic)",
     lambda: yolo.detect_decode(yolo.head_maps(h_convs, na), anchors, stride, half_arithmetic=True), elems * (2 + 2 + 2 + 2)),
    ("fp16, half model's arithmetic: head_maps_decode (fused)",
     lambda: yolo.head_maps_decode(h_convs, na, anchors, stride, half_arithmetic=True), elems * (2 + 2 + 2)),
    ("fp16, half model's arithmetic: head_maps_decode (fused, lazy maps)",
     lambda: yolo.head_maps_decode(h_convs, na, anchors, stride, lazy=True, half_arithmetic=True), elems * (2 + 2)),
]
# detections straight from the convolution outputs against Detect.forward + non_max_suppression
from object_detection_amd import general as G  # noqa: E402

for c in convs:        # the bias statistics of yolo.py:196-204, so that the thresholds see realistic candidate counts
    v = c.view(bs, na, no, c.shape[2], c.shape[3])
    v[:, :, 4] += float(torch.log(torch.tensor(8 / (640 / (640 // c.shape[2])) ** 2)))
    v[:, :, 5:] += float(torch.log(torch.tensor(0.6 / (80 - 0.999999))))
    hot = torch.rand(bs, na, c.shape[2], c.shape[3], device=dev) < 0.002
    v[:, :, 4] += hot * 7.0
    v[:, :, 5] += hot * 8.0
VAL = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)
rows += [
    ("Detect eval + NMS(0.25): fused head + nms", lambda: G.non_max_suppression(yolo.head_maps_decode(convs, na, anchors, stride)[0]), elems * 4),
    ("conv_nms(0.25)", lambda: G.conv_nms(convs, na, anchors, stride), elems * 4),
    ("Detect eval + NMS(0.001, multi-label)", lambda: G.non_max_suppression(yolo.head_maps_decode(convs, na, anchors, stride)[0], **VAL), elems * 4),
    ("conv_nms(0.001, multi-label)", lambda: G.conv_nms(convs, na, anchors, stride, **VAL), elems * 4),
]
# the training step's head: maps from the convolution outputs, loss, backward down to the convolution outputs
from object_detection_amd.yolo_loss import ComputeLoss  # noqa: E402


class _Det:
    na, nl, nc, anchors, stride = na, 3, 80, anchors, stride


class _Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.model = [_Det()]
        self.hyp = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


crit = ComputeLoss(_Model())
nt = 10 * bs
targets = torch.cat((torch.randint(0, bs, (nt, 1)).float(), torch.randint(0, 80, (nt, 1)).float(), torch.rand(nt, 2),
                     torch.exp(torch.empty(nt, 2).uniform_(-4.6, -0.5))), 1).to(dev)
leaves = [c.clone().requires_grad_() for c in convs]


def train_head(make_maps):
    # autograd.grad hands the gradients over as they arrive (backward() would add AccumulateGrad's clone of each, which a
    # convolution's backward -- the real consumer -- does not make)
    loss, _ = crit(make_maps(leaves), targets)
    return [g.contiguous() for g in torch.autograd.grad(loss, leaves)]      # contiguous: what the convolution's backward asks for


def lazy(cs):
    return [c.view(bs, na, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2) for c in cs]


rows += [
    ("train head: torch permute copy + loss + backward", lambda: train_head(lambda cs: [v.contiguous() for v in lazy(cs)]), 5 * elems * 4),
    ("train head: head_maps + loss + backward", lambda: train_head(lambda cs: yolo.head_maps(cs, na)), 5 * elems * 4),
    ("train head: lazy maps + loss + backward", lambda: train_head(lazy), elems * 4),
]
# the same under autocast (train.py:328-330): the convolution outputs, and so the maps, are fp16
h_leaves = [c.half().requires_grad_() for c in convs]


def train_head_h(make_maps):
    loss, _ = crit(make_maps(h_leaves), targets)
    return [g.contiguous() for g in torch.autograd.grad(loss, h_leaves)]


rows += [
    ("fp16 train head: torch permute copy + loss + backward", lambda: train_head_h(lambda cs: [v.contiguous() for v in lazy(cs)]), elems * 10),
    ("fp16 train head: head_maps + loss + backward", lambda: train_head_h(lambda cs: yolo.head_maps(cs, na)), elems * 10),
    ("fp16 train head: lazy maps + loss + backward", lambda: train_head_h(lazy), elems * 6),
]
for name, fn, nbytes in rows:
    ms = timed(fn)
    print(f"{name:42s} {ms * 1e3:8.1f} us   {nbytes / ms / 1e6:8.1f} GB/s of its own algorithmic bytes", flush=True)
