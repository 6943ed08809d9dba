"""Drop-in for the YOLOv5 Detect head (object_detection/yolov5/models/yolo.py:33-81)."""
import weakref

import torch
import torch.nn as nn

from . import _lib as L


_ANCHOR_CACHE = {}
_OBJ_PLANES = {}    # data_ptr of a decoded tensor -> (weakref to it, its version counter, compact objectness (bs, A))
_LOGIT_PLANES = {}  # data_ptr of the first raw map -> ([(weakref, version, shape) per map], raw objectness logits (bs, A))


def objectness_plane(pred):
    """The compact copy of pred[:, :, 4] written by detect_decode for exactly this tensor, or None.

    It is only returned while `pred` is the very tensor detect_decode produced and has not been written to
    since (same storage, shape and version counter), so consumers may read it instead of the strided column.
    """
    hit = _OBJ_PLANES.get(pred.data_ptr())
    if hit is None:
        return None
    ref, version, plane, made_on = hit
    if ref() is pred and pred._version == version and plane.shape == pred.shape[:2]:
        return _borrow(plane, made_on)
    return None


def _borrow(plane, made_on):
    """A plane is written on the stream detect_decode ran on; a reader on another stream has to tell the
    caching allocator, or the block could be handed out again while that stream still reads it."""
    if L._raw_stream(plane.device.index) != made_on:
        plane.record_stream(torch.cuda.current_stream(plane.device))
    return plane


def objectness_logits(maps):
    """The compact copy of the objectness logits (channel 4 of every map, decoded row order, (bs, A)) that
    detect_decode wrote when it last read exactly these maps, or None.

    Valid only while every map still is the storage detect_decode read (the tensor it saw is alive, same
    address, shape and version counter, i.e. no in-place write since), so reading it is reading the maps.
    """
    if not maps:
        return None
    hit = _LOGIT_PLANES.get(maps[0].data_ptr())
    if hit is None or len(hit[0]) != len(maps):
        return None
    for x, (ref, version, shape) in zip(maps, hit[0]):
        seen = ref()
        if (seen is None or seen.data_ptr() != x.data_ptr() or x._version != version or seen._version != version
                or tuple(x.shape) != shape or x.dtype != torch.float32 or not x.is_contiguous()):
            return None
    return _borrow(hit[1], hit[2])


def _anchor_pixels(anchors, stride):
    """Host copies of anchor_grid (= anchors[i] * stride[i], yolo.py:79, formed in fp32 like the reference)
    and of the strides; cached per anchors tensor version so the steady state does no device-to-host copy."""
    key = (anchors.data_ptr(), anchors._version, tuple(anchors.shape), id(stride))
    hit = _ANCHOR_CACHE.get(key)
    if hit is None:
        anc = anchors.detach().float().cpu()
        strd = torch.as_tensor(stride).detach().float().cpu()
        hit = (L.float_array((anc * strd.view(-1, 1, 1)).reshape(-1).tolist()), L.float_array(strd.tolist()))
        if len(_ANCHOR_CACHE) > 64:
            _ANCHOR_CACHE.clear()
        _ANCHOR_CACHE[key] = hit
    return hit


def detect_decode(levels, anchors, stride):
    """Inference decode of raw head maps -> (bs, sum na*ny*nx, no)  (yolo.py:56-70).

    levels: list of (bs,na,ny,nx,no) tensors; anchors: (nl,na,2) in grid units; stride: (nl,).
    One kernel launch covers every level and writes the concatenated tensor directly.
    """
    xs, half = L.head_maps(levels)
    nl = len(xs)
    bs, na, _, _, no = xs[0].shape
    for x in xs:
        if x.dim() != 5 or x.shape[0] != bs or x.shape[1] != na or x.shape[4] != no:
            raise RuntimeError(f"detect_decode: inconsistent level shape {tuple(x.shape)}")
    ny = [int(x.shape[2]) for x in xs]
    nx = [int(x.shape[3]) for x in xs]
    total = sum(na * a * b for a, b in zip(ny, nx))
    anchor_px, strides = _anchor_pixels(anchors, stride)
    out = torch.empty((bs, total, no), dtype=torch.float32, device=xs[0].device)
    planes = torch.empty((2, bs, total), dtype=torch.float32, device=xs[0].device)
    plane, logits = planes[0], planes[1]
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in xs])
    with L.device_guard(out.device):
        made_on = L.raw_stream()
        entry = L.lib().dh_yolo_decode_f16 if half else L.lib().dh_yolo_decode     # fp16 maps are widened on load
        L.check(entry(ptrs, L.int_array(ny), L.int_array(nx), nl, bs, na, no, anchor_px, strides,
                      L.ptr(out), L.ptr(plane), L.ptr(logits), L.c_void_p(made_on)), "dh_yolo_decode")
    # entries whose tensor has been freed go at once (they pin 8 bytes per anchor); the bound only guards against a
    # caller that keeps every decoded tensor alive
    for registry, dead in ((_OBJ_PLANES, lambda v: v[0]() is None), (_LOGIT_PLANES, lambda v: any(r() is None for r, _, _ in v[0]))):
        for k in [k for k, v in registry.items() if dead(v)]:
            del registry[k]
        if len(registry) > 8:
            registry.clear()
    _OBJ_PLANES[out.data_ptr()] = (weakref.ref(out), out._version, plane, made_on)
    _LOGIT_PLANES[xs[0].data_ptr()] = ([(weakref.ref(x), x._version, tuple(x.shape)) for x in xs], logits, made_on)
    return out


def descale_pred(p, flips, scale, img_size):
    """Model._descale_pred, in-place branch (yolo.py:155-163): p (..., no) decoded predictions of one augmented pass;
    boxes /= scale, flips 2 / 3 mirror y / x back inside img_size = (height, width).  Modifies and returns p."""
    if p.dtype != torch.float32 or not p.is_cuda or p.dim() < 2 or p.shape[-1] < 4 or not p.is_contiguous():
        raise RuntimeError(f"descale_pred modifies contiguous fp32 ROCm tensors (..., >=4) in place; got {p.dtype} "
                           f"{tuple(p.shape)} on {p.device}")
    rows = p.numel() // p.shape[-1]
    with L.device_guard(p.device):
        L.check(L.lib().dh_descale_pred(L.ptr(p), int(p.shape[-1]), rows, float(scale), int(flips) if flips else 0,
                                        float(img_size[0]), float(img_size[1]), L.stream()), "dh_descale_pred")
    torch.autograd.graph.increment_version(p)      # the kernel wrote through the raw pointer: keep the version counter honest
    return p


def clip_augmented(y, nl):
    """Model._clip_augmented (yolo.py:174-183): drop the tails of the first and last augmented outputs (views only)."""
    g = sum(4 ** x for x in range(nl))
    e = 1
    i = (y[0].shape[1] // g) * sum(4 ** x for x in range(e))
    y[0] = y[0][:, :-i]
    i = (y[-1].shape[1] // g) * sum(4 ** (nl - 1 - x) for x in range(e))
    y[-1] = y[-1][:, i:]
    return y


class Detect(nn.Module):
    """Same constructor, buffers and forward contract as yolo.py:33-70.

    Training: returns the list of (bs,na,ny,nx,no) maps.  Eval: returns (decoded (bs,A,no), that list).
    The 1x1 convolutions stay ordinary nn.Conv2d layers; only the decode arithmetic is replaced.
    """
    stride = None
    onnx_dynamic = False

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        self.anchor_grid = [torch.zeros(1)] * self.nl
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.inplace = inplace

    def forward(self, x):
        for i in range(self.nl):
            x[i] = self.m[i](x[i])
            bs, _, ny, nx = x[i].shape
            x[i] = x[i].view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        if self.training:
            return x
        return detect_decode(x, self.anchors, self.stride), x
