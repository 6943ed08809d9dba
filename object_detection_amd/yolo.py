"""Drop-in for the YOLOv5 Detect head (object_detection/yolov5/models/yolo.py:33-81)."""
import weakref

import torch
import torch.nn as nn

from . import _lib as L


_ANCHOR_CACHE = {}
_OBJ_PLANES = {}    # data_ptr of a decoded tensor -> (weakref to it, its version counter, compact objectness (bs, A))


def objectness_plane(pred):
    """The compact copy of pred[:, :, 4] written by detect_decode for exactly this tensor, or None.

    It is only returned while `pred` is the very tensor detect_decode produced and has not been written to
    since (same storage, shape and version counter), so consumers may read it instead of the strided column.
    """
    hit = _OBJ_PLANES.get(pred.data_ptr())
    if hit is None:
        return None
    ref, version, plane = hit
    if ref() is pred and pred._version == version and plane.shape == pred.shape[:2]:
        return plane
    return None


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
    xs = [L.dev_f32(x, f"x[{i}]").contiguous() for i, x in enumerate(levels)]
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
    plane = torch.empty((bs, total), dtype=torch.float32, device=xs[0].device)
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in xs])
    with torch.cuda.device(out.device):
        L.check(L.lib().dh_yolo_decode(ptrs, L.int_array(ny), L.int_array(nx), nl, bs, na, no, anchor_px, strides,
                                       L.ptr(out), L.ptr(plane), L.stream()), "dh_yolo_decode")
    if len(_OBJ_PLANES) > 8:
        for k in [k for k, v in _OBJ_PLANES.items() if v[0]() is None]:
            del _OBJ_PLANES[k]
        if len(_OBJ_PLANES) > 8:
            _OBJ_PLANES.clear()
    _OBJ_PLANES[out.data_ptr()] = (weakref.ref(out), out._version, plane)
    return out


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
