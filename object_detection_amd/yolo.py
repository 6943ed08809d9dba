"""Drop-in for the YOLOv5 Detect head (object_detection/yolov5/models/yolo.py:33-81)."""
import os
import weakref

import torch
import torch.nn as nn

from . import _lib as L


_ANCHOR_CACHE = {}
_USE_PLANES = os.environ.get("DH_COMPACT_PLANES", "1") != "0"


def use_compact_planes(enabled):
    """Switch for the compact objectness planes (also the environment variable DH_COMPACT_PLANES=0 at import).  Off:
    NMS and the loss always read column 4 of the tensor they are given, whatever the decode left behind.  For callers that
    write to decoded tensors or maps through channels the version counter does not see (`tensor.data`, raw pointers handed
    to other libraries).  Returns the previous setting."""
    global _USE_PLANES
    before, _USE_PLANES = _USE_PLANES, bool(enabled)
    return before

_OBJ_PLANES = {}    # data_ptr of a decoded tensor -> (weakref to it, its version counter, compact objectness (bs, A))
_LOGIT_PLANES = {}  # data_ptr of the first raw map -> ([(weakref, version, shape) per map], raw objectness logits (bs, A))


def objectness_plane(pred):
    """The compact copy of pred[:, :, 4] written by detect_decode for exactly this tensor, or None.

    It is only returned while `pred` is the very tensor detect_decode produced and has not been written to
    since (same storage, shape and version counter), so consumers may read it instead of the strided column.
    """
    hit = _OBJ_PLANES.get(pred.data_ptr()) if _USE_PLANES else None
    if hit is None:
        return None
    ref, version, plane, made_on, in_capture = hit
    if _stale_for_capture(in_capture):
        return None
    if ref() is pred and pred._version == version and plane.shape == pred.shape[:2]:
        return _borrow(plane, made_on)
    return None


def _stale_for_capture(made_in_capture):
    """A launch sequence that is being captured into a HIP graph must not read a plane an eager call wrote: the graph
    would replay against memory whose contents and lifetime belong to that earlier call.  Planes written inside the same
    capture live in the graph's pool and are rewritten by every replay, so they stay usable there."""
    return (not made_in_capture) and torch.cuda.is_current_stream_capturing()


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
    if not maps or not _USE_PLANES:
        return None
    hit = _LOGIT_PLANES.get(maps[0].data_ptr())
    if hit is None or len(hit[0]) != len(maps) or _stale_for_capture(hit[3]):
        return None
    for x, (ref, version, shape) in zip(maps, hit[0]):
        seen = ref()
        if (seen is None or seen.data_ptr() != x.data_ptr() or x._version != version or seen._version != version
                or tuple(x.shape) != shape or x.dtype != torch.float32 or not x.is_contiguous()):
            return None
    return _borrow(hit[1], hit[2])


_STRIDE_VALUES = {}


def _stride_values(stride):
    """The strides as a tuple of host floats.  A model that has been moved to the device holds Detect.stride there
    (Model._apply, yolo.py:231-236): reading it is a blocking device-to-host copy, which a steady loop should not pay per
    call and which is illegal under stream capture.  A device tensor is therefore read once and remembered by identity
    (the very object, at its address and version); host tensors and sequences are read as they are."""
    if not isinstance(stride, torch.Tensor):
        return tuple(float(v) for v in stride)
    if not stride.is_cuda:
        return tuple(float(v) for v in stride.tolist())
    key = (stride.data_ptr(), stride._version, tuple(stride.shape))
    hit = _STRIDE_VALUES.get(key)
    if hit is not None and hit[0]() is stride:
        return hit[1]
    vals = tuple(float(v) for v in stride.tolist())
    if len(_STRIDE_VALUES) > 64:
        _STRIDE_VALUES.clear()
    _STRIDE_VALUES[key] = (weakref.ref(stride), vals)
    return vals


def _anchor_pixels(anchors, stride, half_model=False):
    """Host copies of anchor_grid (= anchors[i] * stride[i], yolo.py:79, formed in fp32 like the reference)
    and of the strides; cached per anchors tensor (the very object, at its version) and stride values, so the
    steady state does no device-to-host copy (a device-resident stride is read once, see _stride_values).  half_model: the
    anchor_grid of a half-precision model (yolo.py:79-80) -- the fp16 anchors buffer times the stride, rounded to fp16,
    then .float()."""
    svals = _stride_values(stride)
    key = (anchors.data_ptr(), anchors._version, tuple(anchors.shape), svals, bool(half_model))
    hit = _ANCHOR_CACHE.get(key)
    if hit is not None and hit[0]() is anchors:      # same address and version but another tensor: reused storage
        return hit[1]
    strd = torch.tensor(svals, dtype=torch.float32)
    if half_model:
        anc = anchors.detach().half().cpu()
        grid = torch.stack([(anc[l] * torch.tensor(svals[l])).float() for l in range(len(svals))])
    else:
        grid = anchors.detach().float().cpu() * strd.view(-1, 1, 1)
    made = (L.float_array(grid.reshape(-1).tolist()), L.float_array(strd.tolist()))
    if len(_ANCHOR_CACHE) > 64:
        _ANCHOR_CACHE.clear()
    _ANCHOR_CACHE[key] = (weakref.ref(anchors), made)
    return made


def detect_decode(levels, anchors, stride, half_arithmetic=False, inplace=True):
    """Inference decode of raw head maps -> (bs, sum na*ny*nx, no)  (yolo.py:56-70).

    levels: list of (bs,na,ny,nx,no) tensors; anchors: (nl,na,2) in grid units; stride: (nl,).
    One kernel launch covers every level and writes the concatenated tensor directly.
    fp16 maps are widened on load and decoded in fp32 to an fp32 tensor, unless half_arithmetic is set: then the result is
    the fp16 tensor a half-precision model computes (model.half(), val.py:129-130), rounding for rounding -- or, with
    inplace=False, the fp32 tensor its torch.cat form yields (yolo.py:63-65).
    """
    xs, half = L.head_maps(levels)
    if half_arithmetic:
        if not half:
            raise RuntimeError("detect_decode(half_arithmetic=True) needs fp16 maps")
        return _detect_decode_half(xs, anchors, stride, inplace)
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
    _remember_planes(out, xs, plane, logits, made_on)
    return out


def _anchor_pixels_half(anchors, stride):
    return _anchor_pixels(anchors, stride, half_model=True)


def _detect_decode_half(xs, anchors, stride, inplace=True):
    """yolo.py:56-70 on a half-precision model: fp16 maps in; fp16 decoded tensor out, or fp32 for the torch.cat form
    (dh_yolo_decode_half)."""
    nl = len(xs)
    bs, na, _, _, no = xs[0].shape
    for x in xs:
        if x.dim() != 5 or x.shape[0] != bs or x.shape[1] != na or x.shape[4] != no:
            raise RuntimeError(f"detect_decode: inconsistent level shape {tuple(x.shape)}")
    ny = [int(x.shape[2]) for x in xs]
    nx = [int(x.shape[3]) for x in xs]
    total = sum(na * a * b for a, b in zip(ny, nx))
    anchor_grid, svals = _anchor_pixels_half(anchors, stride)
    out = torch.empty((bs, total, no), dtype=torch.float16 if inplace else torch.float32, device=xs[0].device)
    # the objectness column once more, compact and widened (exact): non_max_suppression on `out` reads it instead of the
    # strided column, as it does behind the fp32 decode
    plane = torch.empty((bs, total), dtype=torch.float32, device=xs[0].device) if _USE_PLANES else None
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in xs])
    with L.device_guard(out.device):
        made_on = L.raw_stream()
        L.check(L.lib().dh_yolo_decode_half(ptrs, L.int_array(ny), L.int_array(nx), nl, bs, na, no,
                                            anchor_grid, svals, int(bool(inplace)),
                                            L.ptr(out), L.ptr(plane), L.c_void_p(made_on)), "dh_yolo_decode_half")
    if plane is not None:
        _remember_planes(out, None, plane, None, made_on)
    return out


def _remember_planes(out, xs, plane, logits, made_on):
    """Registers the two compact objectness planes a decode launch wrote: `plane` for the decoded tensor `out`,
    `logits` for the raw maps `xs` it was computed from."""
    # entries whose tensor has been freed go at once (they pin 8 bytes per anchor); the bound only guards against a
    # caller that keeps every decoded tensor alive
    for registry, dead in ((_OBJ_PLANES, lambda v: v[0]() is None), (_LOGIT_PLANES, lambda v: any(r() is None for r, _, _ in v[0]))):
        for k in [k for k, v in registry.items() if dead(v)]:
            del registry[k]
        if len(registry) > 8:
            registry.clear()
    in_capture = torch.cuda.is_current_stream_capturing()
    _OBJ_PLANES[out.data_ptr()] = (weakref.ref(out), out._version, plane, made_on, in_capture)
    if logits is not None:
        _LOGIT_PLANES[xs[0].data_ptr()] = ([(weakref.ref(x), x._version, tuple(x.shape)) for x in xs], logits, made_on, in_capture)


def _head_geometry(convs, na):
    bs = int(convs[0].shape[0])
    if any(c.dim() != 4 or c.shape[0] != bs or c.shape[1] % na for c in convs):
        raise RuntimeError(f"Detect head: convolution outputs must be (bs, na*no, ny, nx); got {[tuple(c.shape) for c in convs]}")
    no = int(convs[0].shape[1]) // na
    if any(c.shape[1] != na * no for c in convs):
        raise RuntimeError("Detect head: every level must have na*no channels")
    return bs, no, [int(c.shape[2]) for c in convs], [int(c.shape[3]) for c in convs]


def _head_kernel_takes(convs, no):
    """The layout-change kernel reads fp32 or fp16 convolution outputs (one type for all levels) on a ROCm device and
    holds a tile of all `no` channels in LDS; anything else keeps the reference's own permute (a torch copy)."""
    return (5 <= no <= 256 and convs[0].dtype in (torch.float32, torch.float16)
            and all(c.is_cuda and c.dtype == convs[0].dtype for c in convs))


def _torch_head_maps(convs, na, no):
    return [c.view(c.shape[0], na, no, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2).contiguous() for c in convs]


class _HeadMaps(torch.autograd.Function):
    """yolo.py:54 for every level in one launch, with the inverse layout change as its backward."""

    @staticmethod
    def forward(ctx, na, *convs):
        convs = [c.contiguous() for c in convs]
        bs, no, ny, nx = _head_geometry(convs, na)
        maps = [torch.empty((bs, na, h, w, no), dtype=c.dtype, device=c.device) for c, h, w in zip(convs, ny, nx)]
        nl = len(convs)
        entry = L.lib().dh_yolo_head_maps_f16 if convs[0].dtype == torch.float16 else L.lib().dh_yolo_head_maps
        with L.device_guard(convs[0].device):
            L.check(entry((L.c_void_p * nl)(*[c.data_ptr() for c in convs]),
                                              (L.c_void_p * nl)(*[m.data_ptr() for m in maps]), L.int_array(ny), L.int_array(nx),
                                              nl, bs, na, no, None, None, None, None, None, L.stream()), "dh_yolo_head_maps")
        ctx.geometry = (na, no, ny, nx, bs, convs[0].dtype)
        return tuple(maps)

    @staticmethod
    def backward(ctx, *grads):
        na, no, ny, nx, bs, dtype = ctx.geometry
        dev = next(g for g in grads if g is not None).device
        gm = [(g.to(dtype).contiguous() if g is not None else torch.zeros((bs, na, h, w, no), dtype=dtype, device=dev))
              for g, h, w in zip(grads, ny, nx)]
        gc = [torch.empty((bs, na * no, h, w), dtype=dtype, device=dev) for h, w in zip(ny, nx)]
        nl = len(gm)
        entry = L.lib().dh_yolo_head_maps_bwd_f16 if dtype == torch.float16 else L.lib().dh_yolo_head_maps_bwd
        with L.device_guard(dev):
            L.check(entry((L.c_void_p * nl)(*[g.data_ptr() for g in gm]),
                                                  (L.c_void_p * nl)(*[g.data_ptr() for g in gc]), L.int_array(ny),
                                                  L.int_array(nx), nl, bs, na, no, L.stream()), "dh_yolo_head_maps_bwd")
        return (None,) + tuple(gc)


def head_maps(convs, na):
    """The head maps of every level from the outputs of the 1x1 convolutions (yolo.py:53-54):
    (bs, na*no, ny, nx) -> (bs, na, ny, nx, no) contiguous; differentiable."""
    convs = list(convs)
    _, no, _, _ = _head_geometry(convs, na)
    if not _head_kernel_takes(convs, no):
        return _torch_head_maps(convs, na, no)
    return list(_HeadMaps.apply(na, *convs))


def head_maps_decode(convs, na, anchors, stride, lazy=False, half_arithmetic=False):
    """Eval branch of Detect.forward after the convolutions (yolo.py:53-70): returns (decoded (bs, A, no), head maps).
    lazy: the maps are returned as permuted views of the convolution outputs instead of being written.
    One launch reads the convolution outputs once and writes both; the results equal head_maps + detect_decode bit for bit.
    A half-precision head keeps fp16 maps; the decoded tensor is fp32, as detect_decode returns it for fp16 maps -- or, with
    half_arithmetic, the fp16 tensor of detect_decode(..., half_arithmetic=True) (a half-precision model's in-place decode)."""
    convs = [c.detach() for c in convs]
    bs, no, ny, nx = _head_geometry(convs, na)
    if half_arithmetic and not all(c.is_cuda and c.dtype == torch.float16 for c in convs):
        raise RuntimeError("head_maps_decode(half_arithmetic=True) needs fp16 convolution outputs")
    if not _head_kernel_takes(convs, no):
        maps = _torch_head_maps(convs, na, no)
        return detect_decode(maps, anchors, stride, half_arithmetic=half_arithmetic), maps
    convs = [c.contiguous() for c in convs]
    nl = len(convs)
    dev = convs[0].device
    total = sum(na * h * w for h, w in zip(ny, nx))
    anchor_px, strides = _anchor_pixels(anchors, stride, half_model=bool(half_arithmetic))
    if lazy:
        maps = [c.view(bs, na, no, h, w).permute(0, 1, 3, 4, 2) for c, h, w in zip(convs, ny, nx)]
    else:
        maps = [torch.empty((bs, na, h, w, no), dtype=convs[0].dtype, device=dev) for h, w in zip(ny, nx)]
    if half_arithmetic:
        out = torch.empty((bs, total, no), dtype=torch.float16, device=dev)
        plane = torch.empty((bs, total), dtype=torch.float32, device=dev) if _USE_PLANES else None
        with L.device_guard(dev):
            made_on = L.raw_stream()
            L.check(L.lib().dh_yolo_head_maps_half((L.c_void_p * nl)(*[c.data_ptr() for c in convs]),
                                                   None if lazy else (L.c_void_p * nl)(*[m.data_ptr() for m in maps]),
                                                   L.int_array(ny), L.int_array(nx), nl, bs, na, no, anchor_px, strides, L.ptr(out),
                                                   L.ptr(plane), L.c_void_p(made_on)), "dh_yolo_head_maps_half")
        if plane is not None:
            _remember_planes(out, None, plane, None, made_on)
        return out, maps
    out = torch.empty((bs, total, no), dtype=torch.float32, device=dev)
    planes = torch.empty((2, bs, total), dtype=torch.float32, device=dev)
    entry = L.lib().dh_yolo_head_maps_f16 if convs[0].dtype == torch.float16 else L.lib().dh_yolo_head_maps
    with L.device_guard(dev):
        made_on = L.raw_stream()
        L.check(entry((L.c_void_p * nl)(*[c.data_ptr() for c in convs]),
                                          None if lazy else (L.c_void_p * nl)(*[m.data_ptr() for m in maps]), L.int_array(ny), L.int_array(nx),
                                          nl, bs, na, no, anchor_px, strides, L.ptr(out), L.ptr(planes[0]), L.ptr(planes[1]),
                                          L.c_void_p(made_on)), "dh_yolo_head_maps")
    _remember_planes(out, maps, planes[0], planes[1], made_on)
    return out, maps


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
    The 1x1 convolutions stay ordinary nn.Conv2d layers; what follows them (the layout change of yolo.py:54 and the
    decode arithmetic) is one launch in eval mode, and a layout-change launch with its own backward in training.
    """
    stride = None
    onnx_dynamic = False
    lazy_maps = False     # return the maps as permuted views of the convolution outputs (no copy of yolo.py:54);
    # ComputeLoss recognises them and works on the convolution outputs directly, gradients included
    half_arithmetic = True    # a half-precision head (model.half(), val.py:129-130) returns the fp16 decoded tensor of
    # yolo.py:56-70 with torch's fp16 roundings; False: fp16 maps are widened on load and decoded in fp32 to an fp32 tensor
    # (one launch instead of two, closer to the fp32 model's result, not what the reference computes)

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

    def detections(self, x, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, labels=(),
                   max_det=300):
        """non_max_suppression(self(x)[0], ...) for callers that want nothing else from the head (detect.py:126-132,
        val.py:194-203 without compute_loss): the convolutions run as in forward, the detections come straight from
        their outputs; neither the head maps nor the decoded tensor are made.  Same rows as the two calls, bit for bit: for
        an fp16 head with half_arithmetic (the default) the fp16 roundings of the reference's half-precision model are kept
        (in-place decode; with inplace=False the two calls run), without it the arithmetic is fp32 on the widened values."""
        from .general import conv_nms, non_max_suppression
        with torch.no_grad():
            half = all(v.is_cuda and v.dtype == torch.float16 for v in x) and self.m[0].weight.dtype == torch.float16
            if half and self.half_arithmetic and not self.inplace:
                return non_max_suppression(self([v for v in x])[0], conf_thres, iou_thres, classes, agnostic, multi_label, labels,
                                           max_det)
            convs = [self.m[i](x[i]) for i in range(self.nl)]
            return conv_nms(convs, self.na, self.anchors, self.stride, conf_thres, iou_thres, classes, agnostic, multi_label,
                            labels, max_det, half_arithmetic=half and self.half_arithmetic)

    def _make_grid(self, nx=20, ny=20, i=0):
        """yolo.py:72-81: cell indices (x, y) and anchors in pixels, both (1, na, ny, nx, 2).  The decode kernel derives
        them from the row index; the attributes are kept for callers that read them (export, plotting)."""
        d = self.anchors[i].device
        yv, xv = torch.meshgrid([torch.arange(ny, device=d), torch.arange(nx, device=d)], indexing='ij')
        grid = torch.stack((xv, yv), 2).expand((1, self.na, ny, nx, 2)).float()
        anchor_grid = (self.anchors[i].clone() * self.stride[i]).view((1, self.na, 1, 1, 2)).expand((1, self.na, ny, nx, 2)).float()
        return grid, anchor_grid

    def forward(self, x):
        for i in range(self.nl):
            x[i] = self.m[i](x[i])
            if not self.training and (self.onnx_dynamic or self.grid[i].shape[2:4] != x[i].shape[2:4]):   # yolo.py:57-58
                self.grid[i], self.anchor_grid[i] = self._make_grid(int(x[i].shape[3]), int(x[i].shape[2]), i)
        if self.training and self.lazy_maps:
            for i in range(self.nl):
                bs, _, ny, nx = x[i].shape
                x[i] = x[i].contiguous().view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2)
            return x
        wants_grad = torch.is_grad_enabled() and any(c.requires_grad for c in x)
        if not self.training and not wants_grad and self.half_arithmetic and all(c.is_cuda and c.dtype == torch.float16 for c in x):
            if self.inplace:          # layout change and the half model's decode in one launch
                z, x[:] = head_maps_decode(x, self.na, self.anchors, self.stride, lazy=self.lazy_maps, half_arithmetic=True)
                return z, x
            x[:] = head_maps(x, self.na)
            return detect_decode(x, self.anchors, self.stride, half_arithmetic=True, inplace=False), x
        if self.training or wants_grad:
            x[:] = head_maps(x, self.na)
            return x if self.training else (detect_decode(x, self.anchors, self.stride), x)
        z, x[:] = head_maps_decode(x, self.na, self.anchors, self.stride, lazy=self.lazy_maps)
        return z, x
