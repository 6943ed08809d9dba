"""Drop-in for ComputeLoss (object_detection/yolov5/utils/loss.py:91-222)."""
import ctypes

import torch
import torch.nn as nn

from . import _lib as L
from .yolo import objectness_logits


def smooth_BCE(eps=0.1):
    """Positive / negative label-smoothing BCE targets (loss.py:13-15)."""
    return 1.0 - 0.5 * eps, 0.5 * eps


def is_parallel(model):
    """object_detection/yolov5/utils/torch_utils.py:145-147."""
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


class _Assignment:
    """Padded device buffers produced by dh_build_targets plus the per-level row counts (device)."""

    def __init__(self, shapes, nt, na, nl, device):
        self.ny = [int(s[0]) for s in shapes]
        self.nx = [int(s[1]) for s in shapes]
        self.nl, self.na, self.nt = nl, na, int(nt)
        self.cap = 5 * na * self.nt
        rows = max(self.cap, 1)
        self.idx = torch.empty((nl, 4, rows), dtype=torch.int64, device=device)
        self.tbox = torch.empty((nl, rows, 4), dtype=torch.float32, device=device)
        self.anch = torch.empty((nl, rows, 2), dtype=torch.float32, device=device)
        self.tcls = torch.empty((nl, rows), dtype=torch.int64, device=device)
        self.counts = torch.empty(nl, dtype=torch.int32, device=device)
        self.made_on = L._raw_stream(device.index)      # the stream the assignment kernel is enqueued on
        self.sums = None                                # set by the loss forward: see _sums_view

    def borrow(self):
        """Called by a consumer: when it runs on another stream than the producer, the caching allocator has to know."""
        dev = self.counts.device
        if L._raw_stream(dev.index) != self.made_on:
            cur = torch.cuda.current_stream(dev)
            for t in (self.idx, self.tbox, self.anch, self.tcls, self.counts):
                t.record_stream(cur)


def _assign(level_shapes, targets, anchors, anchor_t):
    """Run the assignment kernel; nothing is copied back to the host."""
    tg = L.dev_f32(targets, "targets").contiguous()
    if tg.dim() != 2 or tg.shape[1] != 6:
        raise RuntimeError(f"targets must be (nt,6) [image,class,x,y,w,h], got {tuple(targets.shape)}")
    anc = L.dev_f32(anchors, "anchors").contiguous()
    nl, na = int(anc.shape[0]), int(anc.shape[1])
    A = _Assignment(level_shapes, tg.shape[0], na, nl, tg.device)
    lib = L.lib()
    ws_bytes = lib.dh_build_targets_workspace(nl, na, A.nt)
    ws = L.workspace(ws_bytes, tg.device)
    with L.device_guard(tg.device):
        L.check(lib.dh_build_targets(L.ptr(tg), A.nt, L.ptr(anc), nl, na, L.int_array(A.ny), L.int_array(A.nx),
                                     float(anchor_t), A.cap, L.ptr(A.idx), L.ptr(A.tbox), L.ptr(A.anch), L.ptr(A.tcls),
                                     L.ptr(A.counts), L.ptr(ws), ws_bytes, L.stream()), "dh_build_targets")
    return A


def conv_layout_of(p):
    """The convolution outputs behind the head maps `p`, or None.

    Detect.forward with lazy_maps returns every map as the permuted VIEW x.view(bs,na,no,ny,nx).permute(0,1,3,4,2) of its
    convolution's output instead of the contiguous copy of yolo.py:54.  Such a view has the reference's shape and values; only
    its strides tell.  When every p[i] is one (all fp32 or all fp16, on the device), the result is the list of
    (bs, na*no, ny, nx) tensors they are views of, which the loss kernels read -- and write the gradient for -- as they are
    (fp16 elements are widened on load; the gradient is fp32 and autograd converts it to the maps' type)."""
    convs = []
    for x in p:
        if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 5 and x.dtype in (torch.float32, torch.float16)
                and x.dtype == p[0].dtype) or x.is_contiguous():
            return None
        c = x.permute(0, 1, 4, 2, 3)
        if not c.is_contiguous():
            return None
        bs, na, no, ny, nx = c.shape
        convs.append(c.reshape(bs, na * no, ny, nx))
    return convs or None


def _as_map_views(convs, na):
    """(bs, na*no, ny, nx) tensors as (bs, na, ny, nx, no) views (no copy)."""
    return [c.view(c.shape[0], na, c.shape[1] // na, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2) for c in convs]


def _check_levels(maps, assign, nc, na, conv):
    bs = int(maps[0].shape[0])
    for x, ny, nx in zip(maps, assign.ny, assign.nx):
        want = (bs, na * (nc + 5), ny, nx) if conv else (bs, na, ny, nx, nc + 5)
        if tuple(x.shape) != want:
            raise RuntimeError(f"ComputeLoss: level shape {tuple(x.shape)} does not match {want}")
    return bs


class _Sums:
    """Where the loss kernels left their (nl, 5) float64 block of sums -- per level [sum(1 - iou), sum class BCE, sum
    objectness BCE, assigned rows, cells]: the means of loss.py:136,149,155 before the division -- inside a workspace.
    Kept as a recipe: the view is only built when somebody asks for it (a training step that does not report global loss
    terms never does)."""
    __slots__ = ("ws", "geometry")

    def __init__(self, ws, assign, bs, na, nc, nl):
        self.ws, self.geometry = ws, (assign.ny, assign.nx, nl, bs, na, nc, assign.cap)

    _offsets = {}

    def view(self):
        ny, nx, nl, bs, na, nc, cap = self.geometry
        key = (tuple(ny), tuple(nx), nl, bs, na, nc, cap)
        off = _Sums._offsets.get(key)
        if off is None:
            if len(_Sums._offsets) > 64:
                _Sums._offsets.clear()
            off = _Sums._offsets[key] = L.lib().dh_yolo_loss_sums_offset(L.int_array(ny), L.int_array(nx), nl, bs, na, nc, cap)
        return self.ws[off:off + 40 * nl].view(torch.float64).view(nl, 5)


def _sums_view(ws, assign, bs, na, nc, nl):
    return _Sums(ws, assign, bs, na, nc, nl)


def _loss_fwd(assign, hyp, nc, maps, conv_na=0):
    """Launch the forward kernels; returns (out[5+nl], workspace) -- the workspace feeds the backward.
    conv_na = na: `maps` are convolution outputs (bs, na*no, ny, nx)."""
    nl = len(maps)
    na = conv_na or int(maps[0].shape[1])
    bs = _check_levels(maps, assign, nc, na, bool(conv_na))
    dev = maps[0].device
    lib = L.lib()
    ny, nx = L.int_array(assign.ny), L.int_array(assign.nx)
    ws_bytes = lib.dh_yolo_loss_workspace(ny, nx, nl, bs, na, nc, assign.cap)
    if ws_bytes == 0:
        raise RuntimeError("ComputeLoss: unsupported problem size (a level exceeds 2^32 elements)")
    ws = L.workspace(ws_bytes, dev)
    out = torch.empty(6 + nl, dtype=torch.float32, device=dev)
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in maps])
    # when the inference decode has just read these very maps (validation: val.py:194-198 evaluates the loss on the
    # maps it decoded), it left a compact copy of their objectness logits; read that instead of the strided column
    if conv_na:
        entry = lib.dh_yolo_loss_fwd_conv_f16 if maps[0].dtype == torch.float16 else lib.dh_yolo_loss_fwd_conv
        with L.device_guard(dev):
            L.check(entry(ptrs, ny, nx, nl, bs, na, nc, L.ptr(assign.idx), L.ptr(assign.tbox),
                                              L.ptr(assign.anch), L.ptr(assign.tcls), L.ptr(assign.counts), assign.cap,
                                              ctypes.byref(hyp), L.ptr(out), L.ptr(ws), ws_bytes, L.stream()),
                    "dh_yolo_loss_fwd_conv")
        return out, ws
    logits = objectness_logits(maps)
    with L.device_guard(dev):
        L.check(lib.dh_yolo_loss_fwd(ptrs, ny, nx, nl, bs, na, nc, L.ptr(assign.idx), L.ptr(assign.tbox),
                                     L.ptr(assign.anch), L.ptr(assign.tcls), L.ptr(assign.counts), assign.cap,
                                     ctypes.byref(hyp), L.ptr(logits), L.ptr(out), L.ptr(ws), ws_bytes, L.stream()),
                "dh_yolo_loss_fwd")
    return out, ws


def _loss_bwd(assign, hyp, nc, maps, ws, g_loss, conv_na=0, level_scale=None, half_grad=True):
    """Launch the backward kernels for upstream gradient g_loss (1 float on the device); returns grads
    (in the layout of `maps`: convolution layout when conv_na = na, and then in the maps' element type: fp16 convolution
    outputs get the fp16 gradient their convolution's backward takes, unless half_grad is False).
    level_scale: (nl, 2) fp32 on the device, a factor per level on the box / class terms and on the objectness term."""
    nl = len(maps)
    bs, na = int(maps[0].shape[0]), conv_na or int(maps[0].shape[1])
    half_in = bool(conv_na) and maps[0].dtype == torch.float16
    half = half_in and half_grad
    grads = [torch.empty_like(x, dtype=torch.float16 if half else torch.float32) for x in maps]
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in maps])
    gptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in grads])
    head = (ptrs, L.int_array(assign.ny), L.int_array(assign.nx), nl, bs, na, nc, L.ptr(assign.idx), L.ptr(assign.tbox),
            L.ptr(assign.anch), L.ptr(assign.tcls), L.ptr(assign.counts), assign.cap, ctypes.byref(hyp), L.ptr(g_loss))
    tail = (gptrs, L.ptr(ws), ws.numel(), L.stream())
    with L.device_guard(maps[0].device):
        if level_scale is not None:
            layout = 0 if not conv_na else (3 if half else 2 if half_in else 1)
            L.check(L.lib().dh_yolo_loss_bwd_scaled(layout, *head, L.ptr(level_scale), *tail), "dh_yolo_loss_bwd_scaled")
        elif half_in and not half:
            L.check(L.lib().dh_yolo_loss_bwd_conv_f16(*head, 0, *tail), "dh_yolo_loss_bwd_conv_f16")
        elif half:
            L.check(L.lib().dh_yolo_loss_bwd_conv_f16(*head, 1, *tail), "dh_yolo_loss_bwd_conv_f16")
        elif conv_na:
            L.check(L.lib().dh_yolo_loss_bwd_conv(*head, *tail), "dh_yolo_loss_bwd_conv")
        else:
            L.check(L.lib().dh_yolo_loss_bwd(*head, *tail), "dh_yolo_loss_bwd")
    return grads


_ONES = {}


def _one(device):
    """A resident 1.0 per device: the default upstream gradient."""
    hit = _ONES.get(device)
    if hit is None:
        hit = _ONES[device] = torch.ones(1, dtype=torch.float32, device=device)
    return hit


def _loss_fused(shapes, targets, anchors, anchor_t, hyp, nc, maps, grad_loss, conv=False):
    """Assignment, loss forward and backward through the single fused entry; returns (assignment, out, grads, sums).
    conv: `maps` are convolution outputs (bs, na*no, ny, nx) and so are the gradients."""
    tg = L.dev_f32(targets, "targets").contiguous()
    if tg.dim() != 2 or tg.shape[1] != 6:
        raise RuntimeError(f"targets must be (nt,6) [image,class,x,y,w,h], got {tuple(targets.shape)}")
    anc = L.dev_f32(anchors, "anchors").contiguous()
    nl, na = int(anc.shape[0]), int(anc.shape[1])
    dev = maps[0].device
    A = _Assignment(shapes, tg.shape[0], na, nl, dev)
    bs = _check_levels(maps, A, nc, na, conv)
    lib = L.lib()
    ny, nx = L.int_array(A.ny), L.int_array(A.nx)
    aws_bytes = lib.dh_build_targets_workspace(nl, na, A.nt)
    ws_bytes = lib.dh_yolo_loss_workspace(ny, nx, nl, bs, na, nc, A.cap)
    if ws_bytes == 0:
        raise RuntimeError("ComputeLoss: unsupported problem size (a level exceeds 2^32 elements)")
    aws, ws = L.workspace(aws_bytes, dev), L.workspace(ws_bytes, dev)
    out = torch.empty(6 + nl, dtype=torch.float32, device=dev)
    grads = [torch.empty_like(x, dtype=torch.float32) for x in maps]
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in maps])
    gptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in grads])
    if conv:
        head = (L.ptr(tg), A.nt, L.ptr(anc), float(anchor_t), ptrs, ny, nx, nl, bs, na, nc, L.ptr(A.idx), L.ptr(A.tbox),
                L.ptr(A.anch), L.ptr(A.tcls), L.ptr(A.counts), A.cap, ctypes.byref(hyp), L.ptr(grad_loss))
        tail = (gptrs, L.ptr(out), L.ptr(aws), aws_bytes, L.ptr(ws), ws_bytes, L.stream())
        with L.device_guard(dev):
            if maps[0].dtype == torch.float16:      # fp32 gradients: the caller takes them as they are
                L.check(lib.dh_yolo_loss_fwd_bwd_conv_f16(*head, 0, *tail), "dh_yolo_loss_fwd_bwd_conv_f16")
            else:
                L.check(lib.dh_yolo_loss_fwd_bwd_conv(*head, *tail), "dh_yolo_loss_fwd_bwd_conv")
        return A, out, grads, _sums_view(ws, A, bs, na, nc, nl)
    logits = objectness_logits(maps)
    with L.device_guard(dev):
        L.check(lib.dh_yolo_loss_fwd_bwd(L.ptr(tg), A.nt, L.ptr(anc), float(anchor_t), ptrs, ny, nx, nl, bs, na, nc,
                                         L.ptr(A.idx), L.ptr(A.tbox), L.ptr(A.anch), L.ptr(A.tcls), L.ptr(A.counts), A.cap,
                                         ctypes.byref(hyp), L.ptr(logits), L.ptr(grad_loss), gptrs, L.ptr(out), L.ptr(aws),
                                         aws_bytes, L.ptr(ws), ws_bytes, L.stream()), "dh_yolo_loss_fwd_bwd")
    return A, out, grads, _sums_view(ws, A, bs, na, nc, nl)


class _Positives:
    """What ComputeLoss.positives leaves behind for forward_backward: the assignment, the loss workspace with the
    positives' forward results in it, and the stream it was made on."""
    __slots__ = ("assign", "ws", "aws", "hyp", "maps_key", "made_on")

    def borrow(self):
        """About to be used on the current stream: keep the allocator from recycling the buffers before that stream is done."""
        dev = self.ws.device
        if L._raw_stream(dev.index) != self.made_on:
            cur = torch.cuda.current_stream(dev)
            A = self.assign
            for t in (self.ws, self.aws, A.idx, A.tbox, A.anch, A.tcls, A.counts):
                t.record_stream(cur)


def _loss_positives(shapes, targets, anchors, anchor_t, hyp, nc, maps):
    """First half of _loss_fused: assignment and the forward of the assigned rows (reads only targets and maps)."""
    tg = L.dev_f32(targets, "targets").contiguous()
    if tg.dim() != 2 or tg.shape[1] != 6:
        raise RuntimeError(f"targets must be (nt,6) [image,class,x,y,w,h], got {tuple(targets.shape)}")
    anc = L.dev_f32(anchors, "anchors").contiguous()
    nl, na = int(anc.shape[0]), int(anc.shape[1])
    dev = maps[0].device
    A = _Assignment(shapes, tg.shape[0], na, nl, dev)
    bs = int(maps[0].shape[0])
    for x, ny_, nx_ in zip(maps, A.ny, A.nx):
        if tuple(x.shape) != (bs, na, ny_, nx_, nc + 5):
            raise RuntimeError(f"ComputeLoss: level shape {tuple(x.shape)} does not match (bs,na,ny,nx,5+nc)")
    lib = L.lib()
    ny, nx = L.int_array(A.ny), L.int_array(A.nx)
    aws_bytes = lib.dh_build_targets_workspace(nl, na, A.nt)
    ws_bytes = lib.dh_yolo_loss_workspace(ny, nx, nl, bs, na, nc, A.cap)
    if ws_bytes == 0:
        raise RuntimeError("ComputeLoss: unsupported problem size (a level exceeds 2^32 elements)")
    st = _Positives()
    st.assign, st.hyp = A, hyp
    st.aws, st.ws = L.workspace(aws_bytes, dev), L.workspace(ws_bytes, dev)
    st.maps_key = [(x.data_ptr(), tuple(x.shape)) for x in maps]
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in maps])
    with L.device_guard(dev):
        st.made_on = L.raw_stream()
        L.check(lib.dh_yolo_loss_positives(L.ptr(tg), A.nt, L.ptr(anc), float(anchor_t), ptrs, ny, nx, nl, bs, na, nc,
                                           L.ptr(A.idx), L.ptr(A.tbox), L.ptr(A.anch), L.ptr(A.tcls), L.ptr(A.counts), A.cap,
                                           ctypes.byref(hyp), L.ptr(st.aws), aws_bytes, L.ptr(st.ws), ws_bytes,
                                           L.c_void_p(st.made_on)), "dh_yolo_loss_positives")
    return st


def _loss_dense(st, nc, maps, grad_loss):
    """Second half of _loss_fused on the current stream; returns (out, grads)."""
    if st.maps_key != [(x.data_ptr(), tuple(x.shape)) for x in maps]:
        raise RuntimeError("positives= was not produced by ComputeLoss.positives for these maps")
    st.borrow()
    A = st.assign
    nl = len(maps)
    bs, na = int(maps[0].shape[0]), int(maps[0].shape[1])
    dev = maps[0].device
    out = torch.empty(6 + nl, dtype=torch.float32, device=dev)
    grads = [torch.empty_like(x) for x in maps]
    ptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in maps])
    gptrs = (L.c_void_p * nl)(*[x.data_ptr() for x in grads])
    logits = objectness_logits(maps)
    with L.device_guard(dev):
        L.check(L.lib().dh_yolo_loss_dense(ptrs, L.int_array(A.ny), L.int_array(A.nx), nl, bs, na, nc, L.ptr(A.idx), L.ptr(A.tbox),
                                           L.ptr(A.anch), L.ptr(A.tcls), L.ptr(A.counts), A.cap, ctypes.byref(st.hyp),
                                           L.ptr(logits), L.ptr(grad_loss), gptrs, L.ptr(out), L.ptr(st.ws), st.ws.numel(),
                                           L.stream()), "dh_yolo_loss_dense")
    return out, grads


class _YoloLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, assign, hyp, nc, *p):
        convs = conv_layout_of(p)
        if convs is not None:       # lazy maps: the kernels work on the convolution outputs themselves
            na = int(p[0].shape[1])
            out, ws = _loss_fwd(assign, hyp, nc, convs, conv_na=na)
            assign.sums = _sums_view(ws, assign, int(convs[0].shape[0]), na, nc, len(convs))
            ctx.save_for_backward(*convs)
            ctx.state = (assign, hyp, nc, ws, [x.dtype for x in p])
            ctx.conv_na = na
            loss, rest = out[0:1], out[1:]
            ctx.mark_non_differentiable(rest)
            return loss, rest
        ctx.conv_na = 0
        maps = [L.dev_f32(x, f"p[{i}]").contiguous() for i, x in enumerate(p)]
        out, ws = _loss_fwd(assign, hyp, nc, maps)
        assign.sums = _sums_view(ws, assign, int(maps[0].shape[0]), int(maps[0].shape[1]), nc, len(maps))
        ctx.save_for_backward(*maps)
        ctx.state = (assign, hyp, nc, ws, [x.dtype for x in p])
        loss, rest = out[0:1], out[1:]      # [lbox, lobj, lcls, n_pos, obji per level..., ignored rows]
        ctx.mark_non_differentiable(rest)
        return loss, rest

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        maps = ctx.saved_tensors
        assign, hyp, nc, ws, dtypes = ctx.state
        if ctx.conv_na:     # gradients in the convolution layout, handed back as views of the maps' shape: the permute and
            # view nodes between the convolution and the loss pass them on without a copy
            grads = _loss_bwd(assign, hyp, nc, maps, ws, g_loss.reshape(1).float().contiguous(), conv_na=ctx.conv_na)
            return (None, None, None) + tuple(_as_map_views(grads, ctx.conv_na))
        grads = _loss_bwd(assign, hyp, nc, maps, ws, g_loss.reshape(1).float().contiguous())
        return (None, None, None) + tuple(gr if dt == torch.float32 else gr.to(dt) for gr, dt in zip(grads, dtypes))


class ComputeLoss:
    """Same constructor, attributes and call contract as loss.py:91-167.

    __call__(p, targets) -> (loss[1], loss_items[3] detached); build_targets(p, targets) ->
    (tcls, tbox, indices, anch).  __call__ keeps the assignment on the device (no host copy, no
    data-dependent shapes); build_targets, whose return type has data-dependent shapes, copies the
    nl row counts to the host once.
    """

    def __init__(self, model, autobalance=False):
        self.sort_obj_iou = False
        self.last_positive_counts = None
        self.last_stats = None
        self._last_sums = None
        h = model.hyp
        self.cp, self.cn = smooth_BCE(eps=h.get('label_smoothing', 0.0))
        det = model.module.model[-1] if is_parallel(model) else model.model[-1]
        self.balance = {3: [4.0, 1.0, 0.4]}.get(det.nl, [4.0, 1.0, 0.25, 0.06, 0.02])
        self.ssi = list(det.stride).index(16) if autobalance else 0
        self.gr, self.hyp, self.autobalance = 1.0, h, autobalance
        for k in 'na', 'nc', 'nl', 'anchors':
            setattr(self, k, getattr(det, k))

    @property
    def last_sums(self):
        """(nl, 5) float64 on the device, written by the last call's kernels: per level [sum(1 - iou), sum class BCE, sum
        objectness BCE, assigned rows, cells].  A view of that call's workspace."""
        src = self._last_sums
        return src.view() if isinstance(src, _Sums) else src

    @last_sums.setter
    def last_sums(self, value):
        self._last_sums = value

    def _hyp_struct(self):
        h = self.hyp
        s = L.YoloHyp()
        s.box, s.obj, s.cls = float(h['box']), float(h['obj']), float(h['cls'])
        s.cls_pw, s.obj_pw, s.fl_gamma = float(h['cls_pw']), float(h['obj_pw']), float(h['fl_gamma'])
        s.cp, s.cn, s.gr = float(self.cp), float(self.cn), float(self.gr)
        for i, b in enumerate(self.balance[:self.nl]):
            s.balance[i] = float(b)
        s.sort_obj_iou = int(bool(self.sort_obj_iou))                           # loss.py:94,140-142
        return s

    def assign(self, p, targets):
        """Target assignment only (loss.py:169-222), kept on the device: the opaque result can be handed to
        __call__ / forward_backward as `assignment=` so that this short, latency-bound kernel can be enqueued early
        (e.g. on another stream, beside the kernels that produce or decode `p`).  p may be the maps or their
        (ny, nx) shapes."""
        shapes = [(x.shape[2], x.shape[3]) if isinstance(x, torch.Tensor) else (int(x[0]), int(x[1])) for x in p]
        return _assign(shapes, targets, self.anchors, self.hyp['anchor_t'])

    def _assignment_for(self, shapes, targets, assignment):
        if assignment is None:
            return _assign(shapes, targets, self.anchors, self.hyp['anchor_t'])
        if not isinstance(assignment, _Assignment) or list(zip(assignment.ny, assignment.nx)) != [tuple(s) for s in shapes] \
                or assignment.nt != int(targets.shape[0]):
            raise RuntimeError("assignment= was not produced by ComputeLoss.assign for these maps and targets")
        assignment.borrow()
        return assignment

    def __call__(self, p, targets, assignment=None):
        shapes = [(int(x.shape[2]), int(x.shape[3])) for x in p]
        assign = self._assignment_for(shapes, targets, assignment)
        self.last_positive_counts = assign.counts            # (nl,) int32 on the device: rows assigned per level
        loss, rest = _YoloLoss.apply(assign, self._hyp_struct(), int(self.nc), *p)
        self.last_stats = rest[0:4]                                             # [lbox, lobj, lcls, n_pos] on the device
        self.last_sums = assign.sums                                            # (nl, 5) float64: the same before the divisions
        if self.autobalance:                                                    # loss.py:157-161 (host round trip)
            obji = rest[4:4 + self.nl].tolist()
            self.balance = [b * 0.9999 + 0.0001 / o for b, o in zip(self.balance, obji)]
            self.balance = [x / self.balance[self.ssi] for x in self.balance]
        return loss, rest[0:3]

    def positives(self, p, targets):
        """First half of forward_backward, for callers that can start it early: target assignment and the forward of
        the assigned rows, which read only the targets and the raw maps.  Enqueue it on a side stream beside whatever
        produces or decodes `p`, make the main stream wait for it (an event), and hand the result to
        forward_backward(..., positives=...), which then only runs the dense objectness pass and the backward."""
        maps = [L.dev_f32(x.detach(), f"p[{i}]").contiguous() for i, x in enumerate(p)]
        shapes = [(int(x.shape[2]), int(x.shape[3])) for x in maps]
        return _loss_positives(shapes, targets, self.anchors, self.hyp['anchor_t'], self._hyp_struct(), int(self.nc), maps)

    def forward_backward(self, p, targets, grad_loss=None, assignment=None, positives=None):
        """Loss and its gradients in one call, without the autograd engine: (loss[1], loss_items[3], [dL/dp_i]).

        The arithmetic of __call__ followed by backward() in five launches instead of nine; grad_loss (1 float on
        the device, default 1) is the upstream gradient of the loss.  Everything stays on the calling stream with no
        host synchronisation and no hop to the autograd engine's device thread.
        """
        convs = conv_layout_of(p) if assignment is None and positives is None else None
        if convs is not None:       # lazy maps (see conv_layout_of): gradients come back as views of the same kind
            shapes = [(int(x.shape[2]), int(x.shape[3])) for x in p]
            g = _one(convs[0].device) if grad_loss is None else grad_loss
            assign, out, grads, sums = _loss_fused(shapes, targets, self.anchors, self.hyp['anchor_t'], self._hyp_struct(), int(self.nc),
                                                   [c.detach() for c in convs], g.reshape(1).float().contiguous(), conv=True)
            self.last_positive_counts = assign.counts
            self.last_stats, self.last_sums = out[1:5], sums
            self._autobalance_step(out)
            return out[0:1], out[1:4], _as_map_views(grads, int(p[0].shape[1]))
        maps = [L.dev_f32(x.detach(), f"p[{i}]").contiguous() for i, x in enumerate(p)]
        shapes = [(int(x.shape[2]), int(x.shape[3])) for x in maps]
        hyp = self._hyp_struct()
        if grad_loss is None:
            grad_loss = _one(maps[0].device)
        grad_loss = grad_loss.reshape(1).float().contiguous()
        bs, na, nl = int(maps[0].shape[0]), int(maps[0].shape[1]), len(maps)
        if positives is not None:       # assignment and positives' forward already enqueued (possibly on another stream)
            out, grads = _loss_dense(positives, int(self.nc), maps, grad_loss)
            assign = positives.assign
            sums = _sums_view(positives.ws, assign, bs, na, int(self.nc), nl)
        elif assignment is not None:    # an assignment made earlier (possibly on another stream): the separate entries
            assign = self._assignment_for(shapes, targets, assignment)
            out, ws = _loss_fwd(assign, hyp, int(self.nc), maps)
            grads = _loss_bwd(assign, hyp, int(self.nc), maps, ws, grad_loss)
            sums = _sums_view(ws, assign, bs, na, int(self.nc), nl)
        else:
            assign, out, grads, sums = _loss_fused(shapes, targets, self.anchors, self.hyp['anchor_t'], hyp, int(self.nc), maps, grad_loss)
        self.last_positive_counts = assign.counts
        self.last_stats, self.last_sums = out[1:5], sums
        self._autobalance_step(out)
        return out[0:1], out[1:4], grads

    def items_from_sums(self, sums):
        """[lbox, lobj, lcls, n_pos] (fp32, after the gains, as loss.py:162-167 returns them) from a (nl, 5) block of sums:
        this rank's `last_sums`, or the sum of every rank's -- then the means are those of the whole data-parallel batch,
        which a mean of per-rank means only equals when every rank holds the same number of assigned rows."""
        s = sums.double()
        n, cells = s[:, 3], s[:, 4]
        has = n > 0
        safe = torch.where(has, n, torch.ones_like(n))
        lbox = torch.where(has, s[:, 0] / safe, torch.zeros_like(n)).sum() * float(self.hyp['box'])                # loss.py:136,162
        lcls = torch.where(has, s[:, 1] / (safe * float(self.nc)), torch.zeros_like(n)).sum() * float(self.hyp['cls'])   # :149,164
        bal = torch.tensor([float(b) for b in self.balance[:s.shape[0]]], dtype=torch.float64, device=s.device)
        lobj = (s[:, 2] / cells * bal).sum() * float(self.hyp['obj'])                                              # :155-156,163
        return torch.stack((lbox, lobj, lcls, n.sum())).float()

    def forward_backward_global(self, p, targets, group=None, grad_loss=None):
        """forward_backward with the loss normalised over the WHOLE data-parallel batch instead of per rank (opt-in; the
        reference, and forward_backward, normalise per rank: loss.py:136 with train.py:331-332).

        Every rank calls it on its own image shard.  Between forward and backward the per-level sums and divisors are
        all-reduced (5*nl doubles, on the calling stream); the backward then scales the terms of level l by
        world * local divisor / global divisor, so that the gradients DDP averages (after the caller's `loss *= WORLD_SIZE`,
        train.py:331-332, which stays as it is) are those of the global loss.  Returns (loss[1], items[4], grads): items =
        [lbox, lobj, lcls] of the whole batch and its number of assigned rows, loss = their sum * this rank's batch size --
        what the reference's expression gives when its means run over all ranks.  Without a process group the result is
        forward_backward's.  autobalance keeps following this rank's own objectness means."""
        import torch.distributed as dist
        convs = conv_layout_of(p)
        maps = [c.detach() for c in convs] if convs is not None else [L.dev_f32(x.detach(), f"p[{i}]").contiguous() for i, x in enumerate(p)]
        na = int(p[0].shape[1])
        shapes = [(int(x.shape[2]), int(x.shape[3])) for x in p]
        hyp, nc, nl, bs = self._hyp_struct(), int(self.nc), len(maps), int(maps[0].shape[0])
        assign = _assign(shapes, targets, self.anchors, self.hyp['anchor_t'])
        out, ws = _loss_fwd(assign, hyp, nc, maps, conv_na=na if convs is not None else 0)
        local = _sums_view(ws, assign, bs, na, nc, nl).view()
        total, world = local.clone(), 1
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(group)
            if dist.get_backend(group) == "gloo":         # host-side rehearsal backend
                host = total.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                total = host.to(local.device)
            else:
                dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        ones = torch.ones_like(total[:, 3])
        f_pos = torch.where(total[:, 3] > 0, world * local[:, 3] / torch.where(total[:, 3] > 0, total[:, 3], ones), ones)
        f_obj = world * local[:, 4] / total[:, 4]
        level_scale = torch.stack((f_pos, f_obj), 1).float().contiguous()
        g = (_one(maps[0].device) if grad_loss is None else grad_loss).reshape(1).float().contiguous()
        grads = _loss_bwd(assign, hyp, nc, maps, ws, g, conv_na=na if convs is not None else 0, level_scale=level_scale, half_grad=False)
        items = self.items_from_sums(total)
        self.last_positive_counts, self.last_stats, self.last_sums = assign.counts, out[1:5], local
        self._autobalance_step(out)
        loss = (items[0:3].sum() * bs).reshape(1)
        return loss, items, (_as_map_views(grads, na) if convs is not None else grads)

    def _autobalance_step(self, out):
        """loss.py:157-161: the balance weights follow the per-level objectness means (one host round trip, like the
        reference's .item()); out = [loss, lbox, lobj, lcls, n_pos, per-level objectness mean..., ignored rows]."""
        if self.autobalance:
            obji = out[5:5 + self.nl].tolist()
            self.balance = [b * 0.9999 + 0.0001 / o for b, o in zip(self.balance, obji)]
            self.balance = [x / self.balance[self.ssi] for x in self.balance]

    def build_targets(self, p, targets):
        shapes = [(x.shape[2], x.shape[3]) for x in p]
        A = _assign(shapes, targets, self.anchors, self.hyp['anchor_t'])
        counts = A.counts.tolist()
        tcls, tbox, indices, anch = [], [], [], []
        for l, n in enumerate(counts):
            indices.append(tuple(A.idx[l, k, :n] for k in range(4)))            # image, anchor, gridy, gridx
            tbox.append(A.tbox[l, :n])
            anch.append(A.anch[l, :n])
            tcls.append(A.tcls[l, :n])
        return tcls, tbox, indices, anch
