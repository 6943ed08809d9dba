"""Drop-ins for object_detection/loss.py: LabelSmoothingFocalLoss (:8-83), IoULossWithMask and bbox_iou on [...,4] (:102-166)."""
import torch
import torch.nn as nn

from . import _lib as L
from .metrics import _PairIoU, _mode

_EPS = 1e-7   # default eps of loss.py:118, the only value IoULossWithMask can reach


def bbox_iou(box1, box2, x1y1x2y2=False, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """IoU family over the last dimension, centre/size boxes by default (loss.py:118-166)."""
    b1, b2 = torch.broadcast_tensors(box1, box2)
    lead = b1.shape[:-1]
    out = _PairIoU.apply(b1.reshape(-1, 4), b2.reshape(-1, 4), 1, 1, x1y1x2y2, _mode(GIoU, DIoU, CIoU), eps)
    return out.view(lead)


def _plane_view(t, H, W):
    """(B,2,W,H) planes usable in place: unit stride cells, contiguous planes, aligned; else a copy."""
    ok = (t.stride(3) == 1 and t.stride(2) == H and t.stride(1) == W * H and t.stride(0) % 4 == 0
          and t.stride(0) >= 2 * W * H and t.data_ptr() % 16 == 0)
    return t if ok else t.contiguous()


def _check_planes(pxy, pwh, txy, twh, m):
    B, two, W, H = pxy.shape
    for t in (pxy, pwh, txy, twh):
        if tuple(t.shape) != (B, 2, W, H):
            raise RuntimeError(f"masked CIoU loss: expected (B,2,W,H) planes like {(B, 2, W, H)}, got {tuple(t.shape)}")
    if tuple(m.shape) != (B, W, H):
        raise RuntimeError(f"masked CIoU loss: mask must be (B,W,H)={(B, W, H)}, got {tuple(m.shape)}")
    if (W * H) % 4:
        raise RuntimeError("masked CIoU loss: W*H must be a multiple of 4")
    return B, W, H


def _loss_forward(pxy, pwh, txy, twh, m, add_coords):
    B, W, H = _check_planes(pxy, pwh, txy, twh, m)
    lib = L.lib()
    ws_bytes = lib.dh_centernet_box_loss_workspace(B, W, H)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pxy.device)
    out2 = torch.empty(2, dtype=torch.float32, device=pxy.device)
    with L.device_guard(pxy.device):
        L.check(lib.dh_centernet_box_loss_fwd(L.ptr(pxy), pxy.stride(0), L.ptr(pwh), pwh.stride(0), L.ptr(txy),
                                              txy.stride(0), L.ptr(twh), twh.stride(0), L.ptr(m), B, W, H,
                                              int(add_coords), _EPS, L.ptr(out2), L.ptr(ws), ws_bytes, L.stream()),
                "dh_centernet_box_loss_fwd")
    return out2


def _loss_backward(pxy, pwh, txy, twh, m, add_coords, out2, grad_out, gxy, gwh):
    B, _, W, H = pxy.shape
    g = grad_out.reshape(1).float().contiguous()
    with L.device_guard(pxy.device):
        L.check(L.lib().dh_centernet_box_loss_bwd(L.ptr(pxy), pxy.stride(0), L.ptr(pwh), pwh.stride(0), L.ptr(txy),
                                                  txy.stride(0), L.ptr(twh), twh.stride(0), L.ptr(m), B, W, H,
                                                  int(add_coords), _EPS, L.ptr(out2), L.ptr(g), L.ptr(gxy),
                                                  gxy.stride(0), L.ptr(gwh), gwh.stride(0), L.stream()),
                "dh_centernet_box_loss_bwd")


class _MaskedCIoULossPlanes(torch.autograd.Function):
    """1 - sum(CIoU(pred*m, tgt*m)) / (4*sum(m) + 1e-4); centre and size given as separate (B,2,W,H) maps."""

    @staticmethod
    def forward(ctx, pred_xy, pred_wh, tgt_xy, tgt_wh, mask, add_coords):
        B, _, W, H = pred_xy.shape
        pxy, pwh, txy, twh = (_plane_view(L.dev_f32(t, n), H, W) for t, n in (
            (pred_xy, "pred_xy"), (pred_wh, "pred_wh"), (tgt_xy, "tgt_xy"), (tgt_wh, "tgt_wh")))
        m = L.dev_f32(mask, "mask").contiguous()
        out2 = _loss_forward(pxy, pwh, txy, twh, m, add_coords)
        ctx.save_for_backward(pxy, pwh, txy, twh, m, out2)
        ctx.add_coords = bool(add_coords)
        return out2[0]

    @staticmethod
    def backward(ctx, grad_out):
        pxy, pwh, txy, twh, m, out2 = ctx.saved_tensors
        if any(ctx.needs_input_grad[2:5]):
            raise RuntimeError("masked CIoU loss: gradients are provided for the predictions only")
        gxy, gwh = torch.empty_like(pxy, memory_format=torch.contiguous_format), torch.empty_like(
            pwh, memory_format=torch.contiguous_format)
        _loss_backward(pxy, pwh, txy, twh, m, ctx.add_coords, out2, grad_out, gxy, gwh)
        return gxy, gwh, None, None, None, None


class _MaskedCIoULossPacked(torch.autograd.Function):
    """Same loss with (B,4,W,H) inputs (centre planes then size planes), as IoULossWithMask receives them."""

    @staticmethod
    def forward(ctx, pred, target, mask):
        p = L.dev_f32(pred, "pred").contiguous()
        t = L.dev_f32(target, "target").contiguous()
        m = L.dev_f32(mask, "mask").contiguous()
        if p.dim() != 4 or p.shape[1] != 4 or t.shape != p.shape:
            raise RuntimeError(f"IoULossWithMask: expected (B,4,W,H) pred/target, got {tuple(pred.shape)} / {tuple(target.shape)}")
        out2 = _loss_forward(p[:, 0:2], p[:, 2:4], t[:, 0:2], t[:, 2:4], m, False)
        ctx.save_for_backward(p, t, m, out2)
        return out2[0]

    @staticmethod
    def backward(ctx, grad_out):
        p, t, m, out2 = ctx.saved_tensors
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("IoULossWithMask: gradients are provided for the predictions only")
        grad = torch.empty_like(p)
        _loss_backward(p[:, 0:2], p[:, 2:4], t[:, 0:2], t[:, 2:4], m, False, out2, grad_out, grad[:, 0:2], grad[:, 2:4])
        return grad, None, None


def masked_ciou_loss(pred_xy, pred_wh, tgt_xy, tgt_wh, mask, add_coords=False):
    return _MaskedCIoULossPlanes.apply(pred_xy, pred_wh, tgt_xy, tgt_wh, mask, add_coords)


def _loss_forward_backward(pxy, pwh, txy, twh, m, add_coords, grad_loss, gxy, gwh):
    B, W, H = _check_planes(pxy, pwh, txy, twh, m)
    lib = L.lib()
    ws_bytes = lib.dh_centernet_box_loss_workspace(B, W, H)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pxy.device)
    out2 = torch.empty(2, dtype=torch.float32, device=pxy.device)
    g = None if grad_loss is None else grad_loss.reshape(1).float().contiguous()
    with L.device_guard(pxy.device):
        L.check(lib.dh_centernet_box_loss_fwd_bwd(L.ptr(pxy), pxy.stride(0), L.ptr(pwh), pwh.stride(0), L.ptr(txy), txy.stride(0),
                                                  L.ptr(twh), twh.stride(0), L.ptr(m), B, W, H, int(add_coords), _EPS, L.ptr(g),
                                                  L.ptr(out2), L.ptr(gxy), gxy.stride(0), L.ptr(gwh), gwh.stride(0), L.ptr(ws),
                                                  ws_bytes, L.stream()), "dh_centernet_box_loss_fwd_bwd")
    return out2[0]


def masked_ciou_loss_forward_backward(pred_xy, pred_wh, tgt_xy, tgt_wh, mask, add_coords=False, grad_loss=None):
    """masked_ciou_loss and its gradients in one call of two launches, without the autograd engine:
    (loss, d loss / d pred_xy, d loss / d pred_wh).  grad_loss: upstream gradient of the loss (1 float on the device,
    default 1).  Same values, bit for bit, as the loss followed by backward()."""
    B, _, W, H = pred_xy.shape
    pxy, pwh, txy, twh = (_plane_view(L.dev_f32(t.detach(), n), H, W) for t, n in (
        (pred_xy, "pred_xy"), (pred_wh, "pred_wh"), (tgt_xy, "tgt_xy"), (tgt_wh, "tgt_wh")))
    m = L.dev_f32(mask, "mask").contiguous()
    gxy = torch.empty((B, 2, W, H), dtype=torch.float32, device=pxy.device)
    gwh = torch.empty((B, 2, W, H), dtype=torch.float32, device=pxy.device)
    return _loss_forward_backward(pxy, pwh, txy, twh, m, add_coords, grad_loss, gxy, gwh), gxy, gwh


class IoULossWithMask(nn.Module):
    """loss.py:102-115.  As in the reference the constructor flags are stored but CIoU is always used."""

    def __init__(self, GIoU=False, DIoU=False, CIoU=False):
        super().__init__()
        self.GIoU = GIoU
        self.DIoU = DIoU
        self.CIoU = CIoU

    def forward(self, pred, target, mask):
        return _MaskedCIoULossPacked.apply(pred, target, mask)

    def forward_backward(self, pred, target, mask, grad_loss=None):
        """forward(pred, target, mask) and d loss / d pred in one call of two launches, without the autograd engine (a
        training step's criterion["box"](...) + backward(), object_detection/detectors/centernet/loss.py:24-28):
        (loss, gradient of pred's shape).  Same values, bit for bit."""
        p = L.dev_f32(pred.detach(), "pred").contiguous()
        t = L.dev_f32(target, "target").contiguous()
        m = L.dev_f32(mask, "mask").contiguous()
        if p.dim() != 4 or p.shape[1] != 4 or t.shape != p.shape:
            raise RuntimeError(f"IoULossWithMask: expected (B,4,W,H) pred/target, got {tuple(pred.shape)} / {tuple(target.shape)}")
        grad = torch.empty_like(p)
        loss = _loss_forward_backward(p[:, 0:2], p[:, 2:4], t[:, 0:2], t[:, 2:4], m, False, grad_loss, grad[:, 0:2], grad[:, 2:4])
        return loss, grad


class _FocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, one_hot, gamma, smoothing, ignore_index):
        x = L.dev_f32(logits, "logits").contiguous()                      # loss.py:42 (.float())
        N, C = int(x.shape[0]), int(x.shape[1])
        S = 1
        for d in x.shape[2:]:
            S *= int(d)
        if not isinstance(label, torch.Tensor) or not label.is_cuda:
            raise RuntimeError("LabelSmoothingFocalLoss: the label must be a tensor on the logits' ROCm device (no CPU fallback)")
        # .long() (loss.py:43) happens on load: int64 labels are read as they are, everything else as fp32 and truncated
        y = (label if label.dtype == torch.int64 else label.float()).contiguous()
        want = tuple(x.shape) if one_hot else (N,)
        if tuple(y.shape) != want:
            raise RuntimeError(f"LabelSmoothingFocalLoss: label shape {tuple(y.shape)} does not match {want}")
        lib = L.lib()
        ws_bytes = lib.dh_focal_loss_workspace(N * S)
        ws = L.workspace(ws_bytes, x.device)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        args = (N, C, S, int(one_hot), int(y.dtype == torch.int64), float(gamma), float(smoothing),
                int(ignore_index is not None), int(ignore_index) if ignore_index is not None else 0)
        with L.device_guard(x.device):
            L.check(lib.dh_focal_loss_fwd(L.ptr(x), L.ptr(y), *args, L.ptr(out), L.ptr(ws), ws_bytes, L.stream()),
                    "dh_focal_loss_fwd")
        ctx.save_for_backward(x, y, ws)
        ctx.args, ctx.dtype = args, logits.dtype
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        x, y, ws = ctx.saved_tensors
        g = grad_out.reshape(1).float().contiguous()
        grad = torch.empty_like(x)
        with L.device_guard(x.device):
            L.check(L.lib().dh_focal_loss_bwd(L.ptr(x), L.ptr(y), *ctx.args, L.ptr(g), L.ptr(grad), L.ptr(ws), L.stream()),
                    "dh_focal_loss_bwd")
        return (grad if ctx.dtype == torch.float32 else grad.to(ctx.dtype)), None, None, None, None, None


class LabelSmoothingFocalLoss(nn.Module):
    """loss.py:8-83: the CenterNet class loss (object_detection/train.py:147-153 builds it with
    one_hot_label_format=True, gamma=8).  Same constructor, same checks; as in the reference alpha and size_average are
    stored and unused and the result is always the mean over cells.

    One-hot labels are the (N, C, ...) target maps; index labels (N,) go with (N, C) logits (with more axes the
    reference's own broadcast fails).  One forward launch streams the label maps once and touches the logits only at
    labelled cells; the backward writes the dense gradient without reading the labels again.
    """

    def __init__(self, num_classes, one_hot_label_format=False, gamma=0, alpha=None, smoothing=0.0, size_average=True,
                 ignore_index=None):
        super().__init__()
        self._num_classes = num_classes
        self.one_hot_label_format = one_hot_label_format
        self._gamma = gamma
        self._alpha = alpha
        self._smoothing = smoothing
        self._size_average = size_average
        self._ignore_index = ignore_index
        if self._num_classes <= 1:
            raise ValueError("The number of classes must be 2 or higher")
        if self._gamma < 0:
            raise ValueError("Gamma must be 0 or higher")

    def forward(self, logits, label):
        if logits.dim() < 2 or logits.dim() == 3:
            raise RuntimeError("LabelSmoothingFocalLoss: logits must be (N, C) or (N, C, d1, d2, ...): with three axes the "
                               "reference's softmax (loss.py:78) and log_softmax (loss.py:64) run over different axes")
        if not self.one_hot_label_format and logits.dim() != 2:
            raise RuntimeError("LabelSmoothingFocalLoss: index labels need (N, C) logits (loss.py:72-78 cannot broadcast others)")
        if logits.shape[1] != self._num_classes and not self.one_hot_label_format:
            raise RuntimeError(f"LabelSmoothingFocalLoss: {logits.shape[1]} logits per cell for {self._num_classes} classes")
        return _FocalLoss.apply(logits, label, self.one_hot_label_format, self._gamma, self._smoothing,
                                None if self.one_hot_label_format else self._ignore_index)


class _RegressionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, smooth):
        p = L.dev_f32(pred, "pred").contiguous()
        t = L.dev_f32(target, "target").contiguous()
        m = L.dev_f32(mask, "mask").contiguous()
        if p.dim() != 4 or p.shape[1] != 2 or t.shape != p.shape or tuple(m.shape) != (p.shape[0], p.shape[2], p.shape[3]):
            raise RuntimeError(f"RegressionLossWithMask: expected (B,2,W,H) pred/target and a (B,W,H) mask, got "
                               f"{tuple(pred.shape)} / {tuple(target.shape)} / {tuple(mask.shape)}")
        B, S = int(p.shape[0]), int(p.shape[2] * p.shape[3])
        lib = L.lib()
        ws_bytes = lib.dh_regression_loss_workspace(B * S)
        ws = L.workspace(ws_bytes, p.device)
        out2 = torch.empty(2, dtype=torch.float32, device=p.device)
        with L.device_guard(p.device):
            L.check(lib.dh_regression_loss_fwd(L.ptr(p), L.ptr(t), L.ptr(m), B, S, int(smooth), L.ptr(out2), L.ptr(ws), ws_bytes,
                                               L.stream()), "dh_regression_loss_fwd")
        ctx.save_for_backward(p, t, m, out2)
        ctx.cfg = (B, S, int(smooth), pred.dtype)
        return out2[0]

    @staticmethod
    def backward(ctx, grad_out):
        p, t, m, out2 = ctx.saved_tensors
        B, S, smooth, dtype = ctx.cfg
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("RegressionLossWithMask: gradients are provided for the predictions only")
        g = grad_out.reshape(1).float().contiguous()
        grad = torch.empty_like(p)
        with L.device_guard(p.device):
            L.check(L.lib().dh_regression_loss_bwd(L.ptr(p), L.ptr(t), L.ptr(m), B, S, smooth, L.ptr(out2), L.ptr(g), L.ptr(grad),
                                                   L.stream()), "dh_regression_loss_bwd")
        return (grad if dtype == torch.float32 else grad.to(dtype)), None, None, None


class RegressionLossWithMask(nn.Module):
    """loss.py:86-100: masked L1 (smooth=False) or smooth-L1 loss of the (B,2,W,H) size / offset maps."""

    def __init__(self, smooth=False):
        super().__init__()
        self.smooth = smooth

    def forward(self, pred, target, mask):
        return _RegressionLoss.apply(pred, target, mask, self.smooth)
