"""Drop-ins for object_detection/yolov5/utils/metrics.py: bbox_iou, box_iou, wh_iou, bbox_ioa (:201-296),
ConfusionMatrix (:116-171), ap_per_class / fitness (:15-86)."""
import warnings

import numpy as np
import torch

from . import _lib as L

_MODES = {"iou": 0, "giou": 1, "diou": 2, "ciou": 3}


def _mode(GIoU, DIoU, CIoU):
    # precedence of metrics.py:225-239: CIoU, then DIoU, then GIoU
    return 3 if CIoU else 2 if DIoU else 1 if GIoU else 0


def _layout(t, comp_dim, elem_dim):
    return t.stride(comp_dim), t.stride(elem_dim)


class _PairIoU(torch.autograd.Function):
    """IoU-family value of n box pairs; box tensors are 2-D with the 4 components on `comp_dim`."""

    @staticmethod
    def forward(ctx, box1, box2, comp1, comp2, xyxy, mode, eps):
        b1, b2 = L.dev_f32(box1, "box1"), L.dev_f32(box2, "box2")
        n = b1.shape[1 - comp1]
        if b1.shape[comp1] != 4 or b2.shape[comp2] != 4 or b2.shape[1 - comp2] != n:
            raise RuntimeError(f"bbox_iou: incompatible shapes {tuple(box1.shape)} and {tuple(box2.shape)}")
        if n == 1:   # a size-1 dimension may carry any stride; normalise it
            b1, b2 = b1.contiguous(), b2.contiguous()
        out = torch.empty(n, dtype=torch.float32, device=b1.device)
        s1c, s1e = _layout(b1, comp1, 1 - comp1)
        s2c, s2e = _layout(b2, comp2, 1 - comp2)
        with L.device_guard(b1.device):
            L.check(L.lib().dh_bbox_iou_fwd(L.ptr(b1), s1c, s1e, L.ptr(b2), s2c, s2e, L.ptr(out), n, int(xyxy), mode,
                                            float(eps), L.stream()), "dh_bbox_iou_fwd")
        ctx.save_for_backward(b1, b2)
        ctx.cfg = (comp1, comp2, int(xyxy), mode, float(eps), box1.dtype)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        b1, b2 = ctx.saved_tensors
        comp1, comp2, xyxy, mode, eps, dtype = ctx.cfg
        if ctx.needs_input_grad[1]:
            raise RuntimeError("bbox_iou: gradient w.r.t. box2 is not provided (the reference passes constant targets)")
        n = b1.shape[1 - comp1]
        g = grad_out.contiguous().float()
        grad1 = torch.empty_like(b1)   # same strides as box1
        s1c, s1e = _layout(b1, comp1, 1 - comp1)
        s2c, s2e = _layout(b2, comp2, 1 - comp2)
        sgc, sge = _layout(grad1, comp1, 1 - comp1)
        with L.device_guard(b1.device):
            L.check(L.lib().dh_bbox_iou_bwd(L.ptr(b1), s1c, s1e, L.ptr(b2), s2c, s2e, L.ptr(g), L.ptr(grad1), sgc, sge,
                                            n, xyxy, mode, eps, L.stream()), "dh_bbox_iou_bwd")
        return grad1.to(dtype), None, None, None, None, None, None


def bbox_iou(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """IoU / GIoU / DIoU / CIoU of box1 (4,n) against box2 (n,4) -> (n,)  (metrics.py:201-240)."""
    if box1.dim() == 1:
        box1 = box1[:, None].expand(4, box2.shape[0])
    return _PairIoU.apply(box1, box2, 0, 1, x1y1x2y2, _mode(GIoU, DIoU, CIoU), eps)


def _pairwise(name, a, b, width, out=None):
    a = L.dev_f32(a, "box1").contiguous()
    b = L.dev_f32(b, "box2").contiguous()
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != width or b.shape[1] != width:
        raise RuntimeError(f"{name}: expected (N,{width}) and (M,{width}), got {tuple(a.shape)} and {tuple(b.shape)}")
    if out is None:
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    elif out.shape != (a.shape[0], b.shape[0]) or out.dtype != torch.float32 or out.device != a.device or not out.is_contiguous():
        raise RuntimeError(f"{name}: out must be a contiguous fp32 ({a.shape[0]},{b.shape[0]}) matrix on {a.device}")
    with L.device_guard(a.device):
        L.check(getattr(L.lib(), "dh_" + name)(L.ptr(a), L.ptr(b), L.ptr(out), a.shape[0], b.shape[0], L.stream()),
                "dh_" + name)
    return out


def box_iou(box1, box2, out=None):
    """(N,4) x (M,4) xyxy boxes -> (N,M) IoU matrix (metrics.py:242-264).  out: a matrix to fill instead of a new one."""
    return _pairwise("box_iou", box1, box2, 4, out)


def wh_iou(wh1, wh2, out=None):
    """(N,2) x (M,2) width/height pairs -> (N,M) IoU matrix (metrics.py:291-296)."""
    return _pairwise("wh_iou", wh1, wh2, 2, out)


def bbox_ioa(box1, box2, eps=1E-7):
    """Intersection over box2 area, numpy in / numpy out (metrics.py:267-288).

    Host-side by interface: its callers are the CPU augmentation code that works on numpy labels.
    """
    box2 = box2.transpose()
    iw = (np.minimum(box1[2], box2[2]) - np.maximum(box1[0], box2[0])).clip(0)
    ih = (np.minimum(box1[3], box2[3]) - np.maximum(box1[1], box2[1])).clip(0)
    return iw * ih / ((box2[2] - box2[0]) * (box2[3] - box2[1]) + eps)


class ConfusionMatrix:
    """Drop-in for metrics.py:116-171 (matching + counting; plotting is not part of this package).

    The counts live on the device as uint64 and are updated by one kernel launch per call with no device-to-host
    copy; `.matrix` is the reference's float64 (nc+1, nc+1) numpy array, materialised when read.
    """

    def __init__(self, nc, conf=0.25, iou_thres=0.45):
        self.nc = nc
        self.conf = conf
        self.iou_thres = iou_thres
        self._counts = None

    def _update(self, det, lab, det_first, label_first, n_img):
        if self._counts is None:
            self._counts = torch.zeros((self.nc + 1, self.nc + 1), dtype=torch.int64, device=det.device)
        n_det, n_lab = int(det.shape[0]), int(lab.shape[0])
        if n_det == 0 and n_lab == 0:
            return
        lib = L.lib()
        ws_bytes = lib.dh_eval_match_workspace(n_det, n_lab)
        ws = L.workspace(ws_bytes, det.device)
        with L.device_guard(det.device):
            L.check(lib.dh_confusion_update(L.ptr(det), L.ptr(lab), L.ptr(det_first), L.ptr(label_first), n_img, n_det, n_lab,
                                            int(self.nc), float(self.conf), float(self.iou_thres), L.ptr(self._counts),
                                            L.ptr(ws), ws_bytes, L.stream()), "dh_confusion_update")

    def process_batch(self, detections, labels):
        """One image: detections (N,6) [x1,y1,x2,y2,conf,cls], labels (M,5) [cls,x1,y1,x2,y2] (metrics.py:125-163)."""
        from .val import _rows
        self._update(_rows(detections, 6, "detections"), _rows(labels, 5, "labels"), None, None, 1)

    def process_batches(self, detections, labels):
        """All images of a batch in one launch (lists of per-image tensors)."""
        from .val import _offsets, _rows
        if len(detections) != len(labels):
            raise RuntimeError("process_batches needs one label tensor per detection tensor")
        if not detections:
            return
        dets = [_rows(d, 6, f"detections[{i}]") for i, d in enumerate(detections)]
        labs = [_rows(t, 5, f"labels[{i}]") for i, t in enumerate(labels)]
        det_first, _ = _offsets([d.shape[0] for d in dets], dets[0].device)
        label_first, _ = _offsets([t.shape[0] for t in labs], dets[0].device)
        self._update(torch.cat(dets), torch.cat(labs), det_first, label_first, len(dets))

    @property
    def matrix(self):
        if self._counts is None:
            return np.zeros((self.nc + 1, self.nc + 1))
        return self._counts.cpu().numpy().astype(np.float64)

    def tp_fp(self):
        m = self.matrix
        tp = m.diagonal()
        fp = m.sum(1) - tp
        return tp[:-1], fp[:-1]

    def print(self):
        """metrics.py:196-198: one line per row of the matrix."""
        m = self.matrix
        for i in range(self.nc + 1):
            print(' '.join(map(str, m[i])))


AP_CLASS_CAP = 1 << 16      # class ids are integers in [0, AP_CLASS_CAP)


def fitness(x):
    """Weighted combination of [P, R, mAP@0.5, mAP@0.5:0.95] (metrics.py:15-18); host arithmetic on a (k, >=4) array."""
    w = [0.0, 0.0, 0.1, 0.9]
    return (x[:, :4] * w).sum(1)


def compute_ap(recall, precision):
    """AP of one precision / recall curve -> (ap, mpre, mrec) (metrics.py:89-114, method 'interp').  Host arithmetic on
    the small numpy curves it is given, as in the reference; ap_per_class computes the same thing on the device for every
    class and IoU threshold at once and does not go through here."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))           # precision envelope
    x = np.linspace(0, 1, 101)                                      # 101-point interpolation (COCO)
    y = np.interp(x, mrec, mpre)
    ap = (np.diff(x) * (y[1:] + y[:-1]) / 2.0).sum()                # the trapezoid rule written out
    return ap, mpre, mrec


def _to_device(a, dtype, device, name):
    if isinstance(a, torch.Tensor):
        if not a.is_cuda:
            a = a.to(device)
        return a.to(dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a)).to(device=device, dtype=dtype).contiguous()


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, save_dir='.', names=(), eps=1e-16):
    """Average precision per class and IoU threshold, and tp / fp / precision / recall / F1 per class at the confidence
    of the best mean F1 (metrics.py:21-86) -> (tp, fp, p, r, f1, ap, unique_classes) as numpy arrays (float64, classes int32).

    tp (n, n_iou) bool, conf (n,), pred_cls (n,), target_cls (m,): numpy arrays as val.py collects them, or device tensors
    (no per-image host copies are needed then).  One sort and one workgroup per (class, IoU threshold) replace the
    reference's per-class Python loop; two small host reads (the class list, the results).  The curve plots of the
    reference (plot=True) are not drawn.  Detections of one class with equal confidence keep their input order, where
    the reference's unstable argsort may order them either way.
    """
    if plot:
        warnings.warn("ap_per_class: the precision/recall curve plots are not drawn")
    dev = next((a.device for a in (tp, conf, pred_cls, target_cls) if isinstance(a, torch.Tensor) and a.is_cuda),
               torch.device("cuda", torch.cuda.current_device()))
    if not isinstance(conf, torch.Tensor):
        c64 = np.asarray(conf)
        if c64.dtype != np.float32 and not np.array_equal(c64.astype(np.float32), c64):
            raise ValueError("ap_per_class: confidences must be float32 values (as val.py collects them)")
    tp_d = _to_device(tp, torch.uint8, dev, "tp")
    if tp_d.dim() == 1:
        tp_d = tp_d[:, None]
    conf_d = _to_device(conf, torch.float32, dev, "conf")
    cls_d = _to_device(pred_cls, torch.float32, dev, "pred_cls")
    tcls_d = _to_device(target_cls, torch.float64, dev, "target_cls").reshape(-1)
    n, niou = tp_d.shape
    if conf_d.shape != (n,) or cls_d.shape != (n,):
        raise RuntimeError(f"ap_per_class: tp {tuple(tp_d.shape)}, conf {tuple(conf_d.shape)}, pred_cls {tuple(cls_d.shape)} disagree")
    lib = L.lib()
    m = tcls_d.numel()
    ints = torch.empty((3, AP_CLASS_CAP), dtype=torch.int32, device=dev)          # unique, nt, histogram scratch
    meta = torch.empty(2, dtype=torch.int32, device=dev)
    with L.device_guard(dev):
        L.check(lib.dh_ap_classes(L.ptr(tcls_d), m, AP_CLASS_CAP, L.ptr(ints[0]), L.ptr(ints[1]), L.ptr(meta), L.ptr(ints[2]),
                                  L.stream()), "dh_ap_classes")
        nu, bad = meta.tolist()
        if bad:
            raise ValueError(f"ap_per_class: target classes must be integers in [0, {AP_CLASS_CAP})")
        ap = torch.empty((nu, niou), dtype=torch.float64, device=dev)
        out5 = torch.empty((5, nu), dtype=torch.float64, device=dev)
        unique = ints[0, :nu]
        if nu:
            curves = torch.empty((2, nu, 1000), dtype=torch.float64, device=dev)
            best = torch.empty(1, dtype=torch.int32, device=dev)
            ws_bytes = lib.dh_ap_per_class_workspace(n, niou)
            ws = L.workspace(ws_bytes, dev)
            L.check(lib.dh_ap_per_class(L.ptr(tp_d), L.ptr(conf_d), L.ptr(cls_d), n, niou, L.ptr(unique), L.ptr(ints[1]), nu,
                                        AP_CLASS_CAP, float(eps), L.ptr(ap), L.ptr(curves[0]), L.ptr(curves[1]), L.ptr(out5),
                                        L.ptr(best), L.ptr(ws), ws_bytes, L.stream()), "dh_ap_per_class")
    o = out5.cpu().numpy()
    return o[0], o[1], o[2], o[3], o[4], ap.cpu().numpy(), unique.cpu().numpy().astype("int32")
