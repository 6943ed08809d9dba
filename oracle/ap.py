"""Oracle: per-class average precision and the precision / recall / F1 curves (CPU, test-only).

Restates
  object_detection/yolov5/utils/metrics.py:21-86    (ap_per_class, without the plotting branches)
  object_detection/yolov5/utils/metrics.py:89-114   (compute_ap, method 'interp')
  object_detection/yolov5/utils/metrics.py:15-18    (fitness)
in the reference's operation order, float64 throughout.  The reference sorts with `np.argsort(-conf)` (default kind, an
unstable introsort); detections of one class with equal confidence but different tp rows can therefore come out in
either order.  Here the sort is the stable variant (equal confidences keep their input order), which is one of the
orders the reference can produce; inputs with distinct confidences are insensitive to the choice.
`np.trapz` is written out (it is the same three array operations and one `sum`), as newer numpy versions drop the name.
"""
import numpy as np


def fitness(x):
    """metrics.py:15-18."""
    w = [0.0, 0.0, 0.1, 0.9]
    return (x[:, :4] * w).sum(1)


def compute_ap(recall, precision):
    """metrics.py:89-114 -> (ap, mpre, mrec)."""
    mrec = np.concatenate(([0.0], recall, [1.0]))                     # :99
    mpre = np.concatenate(([1.0], precision, [0.0]))                  # :100
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))              # :103
    x = np.linspace(0, 1, 101)                                        # :108
    y = np.interp(x, mrec, mpre)
    d = np.diff(x)
    ap = (d * (y[1:] + y[:-1]) / 2.0).sum()                           # :109  np.trapz(y, x)
    return ap, mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """metrics.py:21-86 -> (tp, fp, p, r, f1, ap, unique_classes int32)."""
    i = np.argsort(-conf, kind="stable")                              # :36
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes, nt = np.unique(target_cls, return_counts=True)    # :40
    nc = unique_classes.shape[0]
    px = np.linspace(0, 1, 1000)                                      # :44
    ap, p, r = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        i = pred_cls == c
        n_l = nt[ci]
        n_p = i.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[i]).cumsum(0)                                   # :55
        tpc = tp[i].cumsum(0)
        recall = tpc / (n_l + eps)                                    # :59
        r[ci] = np.interp(-px, -conf[i], recall[:, 0], left=0)        # :60
        precision = tpc / (tpc + fpc)                                 # :63
        p[ci] = np.interp(-px, -conf[i], precision[:, 0], left=1)     # :64
        for j in range(tp.shape[1]):
            ap[ci, j], _, _ = compute_ap(recall[:, j], precision[:, j])   # :68
    f1 = 2 * p * r / (p + r + eps)                                    # :73
    i = f1.mean(0).argmax()                                           # :82
    p, r, f1 = p[:, i], r[:, i], f1[:, i]
    tp = (r * nt).round()                                             # :84
    fp = (tp / (p + eps) - tp).round()                                # :85
    return tp, fp, p, r, f1, ap, unique_classes.astype("int32")
