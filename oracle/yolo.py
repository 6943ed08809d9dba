"""Oracle: YOLOv5 head decode, target assignment and loss (CPU, test-only).

Restates
  object_detection/yolov5/models/yolo.py:54,56-68,72-81   (Detect inference decode, grids)
  object_detection/yolov5/utils/loss.py:169-222            (ComputeLoss.build_targets)
  object_detection/yolov5/utils/loss.py:117-167            (ComputeLoss.__call__)
  object_detection/yolov5/utils/loss.py:13-62              (smooth_BCE, FocalLoss)
"""
import numpy as np
import torch
import torch.nn.functional as F

from .boxes import bbox_iou_yolo

YOLOV5S_ANCHORS_PX = (  # object_detection/yolov5/models/yolov5s.yaml:7-10
    (10, 13, 16, 30, 33, 23),
    (30, 61, 62, 45, 59, 119),
    (116, 90, 156, 198, 373, 326),
)
YOLOV5S_STRIDES = (8.0, 16.0, 32.0)

HYP_SCRATCH = dict(  # object_detection/yolov5/data/hyps/hyp.scratch.yaml:15-24
    box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


def grid_anchors(anchors_px=YOLOV5S_ANCHORS_PX, strides=YOLOV5S_STRIDES):
    """Anchors in grid units, (nl,na,2) fp32 (yolo.py:45,113: pixels / stride)."""
    a = torch.tensor(anchors_px).float().view(len(anchors_px), -1, 2)
    return a / torch.tensor(strides).view(-1, 1, 1)


def head_maps(convs, na):
    """The layout change between the 1x1 convolutions and the head maps (yolo.py:53-54):
    (bs, na*no, ny, nx) -> view (bs, na, no, ny, nx) -> permute(0, 1, 3, 4, 2) -> contiguous (bs, na, ny, nx, no)."""
    out = []
    for c in convs:
        bs, ch, ny, nx = c.shape
        out.append(c.view(bs, na, ch // na, ny, nx).permute(0, 1, 3, 4, 2).contiguous())
    return out


def detect_decode(levels, anchors, strides, inplace=True):
    """Inference branch of Detect.forward (yolo.py:56-70), input already (bs,na,ny,nx,no).

    xy=(2s-0.5+grid)*stride, wh=(2s)^2*(anchor*stride), rest = sigmoid; levels are
    flattened (anchor, gy, gx) and concatenated in level order.  The same torch operations in the same order and
    with the same operand types as the reference (float grid and anchor grid, yolo.py:78-80), so fp16 levels and
    anchors give what a half-precision model gives; inplace=False is the torch.cat form (yolo.py:63-65), which
    only differs from the slice assignment in that case.
    """
    flat = []
    for lvl, x in enumerate(levels):
        bs, na, ny, nx, no = x.shape
        gy, gx = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
        cell = torch.stack((gx, gy), 2).expand(1, na, ny, nx, 2).float()        # yolo.py:78
        stride = torch.as_tensor(strides[lvl], dtype=torch.float32)
        awh = (anchors[lvl].clone() * stride).view(1, na, 1, 1, 2).expand(1, na, ny, nx, 2).float()
        y = x.sigmoid()
        if inplace:
            y[..., 0:2] = (y[..., 0:2] * 2 - 0.5 + cell) * stride               # yolo.py:59
            y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * awh                          # yolo.py:60
        else:
            xy = (y[..., 0:2] * 2 - 0.5 + cell) * stride                        # yolo.py:63
            wh = (y[..., 2:4] * 2) ** 2 * awh                                   # yolo.py:64
            y = torch.cat((xy, wh, y[..., 4:]), -1)                             # yolo.py:65
        flat.append(y.view(bs, -1, no))
    return torch.cat(flat, 1)


_HALF_STEPS = np.array([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=np.float32) * np.float32(0.5)


def build_targets(level_shapes, targets, anchors, anchor_t=4.0):
    """loss.py:169-222 in numpy fp32.

    level_shapes: [(ny,nx)] per level; targets (nt,6) [img,cls,x,y,w,h] normalised;
    anchors (nl,na,2) grid units.  Returns per level
      (b, a, gj, gi) int64, tbox (n,4) fp32, anch (n,2) fp32, tcls int64
    in the reference's row order: offset-major over {centre,+x,+y,-x,-y} neighbours,
    then anchor-major, then target order (two row-major boolean compactions,
    loss.py:194,202).  The clamp of loss.py:217 acts in place on views of gij, so
    tbox (loss.py:218) is formed with the *clamped* cell indices.
    """
    targets = np.asarray(targets, dtype=np.float32).reshape(-1, 6)
    anchors = np.asarray(anchors, dtype=np.float32)
    nl, na = anchors.shape[0], anchors.shape[1]
    nt = targets.shape[0]
    out = []
    for lvl in range(nl):
        ny, nx = level_shapes[lvl]
        scale = np.array([1, 1, nx, ny, nx, ny], dtype=np.float32)
        rows = np.tile(targets * scale, (na, 1))                  # anchor-major, target-minor
        aidx = np.repeat(np.arange(na, dtype=np.int64), nt)
        steps = np.zeros((0, 2), dtype=np.float32)
        if nt:
            with np.errstate(divide="ignore", invalid="ignore"):
                ratio = rows[:, 4:6] / anchors[lvl][aidx]
                worst = np.maximum(ratio, np.float32(1) / ratio).max(1)
            ok = worst < np.float32(anchor_t)                      # loss.py:191-194
            rows, aidx = rows[ok], aidx[ok]
            gxy = rows[:, 2:4]
            inv = scale[[2, 3]] - gxy                              # loss.py:198
            lo = (np.remainder(gxy, np.float32(1)) < np.float32(0.5)) & (gxy > np.float32(1))
            hi = (np.remainder(inv, np.float32(1)) < np.float32(0.5)) & (inv > np.float32(1))
            take = np.stack((np.ones(len(rows), dtype=bool), lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1]))
            step_id, src = np.nonzero(take)                        # row-major == offset-major
            rows, aidx, steps = rows[src], aidx[src], _HALF_STEPS[step_id]
        b = rows[:, 0].astype(np.int64)
        c = rows[:, 1].astype(np.int64)
        gxy, gwh = rows[:, 2:4], rows[:, 4:6]
        cell = (gxy - steps).astype(np.int64) if nt else gxy.astype(np.int64)   # trunc toward zero
        gi = np.clip(cell[:, 0], 0, nx - 1)
        gj = np.clip(cell[:, 1], 0, ny - 1)
        cell_f = np.stack((gi, gj), 1).astype(np.float32)
        tbox = np.concatenate((gxy - cell_f, gwh), 1).astype(np.float32)
        out.append(dict(b=b, a=aidx, gj=gj, gi=gi, tbox=tbox, anch=anchors[lvl][aidx], tcls=c))
    return out


def _focal(bce_none, logits, target, gamma, alpha=0.25):
    """FocalLoss.forward (loss.py:45-62) around an un-reduced BCE; mean reduction."""
    prob = torch.sigmoid(logits)
    p_t = target * prob + (1 - target) * (1 - prob)
    a_f = target * alpha + (1 - target) * (1 - alpha)
    return (bce_none * (a_f * (1.0 - p_t) ** gamma)).mean()


class YoloLossOracle:
    """ComputeLoss (loss.py:91-167) as a plain callable over explicit hyper-parameters."""

    def __init__(self, anchors, hyp=None, nc=80, autobalance=False, strides=YOLOV5S_STRIDES):
        self.hyp = dict(HYP_SCRATCH)
        if hyp:
            self.hyp.update(hyp)
        self.anchors = anchors.float()
        self.nl, self.na, self.nc = anchors.shape[0], anchors.shape[1], nc
        self.balance = {3: [4.0, 1.0, 0.4]}.get(self.nl, [4.0, 1.0, 0.25, 0.06, 0.02])   # loss.py:111
        self.ssi = list(strides).index(16) if autobalance else 0
        self.autobalance = autobalance
        eps = self.hyp.get("label_smoothing", 0.0)
        self.cp, self.cn = 1.0 - 0.5 * eps, 0.5 * eps                                 # loss.py:13-15
        self.gr = 1.0
        self.sort_obj_iou = False

    def _bce(self, logits, target, pos_weight, sums=None):
        pw = torch.tensor([pos_weight], dtype=logits.dtype)
        g = self.hyp["fl_gamma"]
        if sums is not None:      # the sum of the same terms in double precision, for the data-parallel checks
            with torch.no_grad():
                e = F.binary_cross_entropy_with_logits(logits.double(), target.double(), pos_weight=pw.double(), reduction="none")
                if g > 0:
                    prob = torch.sigmoid(logits.double())
                    p_t = target * prob + (1 - target) * (1 - prob)
                    e = e * ((target * 0.25 + (1 - target) * 0.75) * (1.0 - p_t) ** g)
                sums.append(float(e.sum()))
        if g > 0:
            return _focal(F.binary_cross_entropy_with_logits(logits, target, pos_weight=pw, reduction="none"),
                          logits, target, g)
        return F.binary_cross_entropy_with_logits(logits, target, pos_weight=pw)

    def targets_for(self, p, targets):
        shapes = [(int(x.shape[2]), int(x.shape[3])) for x in p]
        return build_targets(shapes, targets.detach().cpu().numpy(), self.anchors.numpy(), self.hyp["anchor_t"])

    def __call__(self, p, targets):
        """p in float32 is the reference's arithmetic; p in float64 evaluates the same formulas (same fp32 assignment)
        in double precision, the yardstick for gradient entries that are differences of nearly equal terms."""
        dt = p[0].dtype
        lcls, lbox, lobj = torch.zeros(1, dtype=dt), torch.zeros(1, dtype=dt), torch.zeros(1, dtype=dt)
        assigned = self.targets_for(p, targets)
        self.last_sums = torch.zeros(len(p), 5, dtype=torch.float64)   # per level: sum(1-iou), sum cls BCE, sum obj BCE, rows, cells
        for lvl, pi in enumerate(p):
            t = assigned[lvl]
            b, a, gj, gi = (torch.from_numpy(t[k]) for k in ("b", "a", "gj", "gi"))
            tobj = torch.zeros_like(pi[..., 0])
            n = b.shape[0]
            if n:
                ps = pi[b, a, gj, gi]                                                  # loss.py:129
                pxy = ps[:, :2].sigmoid() * 2 - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * torch.from_numpy(t["anch"]).to(dt)
                pbox = torch.cat((pxy, pwh), 1)
                iou = bbox_iou_yolo(pbox.T, torch.from_numpy(t["tbox"]).to(dt), x1y1x2y2=False, CIoU=True)
                lbox = lbox + (1.0 - iou).mean()                                       # loss.py:136
                self.last_sums[lvl, 0] = float((1.0 - iou.detach().double()).sum())
                score_iou = iou.detach().clamp(0).type(tobj.dtype)
                if self.sort_obj_iou:                                                  # loss.py:140-142
                    order = torch.argsort(score_iou)
                    b, a, gj, gi, score_iou = b[order], a[order], gj[order], gi[order], score_iou[order]
                tobj[b, a, gj, gi] = (1.0 - self.gr) + self.gr * score_iou             # loss.py:143
                if self.nc > 1:                                                        # loss.py:146-149
                    tc = torch.full_like(ps[:, 5:], self.cn)
                    tc[range(n), torch.from_numpy(t["tcls"])] = self.cp
                    acc = []
                    lcls = lcls + self._bce(ps[:, 5:], tc, self.hyp["cls_pw"], acc)
                    self.last_sums[lvl, 1] = acc[0]
            acc = []
            obji = self._bce(pi[..., 4], tobj, self.hyp["obj_pw"], acc)                # loss.py:155
            self.last_sums[lvl, 2], self.last_sums[lvl, 3], self.last_sums[lvl, 4] = acc[0], n, tobj.numel()
            lobj = lobj + obji * self.balance[lvl]
            if self.autobalance:
                self.balance[lvl] = self.balance[lvl] * 0.9999 + 0.0001 / obji.detach().item()
        if self.autobalance:
            self.balance = [x / self.balance[self.ssi] for x in self.balance]
        lbox = lbox * self.hyp["box"]
        lobj = lobj * self.hyp["obj"]
        lcls = lcls * self.hyp["cls"]
        bs = p[0].shape[0]
        return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()


def descale_pred(p, flips, scale, img_size):
    """Model._descale_pred, in-place branch (object_detection/yolov5/models/yolo.py:155-163)."""
    p[..., :4] /= scale
    if flips == 2:
        p[..., 1] = img_size[0] - p[..., 1]
    elif flips == 3:
        p[..., 0] = img_size[1] - p[..., 0]
    return p


def clip_augmented(y, nl):
    """Model._clip_augmented (object_detection/yolov5/models/yolo.py:174-183)."""
    g = sum(4 ** x for x in range(nl))
    e = 1
    i = (y[0].shape[1] // g) * sum(4 ** x for x in range(e))
    y[0] = y[0][:, :-i]
    i = (y[-1].shape[1] // g) * sum(4 ** (nl - 1 - x) for x in range(e))
    y[-1] = y[-1][:, i:]
    return y
