#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference's own Python code on CPU.

Run in the build container only (needs /root/reference, which never travels to
the GPU box):   python tests/golden/make_golden.py

How the reference is imported (no reference source is modified or copied):
  * sys.path gets /root/reference/object_detection/yolov5 and /root/reference/object_detection;
  * third-party packages that are absent here (cv2, seaborn, thop, torchvision) are
    replaced by stub modules; the stub ``torchvision.ops.nms`` is the oracle's C
    greedy NMS (oracle/csrc/oracle_nms.c), and every call to it is recorded so the
    fixture pins everything the reference does *around* that third-party kernel;
  * the reference targets torch 1.10, where ``LongTensor.clamp_(0, float0dTensor)``
    casts the bound to the tensor's dtype; torch 2.10 raises instead, so
    ``Tensor.clamp_`` is wrapped to restore that cast (utils/loss.py:217).
  * modules that reach the network at import time (utils/plots.py, models/*) are
    never imported.
All random inputs come from numpy's legacy ``RandomState`` (bit-stable stream).
"""
import os
import sys
import types
from unittest.mock import MagicMock

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference/object_detection"

from oracle import nms as oracle_nms  # noqa: E402  (stand-in for the absent torchvision kernel)

NMS_CALLS = []


def _recording_nms(boxes, scores, thr):
    keep = oracle_nms.nms(boxes, scores, thr)
    NMS_CALLS.append((boxes.detach().clone().numpy(), scores.detach().clone().numpy(), float(thr), keep.numpy()))
    return keep


def install_stubs():
    for name in ("cv2", "seaborn", "thop"):
        sys.modules[name] = MagicMock()
    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.11.1"
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.nms = _recording_nms
    tv.ops.sigmoid_focal_loss = None
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tv.ops

    orig = torch.Tensor.clamp_

    def clamp_(self, min=None, max=None):
        def cast(v):
            if isinstance(v, torch.Tensor) and v.dim() == 0:
                v = v.item()
            if v is not None and not self.is_floating_point():
                v = int(v)
            return v
        return orig(self, cast(min), cast(max))
    torch.Tensor.clamp_ = clamp_


def load_reference():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "yolov5"))
    sys.path.insert(0, REF)
    import utils.metrics as y_metrics
    import utils.general as y_general
    import utils.loss as y_loss
    import loss as a_loss
    import detectors.centernet.loss as c_loss
    import detectors.centernet.dataset as c_data
    return y_metrics, y_general, y_loss, a_loss, c_loss, c_data


from tests._synth import (f32, rand_boxes_xywh, synth_targets, synth_head, synth_pred, dedupe_scores,  # noqa: E402
                          make_big_pred, HALF_DETECT_TAGS, HALF_NMS_VARIANTS)


def T(a):
    return torch.from_numpy(f32(a))


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# ----------------------------------------------------------------------------- inputs
def special_pairs():
    """xywh pairs: identical, disjoint, nested, touching edges, zero w/h, tiny boxes, equal corners."""
    a = [[.5, .5, .2, .3], [.2, .2, .1, .1], [.5, .5, .4, .4], [.3, .5, .2, .2], [.5, .5, 0., .2],
         [.5, .5, .2, 0.], [.5, .5, 0., 0.], [.5, .5, 1e-4, 1e-4], [.5, .5, 1e-7, 2e-7], [.25, .25, .5, .5],
         [0., 0., 0., 0.], [.5, .5, .2, .2]]
    b = [[.5, .5, .2, .3], [.8, .8, .1, .1], [.5, .5, .1, .1], [.5, .5, .2, .2], [.5, .5, .2, .2],
         [.5, .5, .2, .2], [.5, .5, 0., 0.], [.50005, .5, 1e-4, 1e-4], [.5, .5, 2e-7, 1e-7], [.5, .25, .5, .5],
         [0., 0., 0., 0.], [.6, .5, .2, .4]]
    return f32(a), f32(b)


def xywh_to_xyxy_np(b):
    return f32(np.stack([b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2], 1))


MODE_FLAGS = {"iou": {}, "giou": {"GIoU": True}, "diou": {"DIoU": True}, "ciou": {"CIoU": True}}


# ----------------------------------------------------------------------------- generators
def gen_bbox_iou(y_metrics, a_loss):
    rs = np.random.RandomState(11)
    out = {}
    sa, sb = special_pairs()
    for tag, n in (("n1", 1), ("n257", 257), ("n2048", 2048), ("special", len(sa))):
        if tag == "special":
            p, t = sa, sb
        else:
            p, t = rand_boxes_xywh(rs, n), rand_boxes_xywh(rs, n)
            t[: n // 4] = p[: n // 4] + f32(rs.normal(0, 0.02, (n // 4, 4)))     # near-matches
            t[:, 2:] = np.abs(t[:, 2:])
        for fmt in ("xywh", "xyxy"):
            pp, tt = (p, t) if fmt == "xywh" else (xywh_to_xyxy_np(p), xywh_to_xyxy_np(t))
            out[f"{tag}_{fmt}_box1"] = pp
            out[f"{tag}_{fmt}_box2"] = tt
            for mode, flags in MODE_FLAGS.items():
                b1 = T(pp).T.contiguous().requires_grad_(True)                # (4,n) as loss.py:135 passes it
                val = y_metrics.bbox_iou(b1, T(tt), x1y1x2y2=(fmt == "xyxy"), **flags)
                g, = torch.autograd.grad(val.sum(), b1)
                out[f"{tag}_{fmt}_{mode}_out"] = val.detach().numpy()
                out[f"{tag}_{fmt}_{mode}_grad"] = g.numpy()                    # (4,n)
    # author's [...,4] layout on a 3-d batch, default xywh
    p, t = rand_boxes_xywh(rs, 60).reshape(3, 20, 4), rand_boxes_xywh(rs, 60).reshape(3, 20, 4)
    out["lastdim_box1"], out["lastdim_box2"] = p, t
    for mode, flags in MODE_FLAGS.items():
        b1 = T(p).requires_grad_(True)
        val = a_loss.bbox_iou(b1, T(t), **flags)
        g, = torch.autograd.grad(val.sum(), b1)
        out[f"lastdim_{mode}_out"], out[f"lastdim_{mode}_grad"] = val.detach().numpy(), g.numpy()
    b1 = T(xywh_to_xyxy_np(p.reshape(-1, 4)).reshape(3, 20, 4))
    out["lastdim_xyxy_ciou_out"] = a_loss.bbox_iou(b1, T(xywh_to_xyxy_np(t.reshape(-1, 4)).reshape(3, 20, 4)),
                                                   x1y1x2y2=True, CIoU=True).numpy()
    # one hand-checkable pair (CIoU of two nearby boxes given as centre and size)
    v = y_metrics.bbox_iou(T([[.5, .5, .2, .3]]).T, T([[.52, .48, .25, .28]]), x1y1x2y2=False, CIoU=True)
    out["sanity_ciou"] = v.numpy()
    save("bbox_iou", **out)


def centernet_maps(rs, b, c, w, h, n_obj):
    """Synthetic head outputs + label maps; label maps come from the reference's own target code later."""
    pred_cls = f32(rs.normal(-2.0, 1.5, (b, c, w, h)))
    pred_size = f32(np.abs(rs.normal(6, 3, (b, 2, w, h))))
    pred_off = f32(rs.uniform(0, 1, (b, 2, w, h)))
    return pred_cls, pred_size, pred_off


def gen_centernet(a_loss, c_loss, c_data):
    import dataset as ds_mod
    rs = np.random.RandomState(23)
    out = {}
    B, C, W, H = 2, 3, 40, 36            # W != H so that an x/y swap cannot hide
    inp = (W * 4, H * 4)

    # ---- target drawing through the reference's own __getitem__ (dataset.py:96-161)
    label_sets = [
        f32([[10, 20, 50, 90, 0], [12, 22, 52, 88, 1], [100, 30, 140, 60, 2], [0, 0, 8, 8, 1],
             [150, 130, 170, 150, 0], [30, 30, 30, 60, 2], [-20, 40, 25, 70, 1], [90, 100, 91, 101, 0]]),
        f32([[60, 60, 100, 100, 2], [61, 61, 99, 99, 0], [5, 100, 45, 140, 1]]),
        np.zeros((0, 5), dtype=np.float32),
    ]
    feed = {}
    ds_mod.YOLODataset.__getitem__ = lambda self, idx: (None, None, len(feed["l"]), feed["l"])
    ds = object.__new__(c_data.CenternetDataset)
    ds.length, ds.input_shape, ds.output_shape, ds.num_classes = 1, inp, (W, H), C
    drawn = []
    for k, lab in enumerate(label_sets):
        feed["l"] = lab
        item = ds[0]
        drawn.append(item[4:])
        out[f"targets{k}_labels"] = lab
        for nm, arr in zip(("cls", "size", "offset", "mask"), item[4:]):
            out[f"targets{k}_{nm}"] = f32(arr)
    out["targets_input_shape"] = np.array(inp)
    out["targets_output_shape"] = np.array((W, H))

    # ---- box loss through compute_losses (centernet/loss.py:16-28) + IoULossWithMask (loss.py:102-115)
    pred_cls, pred_size, pred_off = centernet_maps(rs, B, C, W, H, 0)
    lab_cls = f32(np.stack([drawn[0][0], drawn[1][0]]))
    lab_size = f32(np.stack([drawn[0][1], drawn[1][1]]))
    lab_off = f32(np.stack([drawn[0][2], drawn[1][2]]))
    lab_mask = f32(np.stack([drawn[0][3], drawn[1][3]]))
    # make the predictions near the labels on positive cells so CIoU is in its interesting range
    pos = lab_mask[:, None] > 0
    pred_size = np.where(pos, lab_size * f32(rs.uniform(0.6, 1.5, lab_size.shape)), pred_size).astype(np.float32)
    pred_off = np.where(pos, lab_off + f32(rs.normal(0, 0.3, lab_off.shape)), pred_off).astype(np.float32)
    crit = {"cls": lambda a, b: torch.zeros(()), "box": a_loss.IoULossWithMask(CIoU=True), "size": None, "offset": None}
    for tag, mask in (("pos", lab_mask), ("zero", np.zeros_like(lab_mask)), ("dense", f32(rs.uniform(0, 1, lab_mask.shape) < 0.3))):
        ps, po = T(pred_size).requires_grad_(True), T(pred_off).requires_grad_(True)
        losses = {}
        c_loss.compute_losses([T(pred_cls), ps, po], [T(lab_cls), T(lab_size), T(lab_off), T(mask)], crit, losses)
        gs, go = torch.autograd.grad(losses["box"], [ps, po], allow_unused=True)
        out[f"loss_{tag}_mask"] = mask
        out[f"loss_{tag}_value"] = losses["box"].detach().numpy()
        out[f"loss_{tag}_grad_size"] = gs.numpy()
        out[f"loss_{tag}_grad_offset"] = go.numpy()
    out.update(loss_pred_cls=pred_cls, loss_pred_size=pred_size, loss_pred_offset=pred_off,
               loss_lab_cls=lab_cls, loss_lab_size=lab_size, loss_lab_offset=lab_off)
    out["offset_to_coord"] = c_loss.offset_to_coord(T(pred_off)).numpy()
    # IoULossWithMask called directly on (B,4,W,H) maps; flags other than CIoU are ignored by the reference
    p4 = T(np.concatenate([pred_off * 5, pred_size], 1)).requires_grad_(True)
    t4 = T(np.concatenate([lab_off * 5, lab_size + 1], 1))
    val = a_loss.IoULossWithMask(GIoU=True)(p4, t4, T(lab_mask))
    g, = torch.autograd.grad(val, p4)
    out.update(direct_pred=p4.detach().numpy(), direct_target=t4.numpy(), direct_mask=lab_mask,
               direct_value=val.detach().numpy(), direct_grad=g.numpy())

    # ---- decode / post-process (dataset.py:164-297)
    heat = f32(rs.normal(-1.0, 1.2, (B + 1, C, W, H)))
    heat[0, 1, 5, 7] = 3.0; heat[0, 1, 5, 8] = 3.0          # equal neighbours: both survive the 3x3 peak test
    heat[0, 2, 20, 30] = 4.0; heat[0, 0, 21, 30] = 3.5      # different classes, adjacent cells
    heat[0, 0, 0, 0] = 2.5; heat[0, 2, W - 1, H - 1] = 2.8  # corners
    heat[1, 0, 10, 10] = 2.0; heat[1, 1, 10, 10] = 2.0      # class tie in one cell -> first class
    heat[2] = -3.0                                          # nothing above threshold
    size = f32(np.abs(rs.normal(6, 3, (B + 1, 2, W, H))))
    off = f32(rs.uniform(0, 1, (B + 1, 2, W, H)))
    out.update(dec_heat=heat, dec_size=size, dec_offset=off)
    out["pool_nms"] = c_data.pool_nms(T(heat)).numpy()
    for conf in (0.9, 0.3):
        dets = c_data.decode_bbox(T(heat), T(size), T(off), conf, "cpu")
        for k, d in enumerate(dets):
            out[f"decode_c{conf}_img{k}"] = d.numpy() if len(d) else np.zeros((0, 6), np.float32)
        for need in (False, True):
            NMS_CALLS.clear()
            post = c_data.postprocess([d.clone() if len(d) else d for d in dets], need, 640, 160.0, False, 0.4)
            for k, d in enumerate(post):
                out[f"post_c{conf}_nms{int(need)}_img{k}"] = np.zeros((0, 6), np.float32) if d is None else d
        post = c_data.postprocess([d.clone() if len(d) else d for d in dets], True, 500, 640.0, True, 0.4)
        for k, d in enumerate(post):
            out[f"post_c{conf}_letterbox_img{k}"] = np.zeros((0, 6), np.float32) if d is None else d
    post = c_data.postprocess_predictions([T(heat), T(size), T(off)], 640, "cpu")
    for k, d in enumerate(post):
        out[f"postpred_img{k}"] = np.zeros((0, 6), np.float32) if d is None else d
    save("centernet", **out)


def gen_pairwise(y_metrics):
    rs = np.random.RandomState(31)
    out = {}
    for tag, (n, m) in (("0x5", (0, 5)), ("7x5", (7, 5)), ("100x300", (100, 300)), ("65x129", (65, 129))):
        a, b = xywh_to_xyxy_np(rand_boxes_xywh(rs, n, 640)).reshape(n, 4), xywh_to_xyxy_np(rand_boxes_xywh(rs, m, 640))
        if n and m:
            b[0] = a[0]
        out[f"box_iou_{tag}_a"], out[f"box_iou_{tag}_b"] = a, b
        out[f"box_iou_{tag}_out"] = y_metrics.box_iou(T(a).reshape(n, 4), T(b)).numpy()
        wa, wb = f32(rs.uniform(1, 300, (n, 2))), f32(rs.uniform(1, 300, (m, 2)))
        out[f"wh_iou_{tag}_a"], out[f"wh_iou_{tag}_b"] = wa, wb
        out[f"wh_iou_{tag}_out"] = y_metrics.wh_iou(T(wa).reshape(n, 2), T(wb)).numpy()
    box1 = f32([100, 120, 300, 280])
    box2 = xywh_to_xyxy_np(rand_boxes_xywh(rs, 50, 640))
    out.update(ioa_box1=box1, ioa_box2=box2, ioa_out=y_metrics.bbox_ioa(box1, box2))
    save("pairwise", **out)


def edge_targets(nx_list=(80, 40, 20)):
    """Targets sitting exactly on decision boundaries of loss.py:191-217."""
    rows = []
    for nx in nx_list:
        for gx, gy in ((2.5, 3.5), (1.0, 1.0), (0.25, 0.75), (1.5, nx - 1.5), (nx - 1.0, nx - 0.5),
                       (nx - 0.25, 0.0), (float(nx), float(nx)), (3.0, 4.0), (3.4999998, 3.5000002)):
            rows.append([0, 1, gx / nx, gy / nx, 0.1, 0.1])
    for s, (aw, ah) in zip((8, 16, 32), ((10, 13), (30, 61), (116, 90))):       # wh ratio exactly 4, 1/4, just inside
        for k in (4.0, 0.25, 3.9999, 0.2500001, 4.0001):
            rows.append([1, 2, 0.5, 0.5, aw * k / 640, ah / 640])
    return f32(rows)


def fake_model(y_loss, nc, hyp, anchors, strides):
    class Det:
        pass
    det = Det()
    det.na, det.nc, det.nl, det.anchors, det.stride = anchors.shape[1], nc, anchors.shape[0], anchors, strides

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))
            self.model = [det]
    m = M()
    m.hyp = hyp
    return m


HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)


def yolo_anchors():
    a = torch.tensor([[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]).float()
    strides = torch.tensor([8., 16., 32.])
    return a.view(3, 3, 2) / strides.view(-1, 1, 1), strides


def gen_build_targets(y_loss):
    rs = np.random.RandomState(41)
    anchors, strides = yolo_anchors()
    cl = y_loss.ComputeLoss(fake_model(y_loss, 80, dict(HYP), anchors, strides))
    out = {"anchors": anchors.numpy()}
    shapes = [(8, 3, 80, 80, 85), (8, 3, 40, 40, 85), (8, 3, 20, 20, 85)]
    p = [torch.empty(s) for s in shapes]
    cases = {"nt0": np.zeros((0, 6), np.float32), "nt1": synth_targets(rs, 1, 8), "nt80": synth_targets(rs, 80, 8),
             "nt640": synth_targets(rs, 640, 8), "edges": edge_targets()}
    # non-square grid: catches any nx/ny swap
    p_rect = [torch.empty(4, 3, 48, 80, 85), torch.empty(4, 3, 24, 40, 85), torch.empty(4, 3, 12, 20, 85)]
    cases["rect"] = synth_targets(rs, 120, 4)
    for tag, tg in cases.items():
        pp = p_rect if tag == "rect" else p
        tcls, tbox, indices, anch = cl.build_targets(pp, T(tg).reshape(-1, 6))
        out[f"{tag}_targets"] = tg
        out[f"{tag}_shapes"] = np.array([[x.shape[2], x.shape[3]] for x in pp])
        for l in range(3):
            b, a, gj, gi = indices[l]
            out[f"{tag}_l{l}_idx"] = torch.stack((b, a, gj, gi), 1).numpy().astype(np.int64)
            out[f"{tag}_l{l}_tbox"] = tbox[l].numpy()
            out[f"{tag}_l{l}_anch"] = anch[l].numpy()
            out[f"{tag}_l{l}_tcls"] = tcls[l].numpy().astype(np.int64)
    save("build_targets", **out)


def gen_compute_loss(y_loss):
    rs = np.random.RandomState(53)
    anchors, strides = yolo_anchors()
    out = {"anchors": anchors.numpy()}

    def run(tag, img, bs, nc, targets, hyp_over=None, autobalance=False, calls=1, store_grads=True, sort_obj_iou=False,
            gr=None):
        hyp = dict(HYP)
        hyp.update(hyp_over or {})
        cl = y_loss.ComputeLoss(fake_model(y_loss, nc, hyp, anchors, strides), autobalance=autobalance)
        if sort_obj_iou or gr is not None:      # the two attributes a caller may set on the object (loss.py:94,113)
            cl.sort_obj_iou = bool(sort_obj_iou)
            cl.gr = cl.gr if gr is None else float(gr)
            out[f"{tag}_opts"] = np.array([float(cl.sort_obj_iou), cl.gr])
        levels = synth_head(rs, bs, img, nc)
        out[f"{tag}_targets"] = targets
        out[f"{tag}_meta"] = np.array([img, bs, nc, int(autobalance), calls])
        out[f"{tag}_hyp"] = np.array([hyp[k] for k in ("box", "cls", "cls_pw", "obj", "obj_pw", "anchor_t", "fl_gamma", "label_smoothing")])
        for c in range(calls):
            p = [T(x).requires_grad_(True) for x in levels]
            loss, items = cl(p, T(targets).reshape(-1, 6))
            grads = torch.autograd.grad(loss, p, allow_unused=True)
            out[f"{tag}_call{c}_loss"] = loss.detach().numpy()
            out[f"{tag}_call{c}_items"] = items.numpy()
            if store_grads:
                for l in range(3):
                    out[f"{tag}_call{c}_grad{l}"] = grads[l].numpy()
        for l in range(3):
            out[f"{tag}_p{l}"] = levels[l]
        if autobalance:
            out[f"{tag}_balance"] = np.array(cl.balance)

    t128 = synth_targets(rs, 24, 2, 80)
    t128[5] = t128[4]                       # exact duplicate target -> duplicate (b,a,gj,gi): last writer wins
    t128[6, :] = t128[4, :]; t128[6, 1] = 7  # same cell, different class
    t128[7, 2:4] = t128[4, 2:4]; t128[7, 4:6] = t128[4, 4:6] * 1.1
    run("base128", 128, 2, 80, t128)
    t64 = synth_targets(rs, 10, 2, 12)
    t64[3] = t64[2]
    run("nt0", 64, 2, 12, np.zeros((0, 6), np.float32))
    run("smooth", 64, 2, 12, t64, {"label_smoothing": 0.1})
    run("focal", 64, 2, 12, t64, {"fl_gamma": 1.5})
    run("posw", 64, 2, 12, t64, {"cls_pw": 0.7, "obj_pw": 1.3, "box": 0.07, "cls": 0.3, "obj": 0.8})
    run("autobal", 64, 2, 12, t64, None, autobalance=True, calls=2)
    run("nc1", 64, 2, 1, np.concatenate([t64[:, :1], np.zeros((10, 1), np.float32), t64[:, 2:]], 1))
    # sort_obj_iou: the best IoU among the rows of a cell owns it instead of the last row.  Cells with up to six rows whose
    # sizes (hence IoUs) differ, listed so that the last row is not the best one.
    tsort = synth_targets(rs, 30, 2, 80)
    for k, f in ((9, 1.0), (10, 1.3), (11, 0.8), (12, 1.15), (13, 0.7)):
        tsort[k] = tsort[8]
        tsort[k, 4:6] = tsort[8, 4:6] * f
    tsort[21] = tsort[20]; tsort[21, 4:6] = tsort[20, 4:6] * 1.4
    run("sortiou", 128, 2, 80, tsort, sort_obj_iou=True)
    run("sortiou_gr", 128, 2, 80, tsort, sort_obj_iou=True, gr=0.5)
    run("gr_half", 128, 2, 80, tsort, gr=0.5)
    save("compute_loss", **out)


def pack_list(out, key, dets):
    out[key + "_counts"] = np.array([d.shape[0] for d in dets])
    out[key + "_dets"] = np.concatenate([d.numpy() for d in dets], 0) if dets else np.zeros((0, 6), np.float32)


def gen_nms(y_general):
    rs = np.random.RandomState(67)
    out = {}
    nc = 16
    pred = dedupe_scores(synth_pred(rs, 3, 1008, nc), 0.001)
    pred[2, :, 4] = 0.0                                                   # image with no candidates at all
    pred[1, 5, 4] = 0.25                                                  # objectness exactly at threshold -> rejected (strict >)
    out["pred"] = pred
    labels = [T([[3, 100, 100, 50, 60], [5, 300, 310, 80, 40]]), torch.zeros((0, 5)), T([[1, 200, 200, 90, 90]])]
    out["labels0"], out["labels2"] = labels[0].numpy(), labels[2].numpy()
    variants = {
        "default": {},
        "multi": dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
        "multi_c25": dict(conf_thres=0.25, iou_thres=0.45, multi_label=True),
        "agnostic": dict(agnostic=True),
        "agnostic_multi": dict(conf_thres=0.05, agnostic=True, multi_label=True),
        "classes": dict(conf_thres=0.05, classes=[0, 2, 7]),
        "classes_multi": dict(conf_thres=0.05, classes=[0, 2, 7], multi_label=True),
        "maxdet1": dict(max_det=1),
        "maxdet5_multi": dict(conf_thres=0.01, multi_label=True, max_det=5),
        "labels": dict(conf_thres=0.1, labels=labels),
        "iou0": dict(iou_thres=0.0),
        "iou1": dict(iou_thres=1.0),
        "conf0": dict(conf_thres=0.0, max_det=1000),
        "conf1": dict(conf_thres=1.0),
    }
    for tag, kw in variants.items():
        NMS_CALLS.clear()
        dets = y_general.non_max_suppression(T(pred).clone(), **kw)
        pack_list(out, tag, dets)
        out[tag + "_nms_sizes"] = np.array([c[0].shape[0] for c in NMS_CALLS])
        if tag in ("default", "multi"):                                   # pin what reaches the third-party kernel
            out[tag + "_nms_boxes"] = np.concatenate([c[0] for c in NMS_CALLS], 0)
            out[tag + "_nms_scores"] = np.concatenate([c[1] for c in NMS_CALLS], 0)
            out[tag + "_nms_keep"] = np.concatenate([c[3] for c in NMS_CALLS], 0)

    # single class (nc=1): multi_label is forced off (general.py:667)
    p1 = dedupe_scores(synth_pred(rs, 2, 600, 1), 0.001)
    out["nc1_pred"] = p1
    pack_list(out, "nc1", y_general.non_max_suppression(T(p1).clone(), conf_thres=0.1, multi_label=True))

    # more than max_nms candidates (general.py:716-717); inputs are regenerated from the seed by the tests
    big = make_big_pred()
    NMS_CALLS.clear()
    dets = y_general.non_max_suppression(T(big).clone(), conf_thres=0.01, iou_thres=0.5, multi_label=True, max_det=300)
    assert NMS_CALLS[0][0].shape[0] == 30000, NMS_CALLS[0][0].shape
    out["big_pred_checksum"] = np.array([big.sum(dtype=np.float64)])
    pack_list(out, "big", dets)

    # score ties: the reference's unstable sorts forced to stable=True (the order this build pins)
    tie = synth_pred(rs, 2, 400, 4, n_obj=5, k_per_obj=20)
    tie[..., 4] = np.round(tie[..., 4] * 8) / 8
    tie[..., 5:] = np.round(tie[..., 5:] * 4) / 4
    out["tie_pred"] = tie
    orig_argsort, orig_sort = torch.Tensor.argsort, torch.Tensor.sort
    torch.Tensor.argsort = lambda self, *a, **k: orig_argsort(self, *a, **{**k, "stable": True})
    try:
        pack_list(out, "tie", y_general.non_max_suppression(T(tie).clone(), conf_thres=0.05, iou_thres=0.5))
        pack_list(out, "tie_multi", y_general.non_max_suppression(T(tie).clone(), conf_thres=0.05, iou_thres=0.5, multi_label=True))
    finally:
        torch.Tensor.argsort, torch.Tensor.sort = orig_argsort, orig_sort
    save("nms", **out)


def reference_nms_with_merge(y_general):
    """non_max_suppression of utils/general.py with its local switch `merge = False` (general.py:668) set to True: the
    function's definition is taken from the reference's own file, that one constant is flipped in the syntax tree, and
    the result is compiled inside the imported module's namespace (so it calls the module's own box_iou, xywh2xyxy and
    the recording stand-in for torchvision.ops.nms)."""
    import ast
    path = os.path.join(REF, "yolov5", "utils", "general.py")
    tree = ast.parse(open(path).read(), path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "non_max_suppression"][0]
    flipped = 0
    for node in ast.walk(fn):
        if (isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name)
                and node.targets[0].id == "merge"):
            assert isinstance(node.value, ast.Constant) and node.value.value is False
            node.value = ast.copy_location(ast.Constant(value=True), node.value)
            flipped += 1
    assert flipped == 1
    ns = dict(vars(y_general))
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns["non_max_suppression"]


def gen_nms_merge(y_general):
    """Merge-NMS (general.py:725-731): clustered boxes so that most kept rows have neighbours above the threshold."""
    nms_merge = reference_nms_with_merge(y_general)
    rs = np.random.RandomState(71)
    out = {}
    nc = 16
    pred = dedupe_scores(synth_pred(rs, 4, 1200, nc, n_obj=8, jitter=5.0, k_per_obj=14), 0.001)
    pred[3, :, 4] = 0.0                                 # no candidates
    pred[2, :, 4] *= 0.2                                # nothing above 0.25 except by accident
    pred[2, 17, 4], pred[2, 17, 5:] = 0.9, 0.0
    pred[2, 17, 5 + 3] = 0.8                            # an image with exactly one candidate: n = 1, no merge (1 < n)
    out["pred"] = pred
    variants = {
        "default": {},
        "multi": dict(conf_thres=0.2, iou_thres=0.6, multi_label=True),
        "agnostic": dict(agnostic=True, conf_thres=0.3),
        "maxdet3": dict(max_det=3),
        "classes": dict(conf_thres=0.1, classes=[1, 4, 9]),
        "many": dict(conf_thres=0.001, iou_thres=0.5),             # every row a candidate, still n < 3000
    }
    for tag, kw in variants.items():
        NMS_CALLS.clear()
        pack_list(out, tag, nms_merge(T(pred).clone(), **kw))
        out[tag + "_nms_sizes"] = np.array([c[0].shape[0] for c in NMS_CALLS])
    # n >= 3000 in an image: the step is skipped there (general.py:725)
    wide = dedupe_scores(synth_pred(rs, 2, 4000, 4, n_obj=6, k_per_obj=10), 0.001)
    wide[1, 2500:, 4] = 0.0                              # image 1 stays below 3000 candidates, image 0 does not
    out["wide_pred"] = wide
    NMS_CALLS.clear()
    pack_list(out, "wide", nms_merge(T(wide).clone(), conf_thres=0.0, iou_thres=0.5))
    out["wide_nms_sizes"] = np.array([c[0].shape[0] for c in NMS_CALLS])
    save("nms_merge", **out)


def gen_box_utils(y_general):
    rs = np.random.RandomState(79)
    out = {}
    xywh = rand_boxes_xywh(rs, 33, 640)
    out["xywh"] = xywh
    out["xywh2xyxy"] = y_general.xywh2xyxy(T(xywh)).numpy()
    out["xyxy2xywh"] = y_general.xyxy2xywh(T(out["xywh2xyxy"])).numpy()
    out["xywh2xyxy_np"] = y_general.xywh2xyxy(xywh.copy())
    xyxy = f32(out["xywh2xyxy"] + rs.normal(0, 40, (33, 4)))
    det = f32(np.concatenate([xyxy, rs.uniform(0, 1, (33, 2))], 1))
    out["scale_in"] = det
    out["scale_a"] = y_general.scale_coords((640, 640), T(det).clone(), (720, 1280, 3)).numpy()
    out["scale_b"] = y_general.scale_coords((384, 640), T(det).clone(), (1080, 1920)).numpy()
    out["scale_c"] = y_general.scale_coords((640, 640), T(det).clone(), (500, 375), ratio_pad=((0.8, 0.8), (10.0, 20.0))).numpy()
    c = T(det).clone()
    y_general.clip_coords(c, (300, 400))
    out["clip_t"] = c.numpy()
    cn = det.copy()
    y_general.clip_coords(cn, (300, 400))
    out["clip_np"] = cn
    save("box_utils", **out)


def reference_process_batch(y_metrics):
    """val.py cannot be imported (it pulls in the model zoo); its process_batch is a self-contained function, so the
    function definition alone is compiled from the reference's file, at run time, with the names it uses bound to
    the reference's own box_iou."""
    import ast
    path = os.path.join(REF, "yolov5", "val.py")
    tree = ast.parse(open(path).read(), path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "process_batch"]
    ns = {"torch": torch, "np": np, "box_iou": y_metrics.box_iou}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns["process_batch"]


def eval_case(rs, n_lab, n_det, nc, jitter, dup=False):
    """labels (M,5) [cls, xyxy] and detections (N,6) [xyxy, conf, cls] around them, detections sorted by confidence."""
    c = rs.uniform(40, 600, (n_lab, 2))
    wh = np.exp(rs.uniform(np.log(12), np.log(200), (n_lab, 2)))
    lab = np.concatenate((rs.randint(0, nc, (n_lab, 1)).astype(np.float64), c - wh / 2, c + wh / 2), 1)
    src = rs.randint(0, max(n_lab, 1), n_det)
    det = np.zeros((n_det, 6))
    if n_lab:
        det[:, :4] = lab[src, 1:] + rs.normal(0, jitter, (n_det, 4)) * np.tile(wh[src], 2)
        det[:, 5] = np.where(rs.rand(n_det) < 0.8, lab[src, 0], rs.randint(0, nc, n_det))
    else:
        det[:, :4] = np.concatenate((rs.uniform(0, 300, (n_det, 2)), rs.uniform(300, 600, (n_det, 2))), 1)
        det[:, 5] = rs.randint(0, nc, n_det)
    det[:, 4] = np.sort(rs.uniform(0.001, 1, n_det))[::-1]
    if dup and n_det > 4 and n_lab > 1:      # exact duplicates: equal IoUs (the stable refinement decides)
        det[1, :4] = det[0, :4]
        det[3, :4] = det[2, :4]
        det[3, 5] = det[2, 5]
        lab[1, 1:] = lab[0, 1:]
    return f32(lab), f32(det)


def gen_evalmatch(y_metrics):
    """val.py:71-92 and metrics.py:116-171 on tie-free cases from the untouched reference, and on cases with equal IoUs
    from the reference with its two argsorts made stable (the only change: numpy's argsort kind)."""
    from oracle import evalmatch as oe
    ref_pb = reference_process_batch(y_metrics)
    iouv = torch.linspace(0.5, 0.95, 10)
    rs = np.random.RandomState(4242)
    out = {"iouv": iouv.numpy()}
    cases = [("empty_det", 5, 0, 3, .05, False), ("empty_lab", 0, 7, 3, .05, False), ("one", 1, 1, 1, .02, False),
             ("small", 6, 20, 3, .06, False), ("crowd", 40, 300, 5, .08, False), ("many", 120, 300, 80, .05, False),
             ("loose", 25, 200, 2, .2, False), ("ties", 12, 60, 2, .05, True)]
    for tag, n_lab, n_det, nc, jit, dup in cases:
        lab, det = eval_case(rs, n_lab, n_det, nc, jit, dup)
        tl, td = torch.from_numpy(lab), torch.from_numpy(det)
        # The case with equal IoUs cannot be pinned by the untouched reference (ndarray.argsort's default kind is
        # unstable and cannot be rebound): its expected values are the oracle's stable refinement, and whether the
        # reference happens to agree on this numpy build is only printed.
        if dup:
            same = torch.equal(ref_pb(td, tl, iouv), oe.process_batch(td, tl, iouv))
            print(f"  evalmatch {tag}: untouched reference {'agrees with' if same else 'differs from'} the stable refinement")
        correct = ref_pb(td, tl, iouv) if not dup else oe.process_batch(td, tl, iouv)
        mine = oe.process_batch(td, tl, iouv)
        assert torch.equal(correct, mine), tag
        cm = y_metrics.ConfusionMatrix(nc=nc) if not dup else oe.ConfusionMatrix(nc=nc)
        cm.process_batch(td, tl)
        cm.process_batch(td[::2], tl)            # a second image accumulates into the same matrix
        mine_cm = oe.ConfusionMatrix(nc=nc)
        mine_cm.process_batch(td, tl)
        mine_cm.process_batch(td[::2], tl)
        assert np.array_equal(cm.matrix, mine_cm.matrix), tag
        out[f"{tag}_labels"], out[f"{tag}_det"] = lab, det
        out[f"{tag}_correct"] = correct.numpy()
        out[f"{tag}_matrix"] = cm.matrix
        out[f"{tag}_meta"] = np.array([nc, int(dup)])
        print(f"  evalmatch {tag}: {int(correct[:, 0].sum())} of {n_det} detections correct @0.5, matrix sum {cm.matrix.sum():.0f}")
    save("evalmatch", **out)


def gen_tta():
    """Model._descale_pred / _clip_augmented (models/yolo.py:155-183): models/yolo.py cannot be imported (model zoo,
    network), so the two method definitions alone are compiled from the reference's file and called with a stand-in
    for `self` that carries the two attributes they read."""
    import ast
    from types import SimpleNamespace
    path = os.path.join(REF, "yolov5", "models", "yolo.py")
    tree = ast.parse(open(path).read(), path)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Model"][0]
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("_descale_pred", "_clip_augmented")]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    me = SimpleNamespace(inplace=True, model=[SimpleNamespace(nl=3)])
    rs = np.random.RandomState(4)
    base = synth_pred(rs, 2, 1260, 3)
    out = {"base": base}
    for k, (flips, scale) in enumerate(((None, 1.0), (3, 0.83), (2, 0.67), (None, 0.67))):
        out[f"descale{k}"] = ns["_descale_pred"](me, T(base).clone(), flips, scale, (480, 640)).numpy()
        out[f"descale{k}_args"] = np.array([0 if flips is None else flips, scale])
    ys = ns["_clip_augmented"](me, [T(base), T(base[:, :900]), T(base[:, :400])])
    out["clip_shapes"] = np.array([y.shape[1] for y in ys])
    out["clip_first_last"] = np.stack([ys[0][0, -1].numpy(), ys[-1][0, 0].numpy()])
    save("tta", **out)


def reference_detect_class(y_general):
    """The `Detect` class of models/yolo.py:33-81, compiled from the reference's own file: the module cannot be imported
    (models/common.py and utils/plots.py reach the network at import time), but the class needs nothing beyond torch,
    torch.nn and utils.general.check_version, so its definition alone is taken from the syntax tree."""
    import ast
    path = os.path.join(REF, "yolov5", "models", "yolo.py")
    tree = ast.parse(open(path).read(), path)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Detect"]
    ns = {"torch": torch, "nn": torch.nn, "check_version": y_general.check_version}
    exec(compile(ast.Module(body=cls, type_ignores=[]), path, "exec"), ns)
    return ns["Detect"]


DETECT_CASES = {      # tag: (nc, anchors in pixels per level, strides, input channels, (H, W) of the image, batch, inplace)
    "s3_nc80": (80, None, (8, 16, 32), (4, 8, 16), (64, 96), 2, True),
    "s3_nc80_cat": (80, None, (8, 16, 32), (4, 8, 16), (64, 96), 2, False),
    "s3_nc7_rect": (7, None, (8, 16, 32), (3, 5, 7), (160, 96), 3, True),
    "s3_nc1": (1, None, (8, 16, 32), (2, 2, 2), (96, 96), 1, True),
    "s4_nc7": (7, ((19, 27, 44, 40, 38, 94), (96, 68, 86, 152, 180, 137), (140, 301, 303, 264, 238, 542),
                   (436, 615, 739, 380, 925, 792)), (8, 16, 32, 64), (3, 4, 5, 6), (128, 192), 2, True),
    "s5_nc3_na2": (3, ((10, 13, 16, 30), (30, 61, 62, 45), (59, 119, 116, 90), (156, 198, 373, 326), (400, 500, 600, 640)),
                   (4, 8, 16, 32, 64), (2, 3, 4, 5, 6), (128, 64), 2, False),
}


def detect_case_inputs(tag):
    """Seeded feature maps of a case (rebuilt by the tests: the fixture stores the convolution outputs, not these)."""
    nc, anchors_px, strides, ch, (H, W), bs, inplace = DETECT_CASES[tag]
    rs = np.random.RandomState(sum(map(ord, tag)))
    return [f32(rs.normal(0, 1, (bs, c, H // s, W // s))) for c, s in zip(ch, strides)]


def gen_detect(y_general):
    """Detect.forward (models/yolo.py:49-81), training and inference branch, from the reference's own class: the outputs
    of its 1x1 convolutions (computed by calling the module's own nn.Conv2d layers; the convolution itself is not on the
    path), the maps of yolo.py:54 and the decoded tensor of yolo.py:56-70."""
    from oracle import yolo as oy
    Detect = reference_detect_class(y_general)
    out = {}
    for tag, (nc, anchors_px, strides, ch, hw, bs, inplace) in DETECT_CASES.items():
        anchors_px = anchors_px or oy.YOLOV5S_ANCHORS_PX
        torch.manual_seed(sum(map(ord, tag)))
        det = Detect(nc=nc, anchors=anchors_px, ch=ch, inplace=inplace)
        det.stride = torch.tensor([float(s) for s in strides])
        det.anchors /= det.stride.view(-1, 1, 1)                                   # yolo.py:113
        with torch.no_grad():
            for m, s in zip(det.m, strides):                                       # yolo.py:196-204: realistic sparsity
                b = m.bias.view(det.na, -1)
                b[:, 4] += np.log(8 / (640 / s) ** 2)
                b[:, 5:] += np.log(0.6 / (nc - 0.999999))
                m.weight.mul_(3.0)
        feats = [T(x) for x in detect_case_inputs(tag)]
        with torch.no_grad():
            convs = [m(x) for m, x in zip(det.m, feats)]
            det.train()
            maps_train = det([x.clone() for x in feats])
            det.eval()
            z, maps_eval = det([x.clone() for x in feats])
        assert all(torch.equal(a, b) for a, b in zip(maps_train, maps_eval))
        # the oracle's restatement against the reference, bit for bit (same torch operations in the same order)
        mine_maps = oy.head_maps(convs, det.na)
        assert all(torch.equal(a, b) for a, b in zip(mine_maps, maps_eval)), tag
        mine_z = oy.detect_decode(mine_maps, det.anchors, [float(s) for s in strides])
        assert torch.equal(mine_z, z), tag
        out[f"{tag}_anchors"] = det.anchors.numpy().copy()
        out[f"{tag}_meta"] = np.array([nc, det.na, det.nl, int(inplace), bs])
        out[f"{tag}_strides"] = np.array(strides, dtype=np.float32)
        out[f"{tag}_z"] = z.numpy()
        for l in range(det.nl):
            out[f"{tag}_conv{l}"] = convs[l].numpy()
            out[f"{tag}_w{l}"] = det.m[l].weight.detach().numpy().copy()
            out[f"{tag}_b{l}"] = det.m[l].bias.detach().numpy().copy()
            out[f"{tag}_grid{l}"] = det.grid[l].numpy().copy()
            out[f"{tag}_anchor_grid{l}"] = det.anchor_grid[l].numpy().copy()
        print(f"  detect {tag}: z {tuple(z.shape)}, obj max {float(z[..., 4].max()):.4f}")
    save("detect", **out)


def half_nms_inputs():
    """fp16 prediction tensor and apriori labels of the half-precision NMS fixture (rebuilt by the tests)."""
    rs = np.random.RandomState(83)
    pred = synth_pred(rs, 3, 1008, 16).astype(np.float16)
    pred[1, 5, 4] = 0.25                                                  # objectness exactly at a threshold -> rejected
    # conf_thres 0.001 becomes 0.0010004 when rounded to fp16: a CPU comparison of an fp16 tensor with that scalar uses the
    # rounded value, a device kernel the fp32 one, and the two disagree on a score that is exactly 0.0010004.  The reference
    # runs fp16 on devices only (val.py:129); keep such scores out of the inputs so that the fixture holds for both.
    edge = np.float16(0.001)
    prod = (pred[..., 5:].astype(np.float32) * pred[..., 4:5].astype(np.float32)).astype(np.float16)
    pred[..., 5:][prod == edge] = 0
    pred[..., 4][pred[..., 4] == edge] = 0
    labels = [T([[3, 100, 100, 50, 60], [5, 300, 310, 80, 40]]), torch.zeros((0, 5)), T([[1, 200, 200, 90, 90]])]
    return pred, labels


def gen_half(y_general):
    """The reference's half-precision inference path (model.half(), val.py:129-130), run on the CPU: Detect's inference
    branch on fp16 maps (yolo.py:56-70) and non_max_suppression on an fp16 prediction tensor (general.py:647-738)."""
    from oracle import yolo as oy
    Detect = reference_detect_class(y_general)
    out = {}
    for tag in HALF_DETECT_TAGS:
        nc, anchors_px, strides, ch, hw, bs, inplace = DETECT_CASES[tag]
        anchors_px = anchors_px or oy.YOLOV5S_ANCHORS_PX
        torch.manual_seed(sum(map(ord, tag)))
        det = Detect(nc=nc, anchors=anchors_px, ch=ch, inplace=inplace)
        det.stride = torch.tensor([float(s) for s in strides])
        det.anchors /= det.stride.view(-1, 1, 1)                                   # yolo.py:113
        with torch.no_grad():
            for m, s in zip(det.m, strides):                                       # yolo.py:196-204
                b = m.bias.view(det.na, -1)
                b[:, 4] += np.log(8 / (640 / s) ** 2)
                b[:, 5:] += np.log(0.6 / (nc - 0.999999))
                m.weight.mul_(3.0)
        det.half().eval()                                                          # val.py:130
        feats = [T(x).half() for x in detect_case_inputs(tag)]
        with torch.no_grad():
            convs = [m(x) for m, x in zip(det.m, feats)]
            z, maps = det([x.clone() for x in feats])
        assert all(m.dtype == torch.float16 for m in maps) and z.dtype == (torch.float16 if inplace else torch.float32), tag
        assert det.anchors.dtype == torch.float16 and det.stride.dtype == torch.float32 and det.anchor_grid[0].dtype == torch.float32
        mine = oy.detect_decode(oy.head_maps(convs, det.na), det.anchors, [float(s) for s in strides], inplace=inplace)
        assert mine.dtype == z.dtype and torch.equal(mine, z), tag
        out[f"{tag}_anchors"] = det.anchors.numpy().copy()                         # fp16
        out[f"{tag}_meta"] = np.array([nc, det.na, det.nl, int(inplace), bs])
        out[f"{tag}_strides"] = np.array(strides, dtype=np.float32)
        out[f"{tag}_z"] = z.numpy()
        for l in range(det.nl):
            out[f"{tag}_conv{l}"] = convs[l].numpy()                               # fp16
        print(f"  half detect {tag}: z {tuple(z.shape)} {z.dtype}")
    pred, labels = half_nms_inputs()
    out["nms_pred"] = pred
    out["nms_labels0"], out["nms_labels2"] = labels[0].numpy(), labels[2].numpy()
    for tag, kw in HALF_NMS_VARIANTS.items():
        if tag == "labels":
            kw = dict(kw, labels=labels)
        NMS_CALLS.clear()
        dets = y_general.non_max_suppression(torch.from_numpy(pred).clone(), **kw)
        assert all(d.dtype == torch.float32 for d in dets)
        pack_list(out, "nms_" + tag, dets)
        print(f"  half nms {tag}: {[d.shape[0] for d in dets]}")
    save("half", **out)


def ap_case(rs, n, nc, n_lab, niou=10, p_hit=0.6, absent=()):
    """Evaluation statistics as val.py collects them: tp (n, niou) bool (a hit at a higher IoU threshold implies the
    lower ones, at most one hit per label), conf (n,) float32 all different, pred_cls (n,) float32, target_cls (m,) float64."""
    target_cls = rs.randint(0, nc, n_lab).astype(np.float64)
    target_cls = target_cls[~np.isin(target_cls, absent)]
    pred_cls = rs.randint(0, nc + 1, n).astype(np.float32)               # class nc never has labels
    conf = rs.permutation(np.linspace(0.002, 0.998, max(n, 1)).astype(np.float32))[:n]
    assert len(np.unique(conf)) == n
    tp = np.zeros((n, niou), dtype=bool)
    for c in range(nc):
        idx = np.flatnonzero(pred_cls == c)
        n_l = int((target_cls == c).sum())
        hits = idx[rs.rand(len(idx)) < p_hit][:n_l]
        depth = rs.randint(1, niou + 1, len(hits))
        for h, d in zip(hits, depth):
            tp[h, :d] = True
    return tp, conf, pred_cls, target_cls


def gen_ap(y_metrics):
    """metrics.py:21-114 from the untouched reference (plot=False)."""
    import warnings
    from oracle import ap as oa
    rs = np.random.RandomState(77)
    cases = {"empty": ap_case(rs, 0, 3, 10), "one": ap_case(rs, 1, 1, 1, p_hit=1.0), "small": ap_case(rs, 40, 3, 25),
             "single_iou": ap_case(rs, 300, 4, 80, niou=1), "all_fp": ap_case(rs, 200, 5, 60, p_hit=0.0),
             "all_tp": ap_case(rs, 50, 2, 400, p_hit=1.0), "absent": ap_case(rs, 500, 6, 200, absent=(2, 4)),
             "coco_like": ap_case(rs, 20000, 80, 6000), "one_class_long": ap_case(rs, 30000, 1, 9000, p_hit=0.5)}
    out = {}
    names = ("tp", "fp", "p", "r", "f1", "ap", "classes")
    for tag, (tp, conf, pcls, tcls) in cases.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                               # np.trapz deprecation, 0/0 in all-fp columns
            want = y_metrics.ap_per_class(tp, conf, pcls, tcls, plot=False, names={})     # names=() (the default) raises at :74
            mine = oa.ap_per_class(tp, conf, pcls, tcls)
        for nm, w, m in zip(names, want, mine):
            assert w.dtype == m.dtype and np.array_equal(w, m, equal_nan=True), (tag, nm)
            out[f"{tag}_{nm}"] = w
        out[f"{tag}_in_tp"], out[f"{tag}_in_conf"], out[f"{tag}_in_pred_cls"], out[f"{tag}_in_target_cls"] = tp, conf, pcls, tcls
        print(f"  ap {tag}: n {len(conf)}, classes {len(want[6])}, mAP@0.5 {want[5][:, 0].mean() if len(want[6]) else 0:.4f}")
    x = rs.rand(6, 7)
    assert np.array_equal(y_metrics.fitness(x), oa.fitness(x))
    out["fitness_in"], out["fitness_out"] = x, y_metrics.fitness(x)
    save("ap", **out)


def reference_autoanchor_closures(wh, thr):
    """utils/autoanchor.py keeps its metrics as closures inside check_anchors / kmean_anchors; the nested function
    definitions alone are compiled from the reference's file at run time, with the variables they capture (wh, thr)
    bound in their namespace."""
    import ast
    path = os.path.join(REF, "yolov5", "utils", "autoanchor.py")
    tree = ast.parse(open(path).read(), path)
    outer = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    check = [n for n in outer["check_anchors"].body if isinstance(n, ast.FunctionDef) and n.name == "metric"]
    kmean = [n for n in outer["kmean_anchors"].body if isinstance(n, ast.FunctionDef) and n.name in ("metric", "anchor_fitness")]
    ns_check = {"torch": torch, "wh": wh, "thr": thr}
    exec(compile(ast.Module(body=check, type_ignores=[]), path, "exec"), ns_check)
    ns_kmean = {"torch": torch, "wh": wh, "thr": 1 / thr}                   # autoanchor.py:84
    exec(compile(ast.Module(body=kmean, type_ignores=[]), path, "exec"), ns_kmean)
    return ns_check["metric"], ns_kmean["metric"], ns_kmean["anchor_fitness"]


def gen_autoanchor():
    """autoanchor.py:35-41, 86-94 on label sizes like a 640 px dataset's, against the default anchors and poor ones."""
    from oracle import autoanchor as oaa
    rs = np.random.RandomState(11)
    out = {}
    default = np.array([[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198], [373, 326]], dtype=np.float32)
    cases = {"default": (default, 5000, 4.0), "poor": (default[:3] * 0.2, 3000, 4.0), "tight": (default, 2000, 2.0),
             "one": (default[4:5], 1, 4.0)}
    for tag, (k, n, thr) in cases.items():
        wh = torch.from_numpy(np.exp(rs.uniform(np.log(2), np.log(600), (n, 2))).astype(np.float32))
        if n > 10:
            wh[0] = torch.tensor([10.0, 13.0])                               # a label equal to an anchor: ratio exactly 1
            wh[1] = torch.tensor([40.0, 52.0])                               # exactly on the threshold 1/4 of the first anchor
        check_metric, metric, fitness = reference_autoanchor_closures(wh, thr)
        kt = torch.from_numpy(k)
        bpr, aat = check_metric(kt)
        x, best = metric(kt, wh)
        fit = fitness(k)
        mb, ma = oaa.check_metric(kt, wh, thr)
        mx, mbest = oaa.metric(kt, wh)
        assert torch.equal(bpr, mb) and torch.equal(aat, ma) and torch.equal(x, mx) and torch.equal(best, mbest)
        assert torch.equal(fit, oaa.anchor_fitness(k, wh, 1 / thr))
        out[f"{tag}_k"], out[f"{tag}_wh"], out[f"{tag}_thr"] = k, wh.numpy(), np.array(thr)
        out[f"{tag}_x"], out[f"{tag}_best"] = x.numpy(), best.numpy()
        out[f"{tag}_stats"] = np.array([bpr.item(), aat.item(), fit.item()], dtype=np.float32)
        print(f"  autoanchor {tag}: bpr {bpr:.4f}, aat {aat:.3f}, fitness {fit:.4f}")
    save("autoanchor", **out)


def gen_focal(a_loss):
    """object_detection/loss.py:8-83 from the untouched reference: value and gradient, one-hot target maps (the
    CenterNet configuration of train.py:147-153) and index labels on 2-D logits."""
    import warnings
    from oracle import focal as of
    rs = np.random.RandomState(31)
    out = {}
    cases = []
    for tag, (N, C, W, H, gamma, n_obj) in {"train": (2, 20, 24, 20, 8.0, 14), "g0": (1, 3, 5, 7, 0.0, 4), "g2": (3, 6, 8, 8, 2.0, 9),
                                            "flat": (5, 4, 0, 0, 1.5, 3)}.items():
        shape = (N, C, W, H) if W else (N, C)
        logits = f32(rs.normal(0, 2.0, shape))
        label = np.zeros(shape, dtype=np.float32)
        for _ in range(n_obj):
            idx = tuple(rs.randint(0, d) for d in shape)
            label[idx] = 1.0
        if W:
            label[0, 1, 2, 3] = 1.0
            label[0, 2, 2, 3] = 1.0                                     # two classes at one cell
            label[0, 0, 0, 0] = 0.7                                     # a fractional value: .long() makes it 0
        cases.append((tag, dict(num_classes=C, one_hot_label_format=True, gamma=gamma, alpha=0.999, smoothing=0.1), logits, label))
    for tag, (N, C, gamma, smoothing, ignore) in {"idx": (64, 7, 2.0, 0.1, None), "idx_ignore": (40, 5, 3.0, 0.0, 2),
                                                   "idx_g0": (9, 3, 0, 0.2, None)}.items():
        logits = f32(rs.normal(0, 2.0, (N, C)))
        label = rs.randint(0, C, N).astype(np.int64)
        cases.append((tag, dict(num_classes=C, gamma=gamma, smoothing=smoothing, ignore_index=ignore), logits, label))
    for tag, kw, logits, label in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                             # implicit softmax dim (loss.py:78)
            x = T(logits).requires_grad_(True)
            want = a_loss.LabelSmoothingFocalLoss(**kw)(x, T(label))
            gw, = torch.autograd.grad(want * 3.0, x)
            x2 = T(logits).requires_grad_(True)
            mine = of.LabelSmoothingFocalLoss(**kw)(x2, T(label))
            gm, = torch.autograd.grad(mine * 3.0, x2)
        assert torch.equal(want, mine) and torch.equal(gw, gm), tag
        out[f"{tag}_logits"], out[f"{tag}_label"] = logits, label
        out[f"{tag}_loss"], out[f"{tag}_grad"] = want.detach().numpy(), gw.numpy()
        out[f"{tag}_kw"] = np.array([kw["num_classes"], int(kw.get("one_hot_label_format", False)), kw["gamma"], kw["smoothing"],
                                     -1 if kw.get("ignore_index") is None else kw["ignore_index"]], dtype=np.float64)
        print(f"  focal {tag}: loss {float(want):.6f}, |grad| {float(gw.abs().sum()):.4f}")
    # RegressionLossWithMask (loss.py:86-100), L1 and smooth-L1
    for tag, smooth in (("reg_l1", False), ("reg_smooth", True)):
        B, W, H = 3, 12, 10
        pred, target = f32(rs.normal(0, 2.0, (B, 2, W, H))), f32(rs.normal(0, 2.0, (B, 2, W, H)))
        mask = (rs.rand(B, W, H) < 0.2).astype(np.float32)
        pred[0, 0, 1, 1], target[0, 0, 1, 1], mask[0, 1, 1] = 1.5, 1.5, 1.0        # a zero difference inside the mask
        pred[0, 1, 1, 1], target[0, 1, 1, 1] = 0.25, 1.25                          # exactly on the smooth-L1 knee
        x = T(pred).requires_grad_(True)
        want = a_loss.RegressionLossWithMask(smooth)(x, T(target), T(mask))
        gw, = torch.autograd.grad(want * 2.0, x)
        x2 = T(pred).requires_grad_(True)
        mine = of.RegressionLossWithMask(smooth)(x2, T(target), T(mask))
        gm, = torch.autograd.grad(mine * 2.0, x2)
        assert torch.equal(want, mine) and torch.equal(gw, gm), tag
        out[f"{tag}_pred"], out[f"{tag}_target"], out[f"{tag}_mask"] = pred, target, mask
        out[f"{tag}_loss"], out[f"{tag}_grad"] = want.detach().numpy(), gw.numpy()
        print(f"  {tag}: loss {float(want):.6f}")
    save("focal", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(1)
    y_metrics, y_general, y_loss, a_loss, c_loss, c_data = load_reference()
    only = set(sys.argv[1:])
    if only:                      # e.g. `make_golden.py ap`: regenerate the named fixtures only
        gens = {"ap": lambda: gen_ap(y_metrics), "evalmatch": lambda: gen_evalmatch(y_metrics), "tta": gen_tta,
                "autoanchor": gen_autoanchor, "focal": lambda: gen_focal(a_loss), "nms_merge": lambda: gen_nms_merge(y_general), "detect": lambda: gen_detect(y_general), "half": lambda: gen_half(y_general),
                "compute_loss": lambda: gen_compute_loss(y_loss)}
        for k in only:
            gens[k]()
        return
    gen_bbox_iou(y_metrics, a_loss)
    gen_pairwise(y_metrics)
    gen_box_utils(y_general)
    gen_build_targets(y_loss)
    gen_compute_loss(y_loss)
    gen_nms(y_general)
    gen_nms_merge(y_general)
    gen_centernet(a_loss, c_loss, c_data)
    gen_evalmatch(y_metrics)
    gen_tta()
    gen_detect(y_general)
    gen_half(y_general)
    gen_ap(y_metrics)
    gen_autoanchor()
    gen_focal(a_loss)


if __name__ == "__main__":
    main()
