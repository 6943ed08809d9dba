"""The CPU oracle must reproduce what the reference's own code produced (tests/golden/*.npz).

Integer / index / mask outputs: bit-exact.  Float outputs: the oracle keeps the
reference's operation order, so they are compared with rtol 1e-6 (they are in
fact bitwise equal on the same torch build; the tolerance only absorbs a
different CPU vectorisation of exp/atan).
"""
import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import centernet as oc
from oracle import nms as onms
from oracle import yolo as oy
from tests._synth import make_big_pred

RT = 1e-6
MODE_FLAGS = {"iou": {}, "giou": {"GIoU": True}, "diou": {"DIoU": True}, "ciou": {"CIoU": True}}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def close(a, b, rtol=RT, atol=1e-7):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag", ["n1", "n257", "n2048", "special"])
@pytest.mark.parametrize("fmt", ["xywh", "xyxy"])
@pytest.mark.parametrize("mode", list(MODE_FLAGS))
def test_bbox_iou_yolo(golden, tag, fmt, mode):
    g = golden("bbox_iou")
    b1 = T(g[f"{tag}_{fmt}_box1"]).T.contiguous().requires_grad_(True)
    val = ob.bbox_iou_yolo(b1, T(g[f"{tag}_{fmt}_box2"]), x1y1x2y2=(fmt == "xyxy"), **MODE_FLAGS[mode])
    grad, = torch.autograd.grad(val.sum(), b1)
    close(val.detach(), g[f"{tag}_{fmt}_{mode}_out"])
    close(grad, g[f"{tag}_{fmt}_{mode}_grad"], atol=1e-6)


def test_bbox_iou_lastdim_and_sanity(golden):
    g = golden("bbox_iou")
    for mode, flags in MODE_FLAGS.items():
        b1 = T(g["lastdim_box1"]).requires_grad_(True)
        val = ob.bbox_iou_lastdim(b1, T(g["lastdim_box2"]), **flags)
        grad, = torch.autograd.grad(val.sum(), b1)
        close(val.detach(), g[f"lastdim_{mode}_out"])
        close(grad, g[f"lastdim_{mode}_grad"], atol=1e-6)
    v = ob.bbox_iou_yolo(T(np.float32([[.5, .5, .2, .3]])).T, T(np.float32([[.52, .48, .25, .28]])), False, CIoU=True)
    close(v, g["sanity_ciou"])
    assert abs(float(v) - 0.705263614654541) < 1e-6


@pytest.mark.parametrize("tag", ["0x5", "7x5", "100x300", "65x129"])
def test_pairwise(golden, tag):
    g = golden("pairwise")
    a, b = g[f"box_iou_{tag}_a"], g[f"box_iou_{tag}_b"]
    close(ob.box_iou(T(a).reshape(-1, 4), T(b)), g[f"box_iou_{tag}_out"])
    close(ob.wh_iou(T(g[f"wh_iou_{tag}_a"]).reshape(-1, 2), T(g[f"wh_iou_{tag}_b"])), g[f"wh_iou_{tag}_out"])
    close(ob.bbox_ioa(g["ioa_box1"], g["ioa_box2"]), g["ioa_out"])


def test_box_utils(golden):
    g = golden("box_utils")
    close(ob.xywh2xyxy(T(g["xywh"])), g["xywh2xyxy"], 0, 0)
    close(ob.xyxy2xywh(T(g["xywh2xyxy"])), g["xyxy2xywh"], 0, 0)
    close(ob.xywh2xyxy(g["xywh"].copy()), g["xywh2xyxy_np"], 0, 0)
    close(ob.scale_coords((640, 640), T(g["scale_in"]).clone(), (720, 1280, 3)), g["scale_a"], 0, 0)
    close(ob.scale_coords((384, 640), T(g["scale_in"]).clone(), (1080, 1920)), g["scale_b"], 0, 0)
    close(ob.scale_coords((640, 640), T(g["scale_in"]).clone(), (500, 375), ratio_pad=((0.8, 0.8), (10.0, 20.0))),
          g["scale_c"], 0, 0)
    c = T(g["scale_in"]).clone()
    ob.clip_coords(c, (300, 400))
    close(c, g["clip_t"], 0, 0)
    cn = g["scale_in"].copy()
    ob.clip_coords(cn, (300, 400))
    close(cn, g["clip_np"], 0, 0)


@pytest.mark.parametrize("tag", ["nt0", "nt1", "nt80", "nt640", "edges", "rect"])
def test_build_targets(golden, tag):
    g = golden("build_targets")
    shapes = [tuple(int(v) for v in r) for r in g[f"{tag}_shapes"]]
    res = oy.build_targets(shapes, g[f"{tag}_targets"], g["anchors"], 4.0)
    for l, r in enumerate(res):
        idx = np.stack((r["b"], r["a"], r["gj"], r["gi"]), 1).reshape(-1, 4)
        np.testing.assert_array_equal(idx, g[f"{tag}_l{l}_idx"])
        np.testing.assert_array_equal(r["tcls"], g[f"{tag}_l{l}_tcls"])
        np.testing.assert_array_equal(r["tbox"], g[f"{tag}_l{l}_tbox"])       # bit-exact fp32
        np.testing.assert_array_equal(r["anch"], g[f"{tag}_l{l}_anch"])


HYP_KEYS = ("box", "cls", "cls_pw", "obj", "obj_pw", "anchor_t", "fl_gamma", "label_smoothing")


LOSS_TAGS = ["base128", "nt0", "smooth", "focal", "posw", "autobal", "nc1", "sortiou", "sortiou_gr", "gr_half"]


@pytest.mark.parametrize("tag", LOSS_TAGS)
def test_compute_loss(golden, tag):
    g = golden("compute_loss")
    img, bs, nc, autobal, calls = (int(v) for v in g[f"{tag}_meta"])
    hyp = dict(zip(HYP_KEYS, (float(v) for v in g[f"{tag}_hyp"])))
    crit = oy.YoloLossOracle(T(g["anchors"]), hyp, nc=nc, autobalance=bool(autobal))
    if f"{tag}_opts" in g:                      # attributes set on the object after construction (loss.py:94,113)
        crit.sort_obj_iou, crit.gr = bool(g[f"{tag}_opts"][0]), float(g[f"{tag}_opts"][1])
    for c in range(calls):
        p = [T(g[f"{tag}_p{l}"]).requires_grad_(True) for l in range(3)]
        loss, items = crit(p, T(g[f"{tag}_targets"]).reshape(-1, 6))
        grads = torch.autograd.grad(loss, p)
        close(loss.detach(), g[f"{tag}_call{c}_loss"])
        close(items, g[f"{tag}_call{c}_items"])
        for l in range(3):
            close(grads[l], g[f"{tag}_call{c}_grad{l}"], atol=1e-9)
    if autobal:
        close(np.array(crit.balance), g[f"{tag}_balance"])


def test_compute_loss_sort_obj_iou_case_discriminates(golden):
    """The sorted case would not pass with last-writer-wins: its objectness loss differs when the flag is off."""
    g = golden("compute_loss")
    tag = "sortiou"
    nc = int(g[f"{tag}_meta"][2])
    hyp = dict(zip(HYP_KEYS, (float(v) for v in g[f"{tag}_hyp"])))
    crit = oy.YoloLossOracle(T(g["anchors"]), hyp, nc=nc)
    _, items = crit([T(g[f"{tag}_p{l}"]) for l in range(3)], T(g[f"{tag}_targets"]).reshape(-1, 6))
    want = g[f"{tag}_call0_items"]
    assert abs(float(items[1]) - float(want[1])) > 1e-4 * float(want[1])
    np.testing.assert_allclose(items[0].numpy(), want[0], rtol=1e-6)


def unpack(g, key):
    counts = g[key + "_counts"]
    dets = g[key + "_dets"]
    out, s = [], 0
    for c in counts:
        out.append(dets[s:s + c])
        s += c
    return out


NMS_VARIANTS = {
    "default": {},
    "multi": dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
    "multi_c25": dict(conf_thres=0.25, iou_thres=0.45, multi_label=True),
    "agnostic": dict(agnostic=True),
    "agnostic_multi": dict(conf_thres=0.05, agnostic=True, multi_label=True),
    "classes": dict(conf_thres=0.05, classes=[0, 2, 7]),
    "classes_multi": dict(conf_thres=0.05, classes=[0, 2, 7], multi_label=True),
    "maxdet1": dict(max_det=1),
    "maxdet5_multi": dict(conf_thres=0.01, multi_label=True, max_det=5),
    "labels": dict(conf_thres=0.1),
    "iou0": dict(iou_thres=0.0),
    "iou1": dict(iou_thres=1.0),
    "conf0": dict(conf_thres=0.0, max_det=1000),
    "conf1": dict(conf_thres=1.0),
}


def nms_labels(g):
    return [T(g["labels0"]), torch.zeros((0, 5)), T(g["labels2"])]


@pytest.mark.parametrize("tag", list(NMS_VARIANTS))
def test_non_max_suppression(golden, tag):
    g = golden("nms")
    kw = dict(NMS_VARIANTS[tag])
    if tag == "labels":
        kw["labels"] = nms_labels(g)
    got = onms.non_max_suppression(T(g["pred"]).clone(), **kw)
    want = unpack(g, tag)
    assert [d.shape[0] for d in got] == [w.shape[0] for w in want]
    for d, w in zip(got, want):
        np.testing.assert_array_equal(d.numpy(), w)


MERGE_VARIANTS = {
    "default": {},
    "multi": dict(conf_thres=0.2, iou_thres=0.6, multi_label=True),
    "agnostic": dict(agnostic=True, conf_thres=0.3),
    "maxdet3": dict(max_det=3),
    "classes": dict(conf_thres=0.1, classes=[1, 4, 9]),
    "many": dict(conf_thres=0.001, iou_thres=0.5),
}


def merge_close(got, want):
    """Counts, scores and classes exactly; the merged boxes are fp32 weighted sums whose order of summation the
    reference leaves to torch.mm, so they are compared to rtol 1e-5."""
    assert [int(d.shape[0]) for d in got] == [int(w.shape[0]) for w in want]
    for d, w in zip(got, want):
        d = d.cpu().numpy()
        np.testing.assert_array_equal(d[:, 4:], w[:, 4:])
        np.testing.assert_allclose(d[:, :4], w[:, :4], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("tag", list(MERGE_VARIANTS))
def test_merge_nms(golden, tag):
    """general.py:725-731 run by the reference with its local switch set (tests/golden/make_golden.py)."""
    g = golden("nms_merge")
    merge_close(onms.non_max_suppression(T(g["pred"]).clone(), merge=True, **MERGE_VARIANTS[tag]), unpack(g, tag))
    if tag == "default":        # the step changed something: without it the result differs
        plain = onms.non_max_suppression(T(g["pred"]).clone())
        assert [d.shape[0] for d in plain] != [w.shape[0] for w in unpack(g, tag)]


def test_merge_nms_skips_images_with_3000_candidates(golden):
    g = golden("nms_merge")
    assert g["wide_nms_sizes"][0] >= 3000 > g["wide_nms_sizes"][1] > 1
    got = onms.non_max_suppression(T(g["wide_pred"]).clone(), conf_thres=0.0, iou_thres=0.5, merge=True)
    merge_close(got, unpack(g, "wide"))
    plain = onms.non_max_suppression(T(g["wide_pred"]).clone(), conf_thres=0.0, iou_thres=0.5)
    np.testing.assert_array_equal(got[0].numpy(), plain[0].numpy())       # image 0 untouched


def test_nms_special_inputs(golden):
    g = golden("nms")
    got = onms.non_max_suppression(T(g["nc1_pred"]).clone(), conf_thres=0.1, multi_label=True)
    for d, w in zip(got, unpack(g, "nc1")):
        np.testing.assert_array_equal(d.numpy(), w)
    for key, kw in (("tie", {}), ("tie_multi", {"multi_label": True})):
        got = onms.non_max_suppression(T(g["tie_pred"]).clone(), conf_thres=0.05, iou_thres=0.5, **kw)
        for d, w in zip(got, unpack(g, key)):
            np.testing.assert_array_equal(d.numpy(), w)


def test_nms_more_than_max_nms(golden):
    g = golden("nms")
    big = make_big_pred()
    assert big.sum(dtype=np.float64) == g["big_pred_checksum"][0]
    got = onms.non_max_suppression(T(big).clone(), conf_thres=0.01, iou_thres=0.5, multi_label=True, max_det=300)
    np.testing.assert_array_equal(got[0].numpy(), unpack(g, "big")[0])


def test_nms_known_answers():
    """Hand-computed cases for the third-party greedy kernel (no reference fixture exists for it)."""
    f = np.float32
    sq = lambda x, y, s: [x, y, x + s, y + s]
    # disjoint boxes: all kept, score order
    k = onms.nms(T(f([sq(0, 0, 10), sq(20, 20, 10), sq(40, 40, 10)])), T(f([.2, .9, .5])), 0.5)
    assert k.tolist() == [1, 2, 0]
    # identical boxes: IoU = 1 > thr -> only the best survives; equal scores -> lowest index first
    k = onms.nms(T(f([sq(0, 0, 10)] * 3)), T(f([.5, .5, .5])), 0.5)
    assert k.tolist() == [0]
    # IoU exactly at threshold is kept (strict >): [0,0,2,1] vs [1,0,3,1]: inter 1, union 3 -> 1/3; use 0.5: [0,0,2,1] vs [0,0,1,1]
    k = onms.nms(T(f([[0, 0, 2, 1], [0, 0, 1, 1]])), T(f([.9, .8])), 0.5)
    assert k.tolist() == [0, 1]
    k = onms.nms(T(f([[0, 0, 2, 1], [0, 0, 1, 1]])), T(f([.9, .8])), 0.4999)
    assert k.tolist() == [0]
    # the final comparison is made in double: IoU == float32(0.6) > 0.6 (double) suppresses
    thr_box = [[0, 0, 5, 1], [0, 0, 3, 1]]                      # IoU = 3/5 -> float32(0.6) = 0.60000002...
    k = onms.nms(T(f(thr_box)), T(f([.9, .8])), 0.6)
    assert k.tolist() == [0]
    # zero-area boxes: 0/0 = NaN, NaN > thr is false -> kept
    k = onms.nms(T(f([[1, 1, 1, 1], [1, 1, 1, 1]])), T(f([.9, .8])), 0.1)
    assert k.tolist() == [0, 1]
    # empty input
    k = onms.nms(torch.zeros((0, 4)), torch.zeros((0,)), 0.5)
    assert k.dtype == torch.int64 and k.numel() == 0
    # chain: A suppresses B, B would have suppressed C but is gone -> C kept
    k = onms.nms(T(f([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]])), T(f([.9, .8, .7])), 0.4)
    assert k.tolist() == [0, 2]
    # C and numpy twins agree on random clusters
    rs = np.random.RandomState(5)
    b = rs.uniform(0, 50, (200, 2)).astype(f)
    b = np.concatenate([b, b + rs.uniform(5, 30, (200, 2)).astype(f)], 1)
    s = rs.uniform(0, 1, 200).astype(f)
    s[::7] = s[3]
    for thr in (0.0, 0.3, 0.6, 1.0):
        assert onms.nms(T(b), T(s), thr).tolist() == onms.nms_py(b, s, thr).tolist()


def test_centernet_targets(golden):
    g = golden("centernet")
    inp, outp = tuple(g["targets_input_shape"]), tuple(g["targets_output_shape"])
    for k in range(3):
        maps = oc.draw_targets(g[f"targets{k}_labels"], 3, inp, outp)
        for nm, arr in zip(("cls", "size", "offset", "mask"), maps):
            np.testing.assert_array_equal(arr, g[f"targets{k}_{nm}"])


@pytest.mark.parametrize("tag", ["pos", "zero", "dense"])
def test_centernet_box_loss(golden, tag):
    g = golden("centernet")
    ps, po = T(g["loss_pred_size"]).requires_grad_(True), T(g["loss_pred_offset"]).requires_grad_(True)
    val = oc.box_loss([T(g["loss_pred_cls"]), ps, po],
                      [T(g["loss_lab_cls"]), T(g["loss_lab_size"]), T(g["loss_lab_offset"]), T(g[f"loss_{tag}_mask"])])
    gs, go = torch.autograd.grad(val, [ps, po])
    close(val.detach(), g[f"loss_{tag}_value"])
    close(gs, g[f"loss_{tag}_grad_size"], atol=1e-9)
    close(go, g[f"loss_{tag}_grad_offset"], atol=1e-9)
    if tag == "zero":
        assert float(val) == 1.0


def test_centernet_direct_loss_and_coords(golden):
    g = golden("centernet")
    close(oc.offset_to_coord(T(g["loss_pred_offset"])), g["offset_to_coord"], 0, 0)
    p = T(g["direct_pred"]).requires_grad_(True)
    val = ob.iou_loss_with_mask(p, T(g["direct_target"]), T(g["direct_mask"]))
    grad, = torch.autograd.grad(val, p)
    close(val.detach(), g["direct_value"])
    close(grad, g["direct_grad"], atol=1e-9)


def test_centernet_decode_and_postprocess(golden):
    g = golden("centernet")
    heat, size, off = T(g["dec_heat"]), T(g["dec_size"]), T(g["dec_offset"])
    np.testing.assert_array_equal(oc.pool_nms(heat).numpy(), g["pool_nms"])
    for conf in (0.9, 0.3):
        dets = oc.decode_bbox(heat, size, off, conf)
        for k, d in enumerate(dets):
            want = g[f"decode_c{conf}_img{k}"]
            assert (len(d) == 0) == (want.shape[0] == 0)
            if len(d):
                close(d, want)
                np.testing.assert_array_equal(d[:, 5].numpy(), want[:, 5])
        for need in (False, True):
            post = oc.postprocess([d.clone() if len(d) else d for d in dets], need, 640, 160.0, False, 0.4)
            for k, d in enumerate(post):
                want = g[f"post_c{conf}_nms{int(need)}_img{k}"]
                assert (d is None) == (want.shape[0] == 0)
                if d is not None:
                    close(d, want)
        post = oc.postprocess([d.clone() if len(d) else d for d in dets], True, 500, 640.0, True, 0.4)
        for k, d in enumerate(post):
            want = g[f"post_c{conf}_letterbox_img{k}"]
            assert (d is None) == (want.shape[0] == 0)
            if d is not None:
                close(d, want)
    post = oc.postprocess_predictions([heat, size, off], 640)
    for k, d in enumerate(post):
        want = g[f"postpred_img{k}"]
        assert (d is None) == (want.shape[0] == 0)
        if d is not None:
            close(d, want)


EVAL_CASES = ["empty_det", "empty_lab", "one", "small", "crowd", "many", "loose", "ties"]


@pytest.mark.parametrize("tag", EVAL_CASES)
def test_evalmatch_oracle_vs_reference(golden, tag):
    """val.py:71-92 and metrics.py:116-171: the fixture holds the reference's own outputs (the `ties` case, which has
    equal IoUs, holds the stable refinement of its unstable sorts)."""
    from oracle import evalmatch as oe
    g = golden("evalmatch")
    det, lab = torch.from_numpy(g[f"{tag}_det"]), torch.from_numpy(g[f"{tag}_labels"])
    correct = oe.process_batch(det, lab, torch.from_numpy(g["iouv"]))
    np.testing.assert_array_equal(correct.numpy(), g[f"{tag}_correct"])
    cm = oe.ConfusionMatrix(nc=int(g[f"{tag}_meta"][0]))
    cm.process_batch(det, lab)
    cm.process_batch(det[::2], lab)
    np.testing.assert_array_equal(cm.matrix, g[f"{tag}_matrix"])


def test_tta_descale_and_clip_oracle_vs_reference(golden):
    """Model._descale_pred (in-place branch) and _clip_augmented, models/yolo.py:155-183."""
    from oracle import yolo as oy
    g = golden("tta")
    base = g["base"]
    for k in range(4):
        flips, scale = g[f"descale{k}_args"]
        got = oy.descale_pred(torch.from_numpy(base.copy()), int(flips) or None, float(scale), (480, 640))
        np.testing.assert_array_equal(got.numpy(), g[f"descale{k}"])
    ys = oy.clip_augmented([torch.from_numpy(base), torch.from_numpy(base[:, :900]), torch.from_numpy(base[:, :400])], 3)
    np.testing.assert_array_equal([y.shape[1] for y in ys], g["clip_shapes"])
    np.testing.assert_array_equal(np.stack([ys[0][0, -1].numpy(), ys[-1][0, 0].numpy()]), g["clip_first_last"])


AP_CASES = ("empty", "one", "small", "single_iou", "all_fp", "all_tp", "absent", "coco_like", "one_class_long")
AP_OUT = ("tp", "fp", "p", "r", "f1", "ap", "classes")


@pytest.mark.parametrize("tag", AP_CASES)
def test_ap_per_class_oracle_vs_reference(golden, tag):
    """metrics.py:21-114: every output of the reference's ap_per_class, bit for bit (float64)."""
    import warnings
    from oracle import ap as oa
    g = golden("ap")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = oa.ap_per_class(g[f"{tag}_in_tp"], g[f"{tag}_in_conf"], g[f"{tag}_in_pred_cls"], g[f"{tag}_in_target_cls"])
    for nm, v in zip(AP_OUT, got):
        want = g[f"{tag}_{nm}"]
        assert v.dtype == want.dtype and v.shape == want.shape, nm
        np.testing.assert_array_equal(v, want, err_msg=nm)
    if tag == "small":
        np.testing.assert_array_equal(oa.fitness(g["fitness_in"]), g["fitness_out"])


@pytest.mark.parametrize("tag", ["default", "poor", "tight", "one"])
def test_autoanchor_metric_oracle_vs_reference(golden, tag):
    """autoanchor.py:35-41, 86-94: the closures' outputs, bit for bit."""
    from oracle import autoanchor as oaa
    g = golden("autoanchor")
    k, wh, thr = torch.from_numpy(g[f"{tag}_k"]), torch.from_numpy(g[f"{tag}_wh"]), float(g[f"{tag}_thr"])
    x, best = oaa.metric(k, wh)
    np.testing.assert_array_equal(x.numpy(), g[f"{tag}_x"])
    np.testing.assert_array_equal(best.numpy(), g[f"{tag}_best"])
    bpr, aat = oaa.check_metric(k, wh, thr)
    fit = oaa.anchor_fitness(g[f"{tag}_k"], wh, 1 / thr)
    np.testing.assert_array_equal(np.array([bpr.item(), aat.item(), fit.item()], dtype=np.float32), g[f"{tag}_stats"])


FOCAL_CASES = ("train", "g0", "g2", "flat", "idx", "idx_ignore", "idx_g0")


def focal_kwargs(kw):
    nc, one_hot, gamma, smoothing, ignore = kw
    return dict(num_classes=int(nc), one_hot_label_format=bool(one_hot), gamma=float(gamma), smoothing=float(smoothing),
                ignore_index=None if ignore < 0 else int(ignore))


@pytest.mark.parametrize("tag", FOCAL_CASES)
def test_focal_loss_oracle_vs_reference(golden, tag):
    """loss.py:8-83: value and gradient of the reference's LabelSmoothingFocalLoss."""
    from oracle import focal as of
    g = golden("focal")
    x = torch.from_numpy(g[f"{tag}_logits"]).requires_grad_(True)
    loss = of.LabelSmoothingFocalLoss(**focal_kwargs(g[f"{tag}_kw"]))(x, torch.from_numpy(g[f"{tag}_label"]))
    grad, = torch.autograd.grad(loss * 3.0, x)
    # (the fixture was made with one torch thread: with more, the reductions inside softmax regroup by an ulp)
    np.testing.assert_allclose(loss.detach().numpy(), g[f"{tag}_loss"], rtol=1e-6)
    np.testing.assert_allclose(grad.numpy(), g[f"{tag}_grad"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("tag", ["reg_l1", "reg_smooth"])
def test_regression_loss_oracle_vs_reference(golden, tag):
    """loss.py:86-100: RegressionLossWithMask value and gradient."""
    from oracle import focal as of
    g = golden("focal")
    x = torch.from_numpy(g[f"{tag}_pred"]).requires_grad_(True)
    loss = of.RegressionLossWithMask(tag == "reg_smooth")(x, torch.from_numpy(g[f"{tag}_target"]), torch.from_numpy(g[f"{tag}_mask"]))
    grad, = torch.autograd.grad(loss * 2.0, x)
    np.testing.assert_allclose(loss.detach().numpy(), g[f"{tag}_loss"], rtol=1e-6)
    np.testing.assert_allclose(grad.numpy(), g[f"{tag}_grad"], rtol=1e-6, atol=1e-9)


DETECT_TAGS = ["s3_nc80", "s3_nc80_cat", "s3_nc7_rect", "s3_nc1", "s4_nc7", "s5_nc3_na2"]


@pytest.mark.parametrize("tag", DETECT_TAGS)
def test_detect_head_maps_and_decode(golden, tag):
    """Detect.forward of the reference's own class (models/yolo.py:49-81, compiled from its file by
    tests/golden/make_golden.py): 3 / 4 / 5 levels, nc 1 / 7 / 80, rectangular grids, both the in-place and the
    concatenating decode.  The oracle repeats the same torch operations in the same order: bit-exact."""
    g = golden("detect")
    nc, na, nl, _, bs = (int(v) for v in g[f"{tag}_meta"])
    convs = [T(g[f"{tag}_conv{l}"]) for l in range(nl)]
    maps = oy.head_maps(convs, na)
    z = oy.detect_decode(maps, T(g[f"{tag}_anchors"]), [float(s) for s in g[f"{tag}_strides"]])
    assert z.shape == g[f"{tag}_z"].shape and z.shape[0] == bs and z.shape[2] == nc + 5
    np.testing.assert_array_equal(z.numpy(), g[f"{tag}_z"])
    for l, m in enumerate(maps):      # anchor_grid = anchors * stride, grid = (x, y) cell indices (yolo.py:72-81)
        assert tuple(g[f"{tag}_grid{l}"].shape) == (1, na, m.shape[2], m.shape[3], 2)
        np.testing.assert_array_equal(g[f"{tag}_anchor_grid{l}"][0, :, 0, 0], g[f"{tag}_anchors"][l] * g[f"{tag}_strides"][l])


def half_nms_kwargs(g, tag):
    from tests._synth import HALF_NMS_VARIANTS
    kw = dict(HALF_NMS_VARIANTS[tag])
    if tag == "labels":
        kw["labels"] = [torch.from_numpy(g["nms_labels0"]), torch.zeros((0, 5)), torch.from_numpy(g["nms_labels2"])]
    return kw


def test_half_precision_detect_and_nms(golden):
    """The reference's half-precision inference path (model.half(), val.py:129-130) run on the CPU by
    tests/golden/make_golden.py: its Detect class on fp16 maps (fp16 result; fp32 for the torch.cat form) and its
    non_max_suppression on an fp16 prediction tensor (fp16 products and corners, fp32 rows).  The oracle's functions are
    the same torch operations in the same order on the same operand types: bit-exact."""
    from tests._synth import HALF_DETECT_TAGS, HALF_NMS_VARIANTS
    g = golden("half")
    for tag in HALF_DETECT_TAGS:
        nc, na, nl, inplace, bs = (int(v) for v in g[f"{tag}_meta"])
        convs = [torch.from_numpy(g[f"{tag}_conv{l}"]) for l in range(nl)]
        assert all(c.dtype == torch.float16 for c in convs)
        z = oy.detect_decode(oy.head_maps(convs, na), torch.from_numpy(g[f"{tag}_anchors"]), [float(s) for s in g[f"{tag}_strides"]],
                             inplace=bool(inplace))
        assert z.dtype == (torch.float16 if inplace else torch.float32)
        np.testing.assert_array_equal(z.numpy(), g[f"{tag}_z"])
    pred = torch.from_numpy(g["nms_pred"])
    assert pred.dtype == torch.float16
    for tag in HALF_NMS_VARIANTS:
        got = onms.non_max_suppression(pred.clone(), **half_nms_kwargs(g, tag))
        want = unpack(g, "nms_" + tag)
        assert [int(d.shape[0]) for d in got] == [w.shape[0] for w in want], tag
        for d, w in zip(got, want):
            assert d.dtype == torch.float32
            np.testing.assert_array_equal(d.numpy(), w)
