"""The validation-threshold NMS path that runs everything after the candidate search in one workgroup per image
(k_select = K_CHAIN): forced on awkward inputs, on class lists of every length (one wave, several chunks of a wave, the
whole workgroup), on what it hands over to the separate launches (coordinate range across classes, too many equal scores,
more classes than its tables hold), from raw maps and convolution outputs, all against the oracle or the separate launches."""
import numpy as np
import pytest
import torch

from oracle import nms as onms
from oracle import yolo as oy
from tests._synth import synth_head, synth_pred
from tests.test_gpu_fuzz import awkward_pred
from tests.test_gpu_nms import same_lists

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def forced(pred_dev, k, **kw):
    """One launch with the given first attempt: (rows per image or None, need, needs_full flags)."""
    from object_detection_amd import general as G
    rows, meta, cap = G.nms_launch(pred_dev, k_select=k, **kw)
    host = meta.tolist()
    bs = pred_dev.shape[0]
    got, need = G.nms_views(rows, host, cap)
    return got, need, host[3 * bs:]


@pytest.mark.parametrize("seed", range(40))
def test_chain_forced_fuzz(seed):
    from object_detection_amd import general as G
    rs = np.random.RandomState(7000 + seed)                       # the inputs of test_nms_fuzz
    pred, nc = awkward_pred(rs)
    kw = dict(conf_thres=float(rs.choice([0.001, 0.05, 0.25, 0.5])), iou_thres=float(rs.choice([0.0, 0.3, 0.45, 0.6, 1.0])),
              agnostic=bool(rs.rand() < 0.3), multi_label=bool(rs.rand() < 0.5), max_det=int(rs.choice([1, 5, 300, 480])))
    if rs.rand() < 0.3:
        kw["classes"] = sorted(set(int(c) for c in rs.randint(0, nc, 3)))
    want = onms.non_max_suppression(T(pred).clone(), **kw)
    got, need, flags = forced(T(pred).cuda(), G.K_CHAIN, **kw)
    if got is None:                                               # capacity, an insufficient prefix, or an image left to the separate launches
        got = G.non_max_suppression(T(pred).cuda(), **kw)
    same_lists(got, want)


@pytest.mark.parametrize("nc,shape", [(80, "spread"), (3, "few classes"), (1, "one class"), (20, "one dominant class"), (512, "many classes")])
def test_chain_list_lengths(nc, shape):
    """~3000 candidates per image spread over the classes in different ways: lists of a few entries (one wave, one chunk),
    of 65-192 (one wave, several chunks), and of hundreds to thousands (the whole workgroup)."""
    from object_detection_amd import general as G
    rs = np.random.RandomState(nc)
    bs, A = 3, 6000
    pred = np.zeros((bs, A, 5 + nc), dtype=np.float32)
    pred[..., 0:2] = rs.uniform(0, 640, (bs, A, 2))
    pred[..., 2:4] = np.exp(rs.uniform(np.log(6), np.log(120), (bs, A, 2)))
    pred[..., 4] = rs.uniform(0.2, 1, (bs, A))
    pred[..., 5:] = rs.uniform(0, 0.1, (bs, A, nc))
    cls = rs.randint(0, nc, (bs, A))
    if shape == "one dominant class":
        cls = np.where(rs.rand(bs, A) < 0.7, 7, cls)
    np.put_along_axis(pred[..., 5:], cls[..., None], rs.uniform(0.5, 1, (bs, A, 1)).astype(np.float32), -1)
    for kw in (dict(conf_thres=0.25, iou_thres=0.45, max_det=300), dict(conf_thres=0.3, iou_thres=0.6, max_det=480),
               dict(conf_thres=0.05, iou_thres=0.5, multi_label=True, max_det=100)):
        want = onms.non_max_suppression(T(pred).clone(), **kw)
        got, need, flags = forced(T(pred).cuda(), G.K_CHAIN, **kw)
        if got is None:
            assert need > 0 or 1 in flags, (need, flags)           # more candidates than rows, or the prefix did not yield max_det
            got = G.non_max_suppression(T(pred).cuda(), **kw)
        else:
            assert not any(flags)
        same_lists(got, want)


def test_chain_hands_over_what_it_cannot_do():
    from object_detection_amd import general as G
    rs = np.random.RandomState(3)
    pred = synth_pred(rs, 3, 5000, 6, n_obj=10, k_per_obj=30)
    kw = dict(conf_thres=0.05, iou_thres=0.45)
    # image 1: boxes wider than the class offset, so classes overlap after the shift (general.py:720-721)
    wide = pred.copy()
    wide[1, :60, 0:2] = rs.uniform(0, 9000, (60, 2))
    wide[1, :60, 2:4] = rs.uniform(6000, 20000, (60, 2))
    wide[1, :60, 4] = rs.uniform(0.5, 1, 60)
    wide[1, :60, 5:] = rs.uniform(0.3, 1, (60, 6))
    got, need, flags = forced(T(wide).cuda(), G.K_CHAIN, **kw)
    assert got is None and need == -2 and flags == [0, 2, 0]
    same_lists(G.non_max_suppression(T(wide).cuda(), **kw), onms.non_max_suppression(T(wide).clone(), **kw))
    # image 2: 4500 rows with one and the same score: the selected prefix cannot be cut inside them
    ties = pred.copy()
    ties[2, :4500, 4] = 0.75
    ties[2, :4500, 5:] = 0.0
    ties[2, :4500, 5] = 0.5
    got, need, flags = forced(T(ties).cuda(), G.K_CHAIN, **kw)
    assert got is None and need == -2 and flags == [0, 0, 2]
    same_lists(G.non_max_suppression(T(ties).cuda(), **kw), onms.non_max_suppression(T(ties).clone(), **kw))
    # agnostic, more classes than the tables hold, a large max_det: the call runs as the separate launches, same answer
    for pr, kw2 in ((pred, dict(conf_thres=0.05, agnostic=True)), (pred, dict(conf_thres=0.05, max_det=1000)),
                    (synth_pred(rs, 2, 3000, 600, n_obj=8, k_per_obj=20), dict(conf_thres=0.05, multi_label=True))):
        got, need, flags = forced(T(pr).cuda(), G.K_CHAIN, **kw2)
        want = onms.non_max_suppression(T(pr).clone(), **kw2)
        if got is None:
            got = G.non_max_suppression(T(pr).cuda(), **kw2)
        same_lists(got, want)


@pytest.mark.parametrize("half", [False, True])
def test_chain_from_raw_maps_and_convolution_outputs(half):
    """detect_nms / conv_nms take the same path (the boxes of the selected keys are decoded from the raw values)."""
    from object_detection_amd import general as G
    from object_detection_amd.yolo import detect_decode
    rs = np.random.RandomState(11)
    bs, nc = 4, 80
    levels = synth_head(rs, bs, 320, nc)
    for x in levels:
        hot = rs.rand(*x.shape[:4]) < 0.05
        x[..., 4] += hot * 7.0
        c = rs.randint(0, nc, x.shape[:4])
        np.put_along_axis(x[..., 5:], c[..., None], np.take_along_axis(x[..., 5:], c[..., None], -1) + hot[..., None] * 8.0, -1)
        x[..., 2:4] *= 0.3
    dev = [T(x).cuda().half() if half else T(x).cuda() for x in levels]
    anchors, strides = oy.grid_anchors().cuda(), torch.tensor(oy.YOLOV5S_STRIDES)
    convs = [x.permute(0, 1, 4, 2, 3).reshape(bs, 3 * (nc + 5), x.shape[2], x.shape[3]).contiguous() for x in dev]
    z = detect_decode(dev, anchors, strides)
    def attempts(launch, first_k):
        """The public wrappers' loop (capacity growth, next attempt) started at a given first attempt; also says which
        attempts were made."""
        cap, k, used = None, first_k, []
        while True:
            rows, meta, cap = launch(cap, k)
            used.append(k)
            got, need = G.nms_finish(rows, meta, cap)
            if got is not None:
                return got, used
            if need < 0:
                k = G._next_attempt(rows, need)
            else:
                cap = need

    for kw in (dict(conf_thres=0.001, iou_thres=0.6, multi_label=True), dict(conf_thres=0.1, iou_thres=0.45)):
        want, _ = attempts(lambda cap, k: G.nms_launch(z, cap=cap, k_select=k, **kw), G.K_SELECT)      # the separate launches
        assert sum(int(w.shape[0]) for w in want) > 100
        for src, extra in ((z, None), (dev, {}), (convs, dict(conv_na=3))):
            if extra is None:
                got, used = attempts(lambda cap, k: G.nms_launch(src, cap=cap, k_select=k, **kw), G.K_CHAIN)
            else:
                got, used = attempts(lambda cap, k: G.detect_nms_launch(src, anchors, strides, cap=cap, k_select=k, **kw, **extra), G.K_CHAIN)
            assert set(used) == {G.K_CHAIN}, used                  # nothing was handed over
            for g, w in zip(got, want):
                assert torch.equal(g, w)
