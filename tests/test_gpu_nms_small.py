"""The one-workgroup-per-image NMS path (k_select = K_SMALL) against the oracle: forced on awkward inputs, through the
automatic choice (second call of the same kind), and its hand-over to the general path when an image has too many
candidates."""
import numpy as np
import pytest
import torch

from oracle import nms as onms
from oracle import yolo as oy
from tests._synth import synth_head
from tests.test_gpu_fuzz import awkward_pred
from tests.test_gpu_nms import same_lists

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def sparse_pred(rs, bs, A, nc, n_obj, k_per_obj, dense=False):
    """(bs, A, 5+nc) predictions like a trained detector's: background objectness below 0.1, a few clusters of
    overlapping confident boxes per image; dense=True makes every row a confident candidate."""
    pred = np.zeros((bs, A, 5 + nc), dtype=np.float32)
    pred[..., 0:2] = rs.uniform(0, 640, (bs, A, 2))
    pred[..., 2:4] = np.exp(rs.uniform(np.log(8), np.log(200), (bs, A, 2)))
    pred[..., 4] = rs.uniform(0.3, 0.9, (bs, A)) if dense else rs.uniform(0, 0.1, (bs, A))
    pred[..., 5:] = rs.uniform(0, 0.2, (bs, A, nc))
    if dense:
        pred[..., 5] = rs.uniform(0.8, 1.0, (bs, A))
    for b in range(bs):
        for _ in range(n_obj):
            rows = rs.choice(A, k_per_obj, replace=False)
            c, wh, cls = rs.uniform(50, 590, 2), rs.uniform(30, 150, 2), rs.randint(nc)
            pred[b, rows, 0:2] = c + rs.normal(0, 3, (k_per_obj, 2))
            pred[b, rows, 2:4] = wh * np.exp(rs.normal(0, 0.05, (k_per_obj, 2)))
            pred[b, rows, 4] = rs.uniform(0.5, 0.95, k_per_obj)
            pred[b, rows, 5 + cls] = rs.uniform(0.6, 0.99, k_per_obj)
    return pred.astype(np.float32)


@pytest.mark.parametrize("seed", range(40))
def test_small_path_forced_fuzz(seed):
    from object_detection_amd import general as G
    rs = np.random.RandomState(7000 + seed)                       # the inputs of test_nms_fuzz
    pred, nc = awkward_pred(rs)
    kw = dict(conf_thres=float(rs.choice([0.001, 0.05, 0.25, 0.5])), iou_thres=float(rs.choice([0.0, 0.3, 0.45, 0.6, 1.0])),
              agnostic=bool(rs.rand() < 0.3), multi_label=bool(rs.rand() < 0.5), max_det=int(rs.choice([1, 5, 300, 1024])))
    if rs.rand() < 0.3:
        kw["classes"] = sorted(set(int(c) for c in rs.randint(0, nc, 3)))
    want = onms.non_max_suppression(T(pred).clone(), **kw)
    rows, meta, cap = G.nms_launch(T(pred).cuda(), k_select=G.K_SMALL, **kw)
    host = meta.tolist()
    bs = pred.shape[0]
    got, need = G.nms_views(rows, host, cap)
    if got is None:
        assert need > cap or max(host[2 * bs:3 * bs]) > G.SMALL_MAX        # only too many candidates hand over
        got = G.non_max_suppression(T(pred).cuda(), **kw)
    same_lists(got, want)


def test_small_path_is_chosen_on_the_second_call_and_agrees():
    from object_detection_amd import general as G
    rs = np.random.RandomState(5)
    pred = sparse_pred(rs, 8, 25200, 80, 6, 12)
    kw = dict(conf_thres=0.25, iou_thres=0.45)
    want = onms.non_max_suppression(T(pred).clone(), **kw)
    d = T(pred).cuda()
    G._SMALL_HINT.clear()
    rows, _, _ = G.nms_launch(d, **kw)
    assert rows._dh_attempt[1] == G.K_CHAIN                       # nothing known yet: the general first attempt
    same_lists(G.non_max_suppression(d, **kw), want)
    rows, _, _ = G.nms_launch(d, **kw)
    assert rows._dh_attempt[1] == G.K_SMALL                       # a few dozen candidates per image last time
    same_lists(G.non_max_suppression(d, **kw), want)
    # the same tensor at a validation threshold is another kind of call, and far too dense for the small path
    kw2 = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)
    want2 = onms.non_max_suppression(T(pred).clone(), **kw2)
    same_lists(G.non_max_suppression(d, **kw2), want2)
    same_lists(G.non_max_suppression(d, **kw2), want2)
    rows, _, _ = G.nms_launch(d, **kw2)
    assert rows._dh_attempt[1] == G.K_CHAIN


def test_small_path_wrong_guess_falls_back():
    """The hint says small, the data has become dense: the call still returns the right answer."""
    from object_detection_amd import general as G
    rs = np.random.RandomState(6)
    sparse = sparse_pred(rs, 2, 5000, 6, 3, 5)
    dense = sparse_pred(rs, 2, 5000, 6, 3, 5, dense=True)         # every row a candidate
    kw = dict(conf_thres=0.2, iou_thres=0.5)
    G._SMALL_HINT.clear()
    same_lists(G.non_max_suppression(T(sparse).cuda(), **kw), onms.non_max_suppression(T(sparse).clone(), **kw))
    rows, _, _ = G.nms_launch(T(dense).cuda(), **kw)
    assert rows._dh_attempt[1] == G.K_SMALL
    same_lists(G.non_max_suppression(T(dense).cuda(), **kw), onms.non_max_suppression(T(dense).clone(), **kw))
    rows, _, _ = G.nms_launch(T(dense).cuda(), **kw)
    assert rows._dh_attempt[1] == G.K_CHAIN                       # and the guess has been corrected


def test_small_path_fused_raw_and_half():
    from object_detection_amd import general as G
    rs = np.random.RandomState(9)
    anchors = oy.grid_anchors().cuda()
    strides = torch.tensor(oy.YOLOV5S_STRIDES)
    levels = [T(x).cuda() for x in synth_head(rs, 4, 320, 80)]
    for lv in (levels, [x.half() for x in levels]):
        from object_detection_amd.yolo import detect_decode
        dec = detect_decode(lv, anchors, strides)
        for kw in (dict(conf_thres=0.25, iou_thres=0.45), dict(conf_thres=0.1, iou_thres=0.5, agnostic=True)):
            want = onms.non_max_suppression(dec.cpu().clone(), **kw)
            rows, meta, cap = G.detect_nms_launch(lv, anchors, strides, k_select=G.K_SMALL, **kw)
            got, _ = G.nms_finish(rows, meta, cap)
            if got is None:
                got = G.detect_nms(lv, anchors, strides, **kw)
            same_lists(got, want)
