"""GPU parity: batched non_max_suppression, plain nms and the box-format helpers.

Kept rows, their order and every float in them must be bit-identical to the oracle / the reference
fixtures: the whole stage consists of IEEE add/sub/mul/div and comparisons only.
"""
import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import nms as onms
from tests._synth import make_big_pred, synth_pred
from tests.test_oracle_golden import MERGE_VARIANTS, NMS_VARIANTS, merge_close, nms_labels, unpack

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def same_lists(got, want):
    assert [int(d.shape[0]) for d in got] == [int(w.shape[0]) for w in want]
    for d, w in zip(got, want):
        w = w.cpu().numpy() if isinstance(w, torch.Tensor) else w
        np.testing.assert_array_equal(d.cpu().numpy(), w)


@pytest.mark.parametrize("tag", list(NMS_VARIANTS))
def test_non_max_suppression_golden(golden, tag):
    from object_detection_amd.general import non_max_suppression
    g = golden("nms")
    kw = dict(NMS_VARIANTS[tag])
    if tag == "labels":
        kw["labels"] = [l.cuda() for l in nms_labels(g)]
    pred = T(g["pred"]).cuda()
    before = pred.clone()
    got = non_max_suppression(pred, **kw)
    same_lists(got, unpack(g, tag))
    assert torch.equal(pred, before)          # the input is not modified (the reference works on a copy)


@pytest.mark.parametrize("tag", list(MERGE_VARIANTS))
def test_merge_nms_golden(golden, tag):
    """merge-NMS (general.py:725-731) against the reference run with its local switch set."""
    from object_detection_amd.general import non_max_suppression
    g = golden("nms_merge")
    pred = T(g["pred"]).cuda()
    merge_close(non_max_suppression(pred, merge=True, **MERGE_VARIANTS[tag]), unpack(g, tag))
    # the default leaves the step out, as the reference does
    same_lists(non_max_suppression(pred, **MERGE_VARIANTS[tag]), onms.non_max_suppression(T(g["pred"]).clone(), **MERGE_VARIANTS[tag]))


def test_merge_nms_wide_and_fused(golden):
    from object_detection_amd.general import detect_nms, non_max_suppression
    from object_detection_amd.yolo import detect_decode
    from oracle import yolo as oy
    from tests._synth import synth_head
    g = golden("nms_merge")
    got = non_max_suppression(T(g["wide_pred"]).cuda(), conf_thres=0.0, iou_thres=0.5, merge=True)
    merge_close(got, unpack(g, "wide"))            # image 0 has >= 3000 candidates and is left as NMS made it
    # fused path: same rows as the two calls, and as the oracle on the decoded tensor
    rs = np.random.RandomState(9)
    levels = [torch.from_numpy(x).cuda() for x in synth_head(rs, 2, 128, 6)]
    for x in levels:
        x[..., 4] += 5.0                           # plenty of candidates per image
    anchors, stride = oy.grid_anchors().cuda(), torch.tensor(oy.YOLOV5S_STRIDES)
    dec = detect_decode(levels, anchors, stride)
    two = non_max_suppression(dec, conf_thres=0.3, iou_thres=0.3, merge=True)
    one = detect_nms(levels, anchors, stride, conf_thres=0.3, iou_thres=0.3, merge=True)
    same_lists(one, two)
    merge_close(two, [w.numpy() for w in onms.non_max_suppression(dec.cpu(), conf_thres=0.3, iou_thres=0.3, merge=True)])
    assert sum(int(d.shape[0]) for d in two) > 0


def test_nms_single_class_ties_and_big(golden):
    from object_detection_amd.general import non_max_suppression
    g = golden("nms")
    same_lists(non_max_suppression(T(g["nc1_pred"]).cuda(), conf_thres=0.1, multi_label=True), unpack(g, "nc1"))
    for key, kw in (("tie", {}), ("tie_multi", {"multi_label": True})):
        same_lists(non_max_suppression(T(g["tie_pred"]).cuda(), conf_thres=0.05, iou_thres=0.5, **kw), unpack(g, key))
    big = make_big_pred()
    got = non_max_suppression(T(big).cuda(), conf_thres=0.01, iou_thres=0.5, multi_label=True, max_det=300)
    same_lists(got, unpack(g, "big"))


def test_nms_asserts_and_empty():
    from object_detection_amd.general import non_max_suppression
    pred = torch.zeros((2, 10, 9), device="cuda")
    with pytest.raises(AssertionError):
        non_max_suppression(pred, conf_thres=1.5)
    with pytest.raises(AssertionError):
        non_max_suppression(pred, iou_thres=-0.1)
    out = non_max_suppression(pred)
    assert len(out) == 2 and all(o.shape == (0, 6) for o in out)


@pytest.mark.parametrize("kw", [dict(conf_thres=0.25, iou_thres=0.45), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
                                dict(conf_thres=0.05, iou_thres=0.5, agnostic=True, max_det=1000),
                                dict(conf_thres=0.02, iou_thres=0.3, multi_label=True, classes=[1, 3, 5, 79], max_det=50)])
def test_nms_bs8_25200_vs_oracle(kw):
    """Decoded-prediction shaped input of the 640 px head: bs 8 x 25200 anchors x 80 classes."""
    from object_detection_amd.general import non_max_suppression
    rs = np.random.RandomState(101)
    pred = synth_pred(rs, 8, 25200, 80, n_obj=12, k_per_obj=40)      # score ties allowed: oracle and HIP pin the same stable order
    want = onms.non_max_suppression(T(pred).clone(), **kw)
    got = non_max_suppression(T(pred).cuda(), **kw)
    same_lists(got, want)
    assert sum(w.shape[0] for w in want) > 50


def test_nms_dense_stress_vs_oracle():
    """> 30000 candidates per image: top-30000 cut, long class segments, kept lists beyond the LDS part."""
    from object_detection_amd.general import non_max_suppression
    rs = np.random.RandomState(77)
    bs, A, nc = 2, 60000, 4
    pred = np.zeros((bs, A, 5 + nc), dtype=np.float32)
    pred[..., 0:2] = rs.uniform(0, 640, (bs, A, 2))
    pred[..., 2:4] = np.exp(rs.uniform(np.log(4), np.log(64), (bs, A, 2)))
    pred[..., 4] = rs.uniform(0.3, 1, (bs, A))
    logits = rs.normal(0, 3, (bs, A, nc))
    pred[..., 5:] = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
    for kw in (dict(conf_thres=0.25, iou_thres=0.45, max_det=300), dict(conf_thres=0.25, iou_thres=0.45, max_det=30000),
               dict(conf_thres=0.25, iou_thres=0.6, agnostic=True, max_det=2000)):
        want = onms.non_max_suppression(T(pred).clone(), **kw)
        got = non_max_suppression(T(pred).cuda(), **kw)
        same_lists(got, want)


@pytest.mark.parametrize("A,nc", [(15, 3), (16, 80), (17, 80), (63, 7), (64, 1), (65, 80), (1000, 80), (1023, 2), (4097, 80),
                                  (4112, 91), (777, 92), (2048, 11)])
def test_nms_dense_single_label_shapes_vs_oracle(A, nc):
    """Single-label candidate search when most rows are candidates, over row counts and widths that take every form of
    the tile path: 16-row tiles as 16-byte vectors (A * no % 4 == 0), the scalar tiles (otherwise, and for the last
    rows of an image when A % 16 != 0), rows too wide for a tile (no > 96: candidate rows four at a time), and a prediction
    tensor that does not start on a 16-byte boundary."""
    from object_detection_amd.general import non_max_suppression
    rs = np.random.RandomState(1000 + A + nc)
    bs = 3
    pred = np.zeros((bs, A, 5 + nc), dtype=np.float32)
    pred[..., 0:2] = rs.uniform(0, 640, (bs, A, 2))
    pred[..., 2:4] = np.exp(rs.uniform(np.log(4), np.log(64), (bs, A, 2)))
    pred[..., 4] = rs.uniform(0.2, 1, (bs, A))                      # ~7 of 8 rows pass conf 0.3
    logits = rs.normal(0, 3, (bs, A, nc))
    pred[..., 5:] = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
    pred[1, : A // 2, 4] = 0.0                                      # an image whose first half has no candidate at all
    kw = dict(conf_thres=0.3, iou_thres=0.5, max_det=1000)
    want = onms.non_max_suppression(T(pred).clone(), **kw)
    same_lists(non_max_suppression(T(pred).cuda(), **kw), want)
    shifted = torch.empty(pred.size + 1, device="cuda")[1:].view(pred.shape)      # 4 bytes past a 16-byte boundary
    shifted.copy_(T(pred))
    assert shifted.data_ptr() % 16 == 4 and shifted.is_contiguous()
    same_lists(non_max_suppression(shifted, **kw), want)
    assert sum(w.shape[0] for w in want) > 0


def test_nms_huge_boxes_cross_class_overlap():
    """Boxes wider than the 7680 px class offset overlap across classes in the reference; so must we."""
    from object_detection_amd.general import non_max_suppression
    rs = np.random.RandomState(8)
    pred = synth_pred(rs, 2, 500, 6, n_obj=4, k_per_obj=10)
    pred[0, :40, 0:2] = rs.uniform(0, 9000, (40, 2))
    pred[0, :40, 2:4] = rs.uniform(6000, 20000, (40, 2))
    pred[0, :40, 4] = rs.uniform(0.5, 1, 40)
    pred[0, :40, 5:] = rs.uniform(0.3, 1, (40, 6))
    for kw in (dict(conf_thres=0.2), dict(conf_thres=0.1, multi_label=True, iou_thres=0.3)):
        same_lists(non_max_suppression(T(pred).cuda(), **kw), onms.non_max_suppression(T(pred).clone(), **kw))


def test_plain_nms_known_answers_and_random():
    from object_detection_amd.ops import nms
    f = np.float32
    sq = lambda x, y, s: [x, y, x + s, y + s]

    def run(b, s, thr):
        return nms(T(f(b)).cuda(), T(f(s)).cuda(), thr).cpu().tolist()
    assert run([sq(0, 0, 10), sq(20, 20, 10), sq(40, 40, 10)], [.2, .9, .5], 0.5) == [1, 2, 0]
    assert run([sq(0, 0, 10)] * 3, [.5, .5, .5], 0.5) == [0]
    assert run([[0, 0, 2, 1], [0, 0, 1, 1]], [.9, .8], 0.5) == [0, 1]            # IoU == thr is kept (strict >)
    assert run([[0, 0, 2, 1], [0, 0, 1, 1]], [.9, .8], 0.4999) == [0]
    assert run([[0, 0, 5, 1], [0, 0, 3, 1]], [.9, .8], 0.6) == [0]               # float32(0.6) > 0.6 in double
    assert run([[1, 1, 1, 1], [1, 1, 1, 1]], [.9, .8], 0.1) == [0, 1]            # 0/0 -> NaN -> kept
    assert run([[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], [.9, .8, .7], 0.4) == [0, 2]
    k = nms(torch.zeros((0, 4), device="cuda"), torch.zeros((0,), device="cuda"), 0.5)
    assert k.dtype == torch.int64 and k.numel() == 0
    rs = np.random.RandomState(5)
    for n in (1, 63, 64, 65, 200, 5000, 20000):
        b = rs.uniform(0, 300, (n, 2)).astype(f)
        b = np.concatenate([b, b + rs.uniform(5, 60, (n, 2)).astype(f)], 1)
        s = rs.uniform(-1, 1, n).astype(f)
        s[::7] = s[n // 2]                                                       # score ties, negative scores
        for thr in (0.0, 0.3, 0.6, 1.0):
            want = onms.nms(T(b), T(s), thr)
            got = nms(T(b).cuda(), T(s).cuda(), thr)
            assert got.cpu().tolist() == want.tolist(), (n, thr)


def test_box_utils(golden):
    from object_detection_amd import general as G
    g = golden("box_utils")
    eq = lambda a, b: np.testing.assert_array_equal(a.cpu().numpy() if isinstance(a, torch.Tensor) else a, b)
    eq(G.xywh2xyxy(T(g["xywh"]).cuda()), g["xywh2xyxy"])
    eq(G.xyxy2xywh(T(g["xywh2xyxy"]).cuda()), g["xyxy2xywh"])
    eq(G.xywh2xyxy(g["xywh"].copy()), g["xywh2xyxy_np"])
    det = T(g["scale_in"]).cuda()
    out = G.scale_coords((640, 640), det, (720, 1280, 3))
    assert out.data_ptr() == det.data_ptr()                       # in place, returns its argument
    eq(det, g["scale_a"])
    eq(G.scale_coords((384, 640), T(g["scale_in"]).cuda(), (1080, 1920)), g["scale_b"])
    eq(G.scale_coords((640, 640), T(g["scale_in"]).cuda(), (500, 375), ratio_pad=((0.8, 0.8), (10.0, 20.0))), g["scale_c"])
    c = T(g["scale_in"]).cuda()
    G.clip_coords(c, (300, 400))
    eq(c, g["clip_t"])
    cn = g["scale_in"].copy()
    G.clip_coords(cn, (300, 400))
    eq(cn, g["clip_np"])
    # a (n,6) detection view as the callers pass it (val.py: scale_coords(im.shape, predn[:, :4], ...))
    d6 = T(g["scale_in"]).cuda()
    G.scale_coords((640, 640), d6[:, :4], (720, 1280, 3))
    eq(d6, g["scale_a"])


def test_nms_uses_decode_objectness_plane_only_while_valid():
    """detect_decode leaves a compact copy of the objectness column for NMS; results must not depend on it,
    and an in-place edit of the decoded tensor must invalidate it."""
    from object_detection_amd.general import non_max_suppression
    from object_detection_amd.yolo import detect_decode, objectness_plane
    from oracle import yolo as oy
    from tests._synth import synth_head
    rs = np.random.RandomState(17)
    levels = synth_head(rs, 4, 640, 80)
    for x in levels:                                    # make a few thousand confident rows
        m = rs.uniform(0, 1, x.shape[:-1]) < 0.01
        x[m, 4] = rs.normal(1.0, 1.0, int(m.sum()))
        x[m, 5 + rs.randint(0, 80)] = 2.0
    z = detect_decode([T(x).cuda() for x in levels], oy.grid_anchors().cuda(), torch.tensor(oy.YOLOV5S_STRIDES))
    plane = objectness_plane(z)
    assert plane is not None and torch.equal(plane, z[:, :, 4])
    with_plane = non_max_suppression(z, 0.25, 0.45)
    assert objectness_plane(z.clone()) is None
    without = non_max_suppression(z.clone(), 0.25, 0.45)
    same_lists(with_plane, [w.cpu() for w in without])
    assert sum(w.shape[0] for w in without) > 100
    from object_detection_amd.yolo import use_compact_planes
    was = use_compact_planes(False)                     # the switch: the plane is there but nobody is handed it
    try:
        assert objectness_plane(z) is None
        same_lists(non_max_suppression(z, 0.25, 0.45), [w.cpu() for w in without])
    finally:
        use_compact_planes(was)
    assert objectness_plane(z) is not None
    z[:, :, 4] *= 0.5                                   # in-place edit: the side copy is stale now
    assert objectness_plane(z) is None
    same_lists(non_max_suppression(z, 0.25, 0.45), [w.cpu() for w in non_max_suppression(z.clone(), 0.25, 0.45)])


@pytest.mark.parametrize("kw", [dict(conf_thres=0.25, iou_thres=0.45), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
                                dict(conf_thres=0.05, iou_thres=0.5, agnostic=True, max_det=1000),
                                dict(conf_thres=0.02, iou_thres=0.3, multi_label=True, classes=[1, 3, 5, 79], max_det=50)])
def test_fused_decode_nms_equals_two_calls_and_oracle(kw):
    """detect_nms(raw maps) == non_max_suppression(detect_decode(raw maps)) bit for bit, and == the oracle run on the
    HIP-decoded tensor; rectangular levels, batch 5."""
    from object_detection_amd.general import detect_nms, non_max_suppression
    from object_detection_amd.yolo import detect_decode
    from oracle import yolo as oy
    from tests._synth import synth_head, synth_targets
    rs = np.random.RandomState(31)
    bs, nc = 5, 80
    levels = [x[:, :, : x.shape[2] * 3 // 4] for x in synth_head(rs, bs, 320, nc)]       # ny != nx
    levels = [np.ascontiguousarray(x) for x in levels]
    anchors = oy.grid_anchors()
    crit = oy.YoloLossOracle(anchors, dict(oy.HYP_SCRATCH), nc=nc)
    targets = synth_targets(rs, 60, bs)
    targets[:, 3] *= 0.7
    for l, t in enumerate(crit.targets_for([T(x) for x in levels], T(targets))):
        levels[l][t["b"], t["a"], t["gj"], t["gi"], 4] = 1.5 + rs.normal(0, 1, len(t["b"]))
        levels[l][t["b"], t["a"], t["gj"], t["gi"], 5 + t["tcls"]] = 2.0 + rs.normal(0, 1, len(t["b"]))
        levels[l][t["b"], t["a"], t["gj"], t["gi"], 0:4] *= 0.3
    dev = [T(x).cuda() for x in levels]
    strides = torch.tensor(oy.YOLOV5S_STRIDES)
    z = detect_decode(dev, anchors.cuda(), strides)
    two = non_max_suppression(z, **kw)
    fused = detect_nms(dev, anchors.cuda(), strides, **kw)
    same_lists(fused, two)
    same_lists(fused, onms.non_max_suppression(z.cpu().clone(), **kw))
    assert sum(t.shape[0] for t in two) > 20


@pytest.mark.parametrize("nc", [2, 3, 20, 80])
def test_streamed_candidate_search_with_most_pairs_passing(nc):
    """Validation threshold, multi-label, nearly every (row, class) pair a candidate: far more keys than one workgroup
    gathers in LDS before it writes them out; row lengths that start rows at every offset of a 16-byte vector; the decoded
    tensor with and without its objectness plane, the raw maps in fp32 and fp16, and a class filter."""
    from object_detection_amd.general import detect_nms, non_max_suppression
    from object_detection_amd.yolo import detect_decode
    from oracle import yolo as oy
    from tests._synth import synth_head
    rs = np.random.RandomState(400 + nc)
    levels = synth_head(rs, 2, 64, nc)
    for x in levels:
        x[..., 4] = rs.normal(0.5, 1.5, x.shape[:-1])
        x[..., 5:] = rs.normal(1.0, 1.0, x[..., 5:].shape)
    anchors, strides = oy.grid_anchors().cuda(), torch.tensor(oy.YOLOV5S_STRIDES)
    for kw in (dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
               dict(conf_thres=0.03, iou_thres=0.5, multi_label=True, classes=[0, nc - 1], max_det=1000)):
        for dev in ([T(x).cuda() for x in levels], [T(x).cuda().half() for x in levels]):
            z = detect_decode(dev, anchors, strides)
            want = onms.non_max_suppression(z.cpu().clone(), **kw)
            assert sum(w.shape[0] for w in want) > 20
            same_lists(non_max_suppression(z, **kw), want)                  # objectness plane of the decode
            same_lists(non_max_suppression(z.clone(), **kw), want)          # strided objectness column
            same_lists(detect_nms(dev, anchors, strides, **kw), want)


def test_nms_prefix_selection_and_its_fallback():
    """Only the best-scoring few thousand candidates go through NMS on the first attempt.  (a) plenty of distinct
    objects: the prefix yields max_det rows and is final; (b) one tight cluster: it does not, the launch says so, and
    the wrapper repeats with the full sort.  Both must equal the oracle."""
    from object_detection_amd.general import nms_finish, nms_launch, non_max_suppression
    rs = np.random.RandomState(5)
    bs, A, nc = 3, 9000, 3
    pred = np.zeros((bs, A, 5 + nc), dtype=np.float32)
    pred[..., 0:2] = rs.uniform(0, 640, (bs, A, 2))
    pred[..., 2:4] = rs.uniform(4, 12, (bs, A, 2))                     # small boxes: most survive
    pred[..., 4] = rs.uniform(0.5, 1, (bs, A))
    pred[..., 5:] = rs.uniform(0.5, 1, (bs, A, nc))
    pred[1, :, 0:2] = 320 + rs.uniform(-2, 2, (A, 2))                  # image 1: one cluster, a handful survive
    pred[1, :, 2:4] = 100
    kw = dict(conf_thres=0.25, iou_thres=0.45, max_det=300)
    want = onms.non_max_suppression(T(pred).clone(), **kw)
    assert want[0].shape[0] == 300 and want[1].shape[0] < 50
    dev = T(pred).cuda()
    rows, meta, cap = nms_launch(dev, **kw)
    flags = meta.cpu().numpy()[3 * bs:]
    np.testing.assert_array_equal(flags, [0, 1, 0])
    assert nms_finish(rows, meta, cap) == (None, -1)
    rows, meta, cap = nms_launch(dev, k_select=0, **kw)
    assert not meta.cpu().numpy()[3 * bs:].any()
    same_lists(nms_finish(rows, meta, cap)[0], want)
    same_lists(non_max_suppression(dev, **kw), want)
    # multi-label: 3 candidates per row, 27000 per image
    kw = dict(conf_thres=0.25, iou_thres=0.45, max_det=300, multi_label=True)
    same_lists(non_max_suppression(dev, **kw), onms.non_max_suppression(T(pred).clone(), **kw))
