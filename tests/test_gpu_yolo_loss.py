"""GPU parity: YOLOv5 target assignment (bit-exact) and ComputeLoss value / gradients (1e-5 relative)."""
import numpy as np
import pytest
import torch

from oracle import yolo as oy
from tests._checks import all_grads_close, decoded_close, level_rows, loss_grads64
from tests._synth import synth_head, synth_targets
from tests.test_oracle_golden import HYP_KEYS

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class FakeDetect:
    def __init__(self, anchors, nc, stride=(8., 16., 32.)):
        self.anchors = anchors
        self.na, self.nl, self.nc = anchors.shape[1], anchors.shape[0], nc
        self.stride = torch.tensor(stride)


class FakeModel(torch.nn.Module):
    def __init__(self, anchors, nc, hyp):
        super().__init__()
        self.model = [FakeDetect(anchors, nc)]
        self.hyp = hyp


def make_loss(anchors, nc, hyp, autobalance=False):
    from object_detection_amd.yolo_loss import ComputeLoss
    return ComputeLoss(FakeModel(anchors.cuda(), nc, hyp), autobalance=autobalance)


def close(got, want, rtol=1e-5, floor=1e-6):
    """Loss values and loss items (a handful of numbers of one magnitude); gradients go through all_grads_close."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    scale = float(np.abs(want).max()) if want.size else 0.0
    np.testing.assert_allclose(got, want, rtol=rtol, atol=floor * max(scale, 1e-30))


@pytest.mark.parametrize("tag", ["nt0", "nt1", "nt80", "nt640", "edges", "rect"])
def test_build_targets_golden(golden, tag):
    g = golden("build_targets")
    crit = make_loss(T(g["anchors"]), 80, dict(oy.HYP_SCRATCH))
    shapes = g[f"{tag}_shapes"]
    bs = 4 if tag == "rect" else 8
    p = [torch.empty((bs, 3, int(ny), int(nx), 85), device="cuda") for ny, nx in shapes]
    tcls, tbox, indices, anch = crit.build_targets(p, T(g[f"{tag}_targets"]).reshape(-1, 6).cuda())
    for l in range(3):
        b, a, gj, gi = indices[l]
        assert b.dtype == torch.int64 and tcls[l].dtype == torch.int64
        idx = torch.stack((b, a, gj, gi), 1).cpu().numpy().reshape(-1, 4)
        np.testing.assert_array_equal(idx, g[f"{tag}_l{l}_idx"])
        np.testing.assert_array_equal(tcls[l].cpu().numpy(), g[f"{tag}_l{l}_tcls"])
        np.testing.assert_array_equal(tbox[l].cpu().numpy().reshape(-1, 4), g[f"{tag}_l{l}_tbox"].reshape(-1, 4))
        np.testing.assert_array_equal(anch[l].cpu().numpy().reshape(-1, 2), g[f"{tag}_l{l}_anch"].reshape(-1, 2))


def test_build_targets_bs64_vs_oracle():
    rs = np.random.RandomState(12)
    anchors = oy.grid_anchors()
    targets = synth_targets(rs, 640, 64)
    targets[:50, 2:4] = np.round(targets[:50, 2:4] * 80) / 80          # centres exactly on cell corners of the finest grid
    targets[50:100, 2:4] = (np.floor(targets[50:100, 2:4] * 40) + 0.5) / 40   # exactly on the half-cell boundary
    crit = make_loss(anchors, 80, dict(oy.HYP_SCRATCH))
    shapes = [(80, 80), (40, 40), (20, 20)]
    p = [torch.empty((64, 3, ny, nx, 85), device="cuda") for ny, nx in shapes]
    tcls, tbox, indices, anch = crit.build_targets(p, T(targets).cuda())
    want = oy.build_targets(shapes, targets, anchors.numpy(), 4.0)
    for l, w in enumerate(want):
        got = torch.stack(indices[l], 1).cpu().numpy()
        np.testing.assert_array_equal(got, np.stack((w["b"], w["a"], w["gj"], w["gi"]), 1))
        np.testing.assert_array_equal(tbox[l].cpu().numpy(), w["tbox"])
        np.testing.assert_array_equal(anch[l].cpu().numpy(), w["anch"])
        np.testing.assert_array_equal(tcls[l].cpu().numpy(), w["tcls"])
    assert sum(len(w["b"]) for w in want) > 3000


LOSS_TAGS = ["base128", "nt0", "smooth", "focal", "posw", "autobal", "nc1", "sortiou", "sortiou_gr", "gr_half"]


def golden_loss(g, tag):
    """The criterion a fixture case was made with: constructor arguments, then the attributes set on the object."""
    img, bs, nc, autobal, calls = (int(v) for v in g[f"{tag}_meta"])
    hyp = dict(zip(HYP_KEYS, (float(v) for v in g[f"{tag}_hyp"])))
    crit = make_loss(T(g["anchors"]), nc, hyp, bool(autobal))
    if f"{tag}_opts" in g:                                   # loss.py:94,113
        crit.sort_obj_iou, crit.gr = bool(g[f"{tag}_opts"][0]), float(g[f"{tag}_opts"][1])
    return crit, autobal, calls


def golden_oracle(g, tag):
    """The oracle with the same settings: its float64 evaluation is the yardstick for ill-conditioned gradient entries."""
    img, bs, nc, autobal, calls = (int(v) for v in g[f"{tag}_meta"])
    hyp = dict(zip(HYP_KEYS, (float(v) for v in g[f"{tag}_hyp"])))
    o = oy.YoloLossOracle(T(g["anchors"]), hyp, nc=nc, autobalance=bool(autobal))
    if f"{tag}_opts" in g:
        o.sort_obj_iou, o.gr = bool(g[f"{tag}_opts"][0]), float(g[f"{tag}_opts"][1])
    return o


@pytest.mark.parametrize("tag", LOSS_TAGS)
def test_compute_loss_golden(golden, tag):
    g = golden("compute_loss")
    crit, autobal, calls = golden_loss(g, tag)
    oracle = golden_oracle(g, tag)
    levels, tg = [g[f"{tag}_p{l}"] for l in range(3)], g[f"{tag}_targets"].reshape(-1, 6)
    for c in range(calls):
        p = [T(x).cuda().requires_grad_(True) for x in levels]
        loss, items = crit(p, T(tg).cuda())
        assert loss.shape == (1,) and items.shape == (3,) and not items.requires_grad
        grads = torch.autograd.grad(loss, p)
        close(loss, g[f"{tag}_call{c}_loss"])
        close(items, g[f"{tag}_call{c}_items"])
        all_grads_close(grads, [g[f"{tag}_call{c}_grad{l}"] for l in range(3)], loss_grads64(oracle, levels, tg))
        if autobal:
            oracle([T(x) for x in levels], T(tg))        # the oracle's balance weights move with the calls
    if autobal:
        close(np.array(crit.balance), g[f"{tag}_balance"])


def test_compute_loss_bs8_640_vs_oracle():
    """The reference's CPU-runnable configuration: bs 8 @ 640 px, 80 classes, 80 targets; upstream gradient != 1."""
    rs = np.random.RandomState(1001)
    anchors = oy.grid_anchors()
    levels = synth_head(rs, 8, 640, 80)
    targets = synth_targets(rs, 80, 8)
    targets[7] = targets[6]                                      # duplicate cell: last writer owns tobj
    hyp = dict(oy.HYP_SCRATCH)
    want_fn = oy.YoloLossOracle(anchors, hyp, nc=80)
    p = [T(x).requires_grad_(True) for x in levels]
    wl, wi = want_fn(p, T(targets))
    wg = torch.autograd.grad(wl * 3.0, p)
    crit = make_loss(anchors, 80, hyp)
    d = [T(x).cuda().requires_grad_(True) for x in levels]
    gl, gi = crit(d, T(targets).cuda())
    gg = torch.autograd.grad(gl * 3.0, d)
    close(gl, wl.detach())
    close(gi, wi)
    all_grads_close(gg, wg, loss_grads64(want_fn, levels, targets, 3.0))
    # property: the objectness gradient is dense, every other channel is zero off the assigned cells
    nz = (gg[0][..., :4] != 0).any(-1).sum().item()
    assert 0 < nz <= 5 * 3 * 80
    assert (gg[0][..., 4] != 0).all()


def test_full_size_bs64_640_pipeline_vs_oracle():
    """The headline configuration at full size: bs 64 @ 640 px, 80 classes, 640 targets.
    decode (1e-5), assignment (bit-exact), loss + gradients (1e-5), NMS on the oracle's decoded tensor (bit-exact)."""
    from object_detection_amd.general import non_max_suppression
    from object_detection_amd.yolo import detect_decode
    from oracle import nms as onms
    rs = np.random.RandomState(2002)
    bs, nc = 64, 80
    anchors = oy.grid_anchors()
    levels = synth_head(rs, bs, 640, nc)
    targets = synth_targets(rs, 640, bs)
    hyp = dict(oy.HYP_SCRATCH)
    want_fn = oy.YoloLossOracle(anchors, hyp, nc=nc)
    # plant confident predictions on the assigned cells so that NMS has clusters to resolve
    for l, t in enumerate(want_fn.targets_for([T(x) for x in levels], T(targets))):
        levels[l][t["b"], t["a"], t["gj"], t["gi"], 4] = 1.5 + rs.normal(0, 1, len(t["b"]))
        levels[l][t["b"], t["a"], t["gj"], t["gi"], 5 + t["tcls"]] = 2.0 + rs.normal(0, 1, len(t["b"]))
        levels[l][t["b"], t["a"], t["gj"], t["gi"], 0:4] *= 0.3
    p = [T(x).requires_grad_(True) for x in levels]
    wl, wi = want_fn(p, T(targets))
    wg = torch.autograd.grad(wl, p)
    crit = make_loss(anchors, nc, hyp)
    d = [T(x).cuda().requires_grad_(True) for x in levels]
    gl, gi = crit(d, T(targets).cuda())
    gg = torch.autograd.grad(gl, d)
    close(gl, wl.detach())
    close(gi, wi)
    w64 = loss_grads64(want_fn, levels, targets)()
    all_grads_close(gg, wg, w64)
    # the fused forward+backward entry (five launches) against the same oracle values
    fl, fi, fg = crit.forward_backward(d, T(targets).cuda())
    close(fl, wl.detach())
    close(fi, wi)
    all_grads_close(fg, wg, w64)
    del w64
    assert int(crit.last_stats[3].item()) == sum(int(c) for c in crit.last_positive_counts.tolist()) > 3000
    del gg, wg, fg
    wz = oy.detect_decode([T(x) for x in levels], anchors, oy.YOLOV5S_STRIDES)
    gz = detect_decode([x.detach() for x in d], anchors.cuda(), torch.tensor(oy.YOLOV5S_STRIDES))
    decoded_close(gz, wz, level_rows(levels), oy.YOLOV5S_STRIDES)
    # NMS is fed the oracle's decoded tensor so that the comparison is exact (stage-wise parity)
    for kw in (dict(conf_thres=0.25, iou_thres=0.45, max_det=300), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)):
        want = onms.non_max_suppression(wz.clone(), **kw)
        got = non_max_suppression(wz.cuda(), **kw)
        assert [int(g.shape[0]) for g in got] == [int(w.shape[0]) for w in want]
        for g, w in zip(got, want):
            assert torch.equal(g.cpu(), w)
        assert sum(w.shape[0] for w in want) > 1000


def test_loss_after_decode_reads_compact_logits_bit_exact():
    """Validation evaluates the loss on the maps the decode has just read; the decode leaves a compact copy of the
    objectness logits which the loss may read instead of the strided column.  Same values, so same bits."""
    from object_detection_amd.yolo import detect_decode, objectness_logits
    rs = np.random.RandomState(77)
    anchors = oy.grid_anchors()
    levels = [T(x).cuda() for x in synth_head(rs, 4, 320, 80)]
    targets = T(synth_targets(rs, 40, 4)).cuda()
    crit = make_loss(anchors, 80, dict(oy.HYP_SCRATCH))
    p = [x.clone().requires_grad_(True) for x in levels]
    assert objectness_logits(p) is None
    l0, i0 = crit(p, targets)
    g0 = torch.autograd.grad(l0, p)
    detect_decode(p, anchors.cuda(), torch.tensor(oy.YOLOV5S_STRIDES))
    plane = objectness_logits(p)
    assert plane is not None and plane.shape == (4, 3 * (40 * 40 + 20 * 20 + 10 * 10))
    want = torch.cat([x.detach()[..., 4].reshape(4, -1) for x in p], 1)
    assert torch.equal(plane, want)
    assert objectness_logits([x.detach() for x in p]) is plane      # another view of the same, unmodified storage
    l1, i1 = crit(p, targets)
    g1 = torch.autograd.grad(l1, p)
    assert torch.equal(l0, l1) and torch.equal(i0, i1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # any in-place write to a map invalidates the copy, and the loss goes back to the strided column
    with torch.no_grad():
        p[1][0, 0, 0, 0, 4] += 3.0
    assert objectness_logits(p) is None
    l2, _ = crit(p, targets)
    assert not torch.equal(l2, l1)
    q = [x.detach().clone().requires_grad_(True) for x in p]
    l3, _ = crit(q, targets)
    assert torch.equal(l2, l3)


def test_forward_backward_in_two_halves_matches_the_fused_call():
    """ComputeLoss.positives on a side stream, then forward_backward(positives=...) on the main one: the same kernels
    as the single call (float atomics on shared cells and the reduction's grouping may move the last bits)."""
    from object_detection_amd.yolo import detect_decode
    from object_detection_amd.yolo_loss import ComputeLoss
    rs = np.random.RandomState(77)
    anchors = oy.grid_anchors()
    levels = [T(x).cuda() for x in synth_head(rs, 8, 320, 80)]
    targets = T(synth_targets(rs, 90, 8)).cuda()
    crit = ComputeLoss(FakeModel(anchors.cuda(), 80, dict(oy.HYP_SCRATCH)))
    l0, i0, g0 = crit.forward_backward(levels, targets)
    side = torch.cuda.Stream()
    for decode_between in (False, True):
        with torch.cuda.stream(side):
            st = crit.positives(levels, targets)
            done = torch.cuda.Event()
            done.record()
        if decode_between:      # the dense half then reads the objectness logits the decode left behind
            detect_decode(levels, anchors.cuda(), torch.tensor(oy.YOLOV5S_STRIDES))
        torch.cuda.current_stream().wait_event(done)
        l1, i1, g1 = crit.forward_backward(levels, targets, positives=st)
        close(l1, l0, rtol=1e-6)
        close(i1, i0, rtol=1e-6)
        for a, b in zip(g1, g0):
            assert torch.equal(a, b)          # the same kernels on the same values
    with pytest.raises(RuntimeError):
        crit.forward_backward([x.clone() for x in levels], targets, positives=st)


def test_gradient_is_bit_reproducible_on_shared_cells():
    """Many targets on the same few cells: the rows of a cell are summed in row order by one wave, so the gradient is the
    same bit for bit on every run (and within 1e-5 of the oracle).  A cell shared by more rows than one wave holds (here
    120 copies of one target) takes the atomic fallback and is only compared with the oracle."""
    from object_detection_amd.yolo_loss import ComputeLoss
    rs = np.random.RandomState(5)
    anchors = oy.grid_anchors()
    levels = synth_head(rs, 4, 320, 80)
    crowd = synth_targets(rs, 60, 4)
    crowd[:, 2:4] = 0.5 + rs.uniform(-0.08, 0.08, (60, 2))           # 60 boxes around the centres of two images
    crowd[:, 4:6] = rs.uniform(0.1, 0.3, (60, 2))
    crowd[:, 0] = rs.randint(0, 2, 60)
    hyp = dict(oy.HYP_SCRATCH)
    crit = ComputeLoss(FakeModel(anchors.cuda(), 80, hyp))
    d = [T(x).cuda() for x in levels]
    tg = T(crowd).cuda()
    l0, i0, g0 = crit.forward_backward(d, tg)
    for cell_index in crit.build_targets(d, tg)[2]:                   # up to 7 / 15 / 24 rows on one cell of the three levels
        rows_per_cell = np.unique(np.stack([a.cpu().numpy() for a in cell_index], 1), axis=0, return_counts=True)[1]
        assert 3 <= rows_per_cell.max() <= 64 and (rows_per_cell >= 3).sum() > 20
    for _ in range(4):
        l1, i1, g1 = crit.forward_backward(d, tg)
        assert torch.equal(l1, l0) and torch.equal(i1, i0)
        for a, b in zip(g1, g0):
            assert torch.equal(a, b)
    p = [T(x).requires_grad_(True) for x in levels]
    wl, wi = oy.YoloLossOracle(anchors, hyp, nc=80)(p, T(crowd))
    wg = torch.autograd.grad(wl, p)
    close(l0, wl.detach())
    all_grads_close(g0, wg, loss_grads64(oy.YoloLossOracle(anchors, hyp, nc=80), levels, crowd), suspects=0.05)
    # the autograd path runs the same kernels
    q = [x.clone().requires_grad_(True) for x in d]
    la, _ = crit(q, tg)
    for a, b in zip(torch.autograd.grad(la, q), g0):
        assert torch.equal(a, b)
    # more rows on one cell than a wave takes
    many = np.repeat(crowd[:1], 120, axis=0)
    lm, im, gm = crit.forward_backward(d, T(many).cuda())
    wl, wi = oy.YoloLossOracle(anchors, hyp, nc=80)(p, T(many))
    wg = torch.autograd.grad(wl, p)
    close(lm, wl.detach())
    # 120 terms per element, added by atomics in any order: 2e-5
    all_grads_close(gm, wg, loss_grads64(oy.YoloLossOracle(anchors, hyp, nc=80), levels, many), rtol=2e-5, suspects=1.0)


def test_forward_backward_autobalance_golden(golden):
    """autobalance=True (loss.py:157-161) through the autograd-free entry: the reference's values call after call (the
    balance weights move between calls) and its final balance."""
    g = golden("compute_loss")
    tag = "autobal"
    img, bs, nc, autobal, calls = (int(v) for v in g[f"{tag}_meta"])
    assert autobal and calls > 1
    hyp = dict(zip(HYP_KEYS, (float(v) for v in g[f"{tag}_hyp"])))
    crit = make_loss(T(g["anchors"]), nc, hyp, True)
    oracle = golden_oracle(g, tag)
    for c in range(calls):
        p = [T(g[f"{tag}_p{l}"]).cuda() for l in range(3)]
        loss, items, grads = crit.forward_backward(p, T(g[f"{tag}_targets"]).reshape(-1, 6).cuda())
        close(loss, g[f"{tag}_call{c}_loss"])
        close(items, g[f"{tag}_call{c}_items"])
        levels, tg = [g[f"{tag}_p{l}"] for l in range(3)], g[f"{tag}_targets"].reshape(-1, 6)
        all_grads_close(grads, [g[f"{tag}_call{c}_grad{l}"] for l in range(3)], loss_grads64(oracle, levels, tg))
        oracle([T(x) for x in levels], T(tg))
    close(np.array(crit.balance), g[f"{tag}_balance"])


@pytest.mark.parametrize("tag", ["sortiou", "sortiou_gr", "gr_half"])
def test_forward_backward_sort_obj_iou_golden(golden, tag):
    """sort_obj_iou / gr set on the object (loss.py:94,113,140-143) through the autograd-free entry and through its two
    halves: the reference's values."""
    g = golden("compute_loss")
    crit, _, _ = golden_loss(g, tag)
    p = [T(g[f"{tag}_p{l}"]).cuda() for l in range(3)]
    tg = T(g[f"{tag}_targets"]).reshape(-1, 6).cuda()
    for halves in (False, True):
        kw = dict(positives=crit.positives(p, tg)) if halves else {}
        loss, items, grads = crit.forward_backward(p, tg, **kw)
        close(loss, g[f"{tag}_call0_loss"])
        close(items, g[f"{tag}_call0_items"])
        all_grads_close(grads, [g[f"{tag}_call0_grad{l}"] for l in range(3)],
                        loss_grads64(golden_oracle(g, tag), [g[f"{tag}_p{l}"] for l in range(3)], g[f"{tag}_targets"].reshape(-1, 6)))


def test_sort_obj_iou_on_crowded_cells_vs_oracle():
    """Up to 24 rows on one cell, and 120 copies of one target: with sort_obj_iou the cell's objectness target is that of
    its best row, whichever path (list walk or atomic fallback) the gradient takes; also on the convolution layout."""
    from object_detection_amd.yolo_loss import ComputeLoss, conv_layout_of
    rs = np.random.RandomState(5)
    anchors = oy.grid_anchors()
    levels = synth_head(rs, 4, 320, 80)
    crowd = synth_targets(rs, 60, 4)
    crowd[:, 2:4] = 0.5 + rs.uniform(-0.08, 0.08, (60, 2))
    crowd[:, 4:6] = rs.uniform(0.1, 0.3, (60, 2))
    crowd[:, 0] = rs.randint(0, 2, 60)
    many = np.repeat(crowd[:1], 120, axis=0)
    many[:, 4:6] *= rs.uniform(0.7, 1.4, (120, 1)).astype(np.float32)
    hyp = dict(oy.HYP_SCRATCH)
    crit = ComputeLoss(FakeModel(anchors.cuda(), 80, hyp))
    crit.sort_obj_iou, crit.gr = True, 0.75
    want_fn = oy.YoloLossOracle(anchors, hyp, nc=80)
    want_fn.sort_obj_iou, want_fn.gr = True, 0.75
    plain = oy.YoloLossOracle(anchors, hyp, nc=80)
    plain.gr = 0.75
    d = [T(x).cuda() for x in levels]
    for targets in (crowd, many):
        p = [T(x).requires_grad_(True) for x in levels]
        wl, wi = want_fn(p, T(targets))
        wg = torch.autograd.grad(wl, p)
        assert abs(float(plain(p, T(targets))[1][1]) - float(wi[1])) > 1e-4 * float(wi[1])    # the flag matters here
        w64 = loss_grads64(want_fn, levels, targets)()
        tol = dict(rtol=2e-5, suspects=1.0) if targets is many else dict(suspects=0.05)     # 120 atomically added terms: 2e-5
        loss, items, grads = crit.forward_backward(d, T(targets).cuda())
        close(loss, wl.detach())
        close(items, wi)
        all_grads_close(grads, wg, w64, **tol)
        q = [x.clone().requires_grad_(True) for x in d]
        la, ia = crit(q, T(targets).cuda())
        close(la, wl.detach())
        close(ia, wi)
        all_grads_close(torch.autograd.grad(la, q), wg, w64, **tol)
        # the same maps as permuted views of convolution outputs (bs, na*no, ny, nx)
        convs = [x.permute(0, 1, 4, 2, 3).reshape(x.shape[0], -1, x.shape[2], x.shape[3]).contiguous() for x in d]
        views = [c.view(c.shape[0], 3, 85, c.shape[2], c.shape[3]).permute(0, 1, 3, 4, 2) for c in convs]
        assert conv_layout_of(views) is not None
        lc, ic, gc = crit.forward_backward(views, T(targets).cuda())
        close(lc, wl.detach())
        close(ic, wi)
        all_grads_close(gc, wg, w64, **tol)
