"""GPU parity on the shapes and dtypes around the common case: other level counts, class counts beyond the
unrolled paths, half-precision and non-contiguous inputs, spans that cannot use 16-byte vectors, the 1280 px head."""
import numpy as np
import pytest
import torch

from oracle import nms as onms
from oracle import yolo as oy
from tests._synth import synth_head, synth_pred, synth_targets
from tests.test_gpu_nms import same_lists
from tests._checks import all_grads_close, decoded_close, level_rows, loss_grads64
from tests.test_gpu_yolo_loss import FakeModel, close

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_detect_module_matches_reference_contract():
    """Detect: training branch returns the permuted maps, eval returns (decoded, maps); buffers and parameters carry
    the reference's names so its checkpoints load (yolo.py:33-70)."""
    from object_detection_amd.yolo import Detect
    torch.manual_seed(3)
    anchors = ((10, 13, 16, 30, 33, 23), (30, 61, 62, 45, 59, 119), (116, 90, 156, 198, 373, 326))
    det = Detect(nc=7, anchors=anchors, ch=(16, 32, 64)).cuda()
    det.stride = torch.tensor([8., 16., 32.])
    det.anchors /= det.stride.view(-1, 1, 1).cuda()
    assert sorted(det.state_dict()) == sorted(["anchors"] + [f"m.{i}.{k}" for i in range(3) for k in ("weight", "bias")])
    feats = [torch.randn(2, c, 64 // s, 48 // s, device="cuda") for c, s in zip((16, 32, 64), (8, 16, 32))]
    det.train()
    maps = det([f.clone() for f in feats])
    assert [tuple(m.shape) for m in maps] == [(2, 3, 8, 6, 12), (2, 3, 4, 3, 12), (2, 3, 2, 1, 12)]
    det.eval()
    with torch.no_grad():
        z, maps2 = det([f.clone() for f in feats])
    for a, b in zip(maps, maps2):
        assert torch.equal(a, b)
    want = oy.detect_decode([m.detach().cpu() for m in maps], det.anchors.cpu(), [8., 16., 32.])
    assert z.shape == (2, 3 * (48 + 12 + 2), 12)
    xy_tol = 1e-5 * 32
    np.testing.assert_allclose(z[..., :2].cpu().numpy(), want[..., :2].numpy(), rtol=1e-5, atol=xy_tol)
    np.testing.assert_allclose(z[..., 2:].cpu().numpy(), want[..., 2:].numpy(), rtol=1e-5, atol=1e-7)


def test_decode_scalar_path_and_single_class():
    """Level spans that are not a multiple of four floats (no 16-byte vectors), nc = 1, one anchor."""
    from object_detection_amd.general import detect_nms, non_max_suppression
    from object_detection_amd.yolo import detect_decode
    rs = np.random.RandomState(2)
    levels = [rs.normal(0, 1.5, (3, 1, ny, nx, 6)).astype(np.float32) for ny, nx in ((7, 5), (3, 3))]
    anchors = torch.tensor([[[1.5, 2.0]], [[3.0, 2.5]]])
    strides = [8., 16.]
    want = oy.detect_decode([T(x) for x in levels], anchors, strides)
    dev = [T(x).cuda() for x in levels]
    got = detect_decode(dev, anchors.cuda(), torch.tensor(strides))
    decoded_close(got, want, level_rows(levels), strides)
    for kw in (dict(conf_thres=0.05, iou_thres=0.3), dict(conf_thres=0.05, iou_thres=0.3, multi_label=True)):
        two = non_max_suppression(got, **kw)
        same_lists(two, onms.non_max_suppression(got.cpu().clone(), **kw))
        same_lists(detect_nms(dev, anchors.cuda(), torch.tensor(strides), **kw), two)
    assert sum(t.shape[0] for t in two) > 3


@pytest.mark.parametrize("nc", [1, 2, 64, 65, 128, 129, 200])
def test_nms_class_count_paths(nc):
    """nc <= 64, <= 128 and beyond use different class-score prefetch code in the prefilter."""
    from object_detection_amd.general import non_max_suppression
    rs = np.random.RandomState(nc)
    pred = synth_pred(rs, 2, 3000, nc, n_obj=5, k_per_obj=20)
    for kw in (dict(conf_thres=0.2, iou_thres=0.45), dict(conf_thres=0.05, iou_thres=0.5, multi_label=True)):
        same_lists(non_max_suppression(T(pred).cuda(), **kw), onms.non_max_suppression(T(pred).clone(), **kw))


def test_half_and_non_contiguous_inputs():
    """fp16 inputs are computed in fp32 on the converted values; strided views are accepted."""
    from object_detection_amd.general import non_max_suppression
    from object_detection_amd.yolo import detect_decode
    from object_detection_amd.yolo_loss import ComputeLoss
    rs = np.random.RandomState(9)
    anchors = oy.grid_anchors()
    levels = synth_head(rs, 2, 160, 80)
    targets = synth_targets(rs, 12, 2)
    half = [T(x).cuda().half() for x in levels]
    as32 = [h.float() for h in half]
    strides = torch.tensor(oy.YOLOV5S_STRIDES)
    z16, z32 = detect_decode(half, anchors.cuda(), strides), detect_decode(as32, anchors.cuda(), strides)
    assert z16.dtype == torch.float32 and torch.equal(z16, z32)
    crit = ComputeLoss(FakeModel(anchors.cuda(), 80, dict(oy.HYP_SCRATCH)))
    p16 = [h.clone().requires_grad_(True) for h in half]
    p32 = [h.clone().requires_grad_(True) for h in as32]
    l16, _ = crit(p16, T(targets).cuda())
    l32, _ = crit(p32, T(targets).cuda())
    g16, g32 = torch.autograd.grad(l16, p16), torch.autograd.grad(l32, p32)
    assert torch.equal(l16, l32)
    for a, b in zip(g16, g32):
        assert a.dtype == torch.float16 and torch.equal(a, b.half())
    # a strided view of a wider tensor (every other anchor row) and a half-precision prediction tensor
    pred = synth_pred(rs, 2, 4000, 6, n_obj=5, k_per_obj=20)
    wide = T(pred).cuda()
    view = wide[:, ::2]
    assert not view.is_contiguous()
    same_lists(non_max_suppression(view, 0.2, 0.45), onms.non_max_suppression(T(pred)[:, ::2].clone(), 0.2, 0.45))
    h = wide.half()              # an fp16 tensor gets the reference's fp16 products and corners (tests/test_gpu_half.py)
    same_lists(non_max_suppression(h, 0.2, 0.45), onms.non_max_suppression(h.cpu(), 0.2, 0.45))
    same_lists(non_max_suppression(h.float(), 0.2, 0.45), onms.non_max_suppression(h.float().cpu(), 0.2, 0.45))


@pytest.mark.parametrize("shapes,nc", [(((20, 20), (10, 10), (5, 5)), 80), (((7, 5), (3, 3)), 1), (((9, 11), (4, 6), (2, 3)), 130)])
def test_half_precision_head_is_read_directly(shapes, nc):
    """fp16 head maps (model.half(), val.py:115,129) go to the kernels as they are: decode and the fused decode+NMS give
    exactly what the maps converted with .float() give -- every prefilter variant, vector and scalar level spans."""
    from object_detection_amd.general import detect_nms
    from object_detection_amd.yolo import detect_decode
    rs = np.random.RandomState(nc)
    na, bs = 3, 3
    anchors = torch.from_numpy(rs.uniform(1.0, 6.0, (len(shapes), na, 2)).astype(np.float32)).cuda()
    strides = torch.tensor([8.0 * 2 ** i for i in range(len(shapes))])
    half = []
    for ny, nx in shapes:
        x = rs.normal(0, 1.5, (bs, na, ny, nx, nc + 5)).astype(np.float32)
        x[..., 4] -= 1.0
        half.append(T(x).cuda().half())
    as32 = [h.float() for h in half]
    z16, z32 = detect_decode(half, anchors, strides), detect_decode(as32, anchors, strides)
    assert z16.dtype == torch.float32 and torch.equal(z16, z32)
    some = 0
    for kw in (dict(conf_thres=0.25, iou_thres=0.45), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
               dict(conf_thres=0.1, iou_thres=0.5, multi_label=True), dict(conf_thres=0.05, iou_thres=0.5, agnostic=True)):
        got, want = detect_nms(half, anchors, strides, **kw), detect_nms(as32, anchors, strides, **kw)
        same_lists(got, want)
        some += sum(g.shape[0] for g in got)
    assert some > 10
    # an odd base address (a view that starts one element into a buffer) still works: it is made contiguous first
    buf = torch.zeros(half[0].numel() + 1, dtype=torch.float16, device="cuda")
    buf[1:] = half[0].reshape(-1)
    view = [buf[1:].view(half[0].shape)] + half[1:]
    assert torch.equal(detect_decode(view, anchors, strides), z32)


@pytest.mark.parametrize("strides", [(8., 16., 32., 64.), (8., 16., 32., 64., 128.)])
def test_loss_and_decode_with_more_levels(strides):
    """P6/P7-style heads: 4 or 5 levels take the 5-entry objectness balance (loss.py:111)."""
    from object_detection_amd.yolo import detect_decode
    from object_detection_amd.yolo_loss import ComputeLoss
    rs = np.random.RandomState(len(strides))
    nl, nc, bs, img = len(strides), 5, 2, 256
    anchors = torch.from_numpy(rs.uniform(1.0, 5.0, (nl, 3, 2)).astype(np.float32))
    levels = [rs.normal(0, 1, (bs, 3, max(int(img // s), 1), max(int(img // s), 1), nc + 5)).astype(np.float32) for s in strides]
    targets = synth_targets(rs, 30, bs, nc)
    hyp = dict(oy.HYP_SCRATCH)
    want_fn = oy.YoloLossOracle(anchors, hyp, nc=nc, strides=strides)
    p = [T(x).requires_grad_(True) for x in levels]
    wl, wi = want_fn(p, T(targets))
    wg = torch.autograd.grad(wl, p)
    model = FakeModel(anchors.cuda(), nc, hyp)
    model.model[0].stride = torch.tensor(strides)
    crit = ComputeLoss(model)
    assert len(crit.balance) == 5
    d = [T(x).cuda().requires_grad_(True) for x in levels]
    gl, gi = crit(d, T(targets).cuda())
    gg = torch.autograd.grad(gl, d)
    close(gl, wl.detach())
    close(gi, wi)
    all_grads_close(gg, wg, loss_grads64(want_fn, levels, targets))
    z = detect_decode([x.detach() for x in d], anchors.cuda(), torch.tensor(strides))
    want = oy.detect_decode([T(x) for x in levels], anchors, list(strides))
    decoded_close(z, want, level_rows(levels), strides)


def test_1280px_head_decode_nms_vs_oracle():
    """One image of the 1280 px sweep: 100 800 anchors, validation thresholds, multi-label."""
    from object_detection_amd.general import detect_nms, non_max_suppression
    from object_detection_amd.yolo import detect_decode
    rs = np.random.RandomState(1280)
    anchors = oy.grid_anchors()
    levels = []
    for s in (8, 16, 32):
        n = 1280 // s
        x = rs.normal(0, 1, (1, 3, n, n, 85)).astype(np.float32)
        x[..., 4] += np.log(8 / (1280 / s) ** 2)
        x[..., 5:] += np.log(0.6 / (80 - 0.999999))
        levels.append(x)
    dev = [T(x).cuda() for x in levels]
    strides = torch.tensor(oy.YOLOV5S_STRIDES)
    z = detect_decode(dev, anchors.cuda(), strides)
    assert z.shape == (1, 100800, 85)
    want = oy.detect_decode([T(x) for x in levels], anchors, oy.YOLOV5S_STRIDES)
    decoded_close(z, want, level_rows(levels), oy.YOLOV5S_STRIDES)
    kw = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)
    ref = onms.non_max_suppression(z.cpu().clone(), **kw)
    same_lists(non_max_suppression(z, **kw), ref)
    same_lists(detect_nms(dev, anchors.cuda(), strides, **kw), ref)
    assert ref[0].shape[0] == 300


def test_tta_descale_and_clip_vs_oracle():
    """Augmented inference: de-scale / de-flip in place (bit-exact: IEEE division and subtraction), tail clipping."""
    from object_detection_amd.general import clip_coords
    from object_detection_amd.yolo import clip_augmented, descale_pred
    rs = np.random.RandomState(4)
    base = synth_pred(rs, 2, 1260, 3)
    for flips, scale in ((None, 1.0), (3, 0.83), (2, 0.67), (None, 0.67)):
        want = oy.descale_pred(T(base).clone(), flips, scale, (480, 640))
        d = T(base).cuda()
        v0 = d._version
        got = descale_pred(d, flips, scale, (480, 640))
        assert got is d and d._version > v0
        np.testing.assert_array_equal(d.cpu().numpy(), want.numpy())
    ys = [T(base), T(base[:, :900]), T(base[:, :400])]
    want = oy.clip_augmented([y.clone() for y in ys], 3)
    got = clip_augmented([y.cuda() for y in ys], 3)
    for g, w in zip(got, want):
        assert torch.equal(g.cpu(), w)
    boxes = torch.rand(7, 4, device="cuda") * 900 - 100
    v0 = boxes._version
    clip_coords(boxes, (480, 640))
    assert boxes._version > v0
