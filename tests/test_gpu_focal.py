"""GPU LabelSmoothingFocalLoss against the reference's outputs (golden fixture) and the oracle: value and gradient to
1e-5 relative (fp32 exp / log / pow differ by ulps between the CPU and GPU libraries)."""
import numpy as np
import pytest
import torch

from oracle import focal as of
from tests.test_oracle_golden import FOCAL_CASES, focal_kwargs

pytestmark = pytest.mark.gpu


def close(got, want, rtol=1e-5):
    want = np.asarray(want)
    np.testing.assert_allclose(np.asarray(got), want, rtol=rtol, atol=rtol * max(float(np.abs(want).max()), 1e-30))


@pytest.mark.parametrize("tag", FOCAL_CASES)
def test_focal_loss_golden(golden, tag):
    from object_detection_amd.loss import LabelSmoothingFocalLoss
    g = golden("focal")
    x = torch.from_numpy(g[f"{tag}_logits"]).cuda().requires_grad_(True)
    loss = LabelSmoothingFocalLoss(**focal_kwargs(g[f"{tag}_kw"]))(x, torch.from_numpy(g[f"{tag}_label"]).cuda())
    grad, = torch.autograd.grad(loss * 3.0, x)
    assert loss.dtype == torch.float32 and loss.dim() == 0
    close(loss.item(), g[f"{tag}_loss"])
    close(grad.cpu().numpy(), g[f"{tag}_grad"])


@pytest.mark.parametrize("label_dtype", [torch.float32, torch.int64, torch.float16])
def test_focal_loss_centernet_size_vs_oracle(label_dtype):
    """bs 4 x 80 classes x 160x160 target maps with a few hundred objects, gamma 8 as train.py:147-153."""
    from object_detection_amd.loss import LabelSmoothingFocalLoss
    rs = np.random.RandomState(8)
    N, C, W, H = 4, 80, 160, 160
    logits = rs.normal(-2.0, 1.5, (N, C, W, H)).astype(np.float32)
    label = np.zeros((N, C, W, H), dtype=np.float32)
    for _ in range(300):
        label[rs.randint(N), rs.randint(C), rs.randint(W), rs.randint(H)] = 1.0
    kw = dict(num_classes=C, one_hot_label_format=True, gamma=8.0, alpha=0.999, smoothing=0.1)
    x = torch.from_numpy(logits).requires_grad_(True)
    want = of.LabelSmoothingFocalLoss(**kw)(x, torch.from_numpy(label))
    gw, = torch.autograd.grad(want, x)
    d = torch.from_numpy(logits).cuda().requires_grad_(True)
    got = LabelSmoothingFocalLoss(**kw)(d, torch.from_numpy(label).cuda().to(label_dtype))
    gg, = torch.autograd.grad(got, d)
    close(got.item(), want.item())
    close(gg.cpu().numpy(), gw.numpy())
    assert int((gg != 0).sum()) <= 300 * C


@pytest.mark.parametrize("shape,one_hot", [((3, 300, 6, 10), True), ((50, 300), False), ((2, 130, 5, 7), True), ((4, 5, 9, 3), True)])
def test_focal_loss_class_counts_and_odd_shapes(shape, one_hot):
    """More classes than the per-cell pass keeps in registers, spatial sizes that rule out 16-byte label loads, index labels."""
    from object_detection_amd.loss import LabelSmoothingFocalLoss
    rs = np.random.RandomState(shape[1])
    C = shape[1]
    logits = rs.normal(0, 2.0, shape).astype(np.float32)
    if one_hot:
        label = (rs.rand(*shape) < 0.02).astype(np.float32)
    else:
        label = rs.randint(0, C, shape[0]).astype(np.int64)
    kw = dict(num_classes=C, one_hot_label_format=one_hot, gamma=2.0, smoothing=0.1, ignore_index=None if one_hot else 3)
    x = torch.from_numpy(logits).requires_grad_(True)
    want = of.LabelSmoothingFocalLoss(**kw)(x, torch.from_numpy(label))
    gw, = torch.autograd.grad(want, x)
    d = torch.from_numpy(logits).cuda().requires_grad_(True)
    got = LabelSmoothingFocalLoss(**kw)(d, torch.from_numpy(label).cuda())
    gg, = torch.autograd.grad(got, d)
    close(got.item(), want.item())
    close(gg.cpu().numpy(), gw.numpy())


def test_focal_loss_half_logits_and_errors():
    from object_detection_amd.loss import LabelSmoothingFocalLoss
    rs = np.random.RandomState(2)
    logits = torch.from_numpy(rs.normal(0, 2, (3, 6, 8, 8)).astype(np.float32)).half()
    label = torch.zeros(3, 6, 8, 8)
    label[0, 1, 2, 3] = label[2, 5, 7, 7] = 1.0
    kw = dict(num_classes=6, one_hot_label_format=True, gamma=2.0)
    x = logits.float().requires_grad_(True)
    want = of.LabelSmoothingFocalLoss(**kw)(x, label)
    gw, = torch.autograd.grad(want, x)
    d = logits.cuda().requires_grad_(True)
    got = LabelSmoothingFocalLoss(**kw)(d, label.cuda())
    gg, = torch.autograd.grad(got, d)
    assert gg.dtype == torch.float16
    close(got.item(), want.item())
    close(gg.float().cpu().numpy(), gw.half().float().numpy(), rtol=2e-3)
    with pytest.raises(ValueError):
        LabelSmoothingFocalLoss(num_classes=1)
    with pytest.raises(ValueError):
        LabelSmoothingFocalLoss(num_classes=3, gamma=-1)
    with pytest.raises(RuntimeError):
        LabelSmoothingFocalLoss(**kw)(d, label[:, :5].cuda())
    with pytest.raises(RuntimeError):
        LabelSmoothingFocalLoss(num_classes=6)(d, torch.zeros(3, 8, 8, dtype=torch.int64).cuda())
    with pytest.raises(RuntimeError):
        LabelSmoothingFocalLoss(**kw)(d, label)                       # a CPU label


@pytest.mark.parametrize("tag", ["reg_l1", "reg_smooth"])
def test_regression_loss_golden(golden, tag):
    from object_detection_amd.loss import RegressionLossWithMask
    g = golden("focal")
    x = torch.from_numpy(g[f"{tag}_pred"]).cuda().requires_grad_(True)
    loss = RegressionLossWithMask(tag == "reg_smooth")(x, torch.from_numpy(g[f"{tag}_target"]).cuda(),
                                                       torch.from_numpy(g[f"{tag}_mask"]).cuda())
    grad, = torch.autograd.grad(loss * 2.0, x)
    close(loss.item(), g[f"{tag}_loss"])
    close(grad.cpu().numpy(), g[f"{tag}_grad"])


def test_regression_loss_full_size_and_compute_losses():
    """(B,2,160,160) maps against the oracle, and through compute_losses with criterion['box'] = None
    (detectors/centernet/loss.py:18-21)."""
    from object_detection_amd.centernet import compute_losses
    from object_detection_amd.loss import LabelSmoothingFocalLoss, RegressionLossWithMask
    rs = np.random.RandomState(12)
    B, C, W, H = 4, 6, 160, 160
    size_p, size_t = rs.uniform(0, 20, (B, 2, W, H)).astype(np.float32), rs.uniform(0, 20, (B, 2, W, H)).astype(np.float32)
    off_p, off_t = rs.uniform(0, 1, (B, 2, W, H)).astype(np.float32), rs.uniform(0, 1, (B, 2, W, H)).astype(np.float32)
    mask = (rs.rand(B, W, H) < 0.01).astype(np.float32)
    cls_p = rs.normal(-2, 1, (B, C, W, H)).astype(np.float32)
    cls_t = np.zeros((B, C, W, H), dtype=np.float32)
    cls_t[:, 2] = mask
    T = torch.from_numpy
    for smooth in (False, True):
        x = T(size_p).requires_grad_(True)
        want = of.RegressionLossWithMask(smooth)(x, T(size_t), T(mask))
        gw, = torch.autograd.grad(want, x)
        d = T(size_p).cuda().requires_grad_(True)
        got = RegressionLossWithMask(smooth)(d, T(size_t).cuda(), T(mask).cuda())
        gg, = torch.autograd.grad(got, d)
        close(got.item(), want.item())
        close(gg.cpu().numpy(), gw.numpy())
    crit = {"cls": LabelSmoothingFocalLoss(C, one_hot_label_format=True, gamma=2.0), "box": None,
            "size": RegressionLossWithMask(False), "offset": RegressionLossWithMask(True)}
    ocrit = {"cls": of.LabelSmoothingFocalLoss(C, one_hot_label_format=True, gamma=2.0), "box": None,
             "size": of.RegressionLossWithMask(False), "offset": of.RegressionLossWithMask(True)}
    losses, want = {}, {}
    compute_losses([T(a).cuda() for a in (cls_p, size_p, off_p)], [T(a).cuda() for a in (cls_t, size_t, off_t, mask)], crit, losses)
    want["cls"] = ocrit["cls"](T(cls_p), T(cls_t))
    want["size"] = ocrit["size"](T(size_p), T(size_t), T(mask))
    want["offset"] = ocrit["offset"](T(off_p), T(off_t), T(mask))
    assert sorted(losses) == ["cls", "offset", "size"]
    for k in want:
        close(losses[k].item(), want[k].item())
