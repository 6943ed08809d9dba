"""GPU anchor-fit metrics against the reference's closures (golden fixture): ratios and counts exact, the fp32 mean of the
fitness to 1e-6 (the reference adds fp32 values in torch's order, the kernel in float64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["default", "poor", "tight", "one"])
def test_anchor_metric_golden(golden, tag):
    from object_detection_amd import autoanchor as A
    g = golden("autoanchor")
    k, wh, thr = g[f"{tag}_k"], torch.from_numpy(g[f"{tag}_wh"]).cuda(), float(g[f"{tag}_thr"])
    x, best = A.metric(k, wh)
    np.testing.assert_array_equal(x.cpu().numpy(), g[f"{tag}_x"])
    np.testing.assert_array_equal(best.cpu().numpy(), g[f"{tag}_best"])
    bpr, aat = A.check_metric(torch.from_numpy(k).cuda(), wh, thr)
    want = g[f"{tag}_stats"]
    assert bpr.dtype == torch.float32 and bpr.item() == want[0] and aat.item() == want[1]
    np.testing.assert_allclose(A.anchor_fitness(k, wh, 1 / thr).item(), want[2], rtol=1e-6)


def test_anchor_metric_large_vs_oracle():
    from object_detection_amd import autoanchor as A
    from oracle import autoanchor as oa
    rs = np.random.RandomState(3)
    wh = torch.from_numpy(np.exp(rs.uniform(np.log(1), np.log(640), (300000, 2))).astype(np.float32))
    k = torch.from_numpy(np.exp(rs.uniform(np.log(4), np.log(400), (12, 2))).astype(np.float32))
    x, best = A.metric(k, wh.cuda())
    wx, wbest = oa.metric(k, wh)
    assert torch.equal(x.cpu(), wx) and torch.equal(best.cpu(), wbest)
    bpr, aat = A.check_metric(k, wh.cuda(), 4.0)
    wb, wa = oa.check_metric(k, wh, 4.0)
    assert bpr.item() == wb.item() and aat.item() == wa.item()
    np.testing.assert_allclose(A.anchor_fitness(k, wh.cuda(), 0.25).item(), oa.anchor_fitness(k, wh, 0.25).item(), rtol=1e-6)
    with pytest.raises(RuntimeError):
        A.metric(k[:, :1], wh.cuda())
