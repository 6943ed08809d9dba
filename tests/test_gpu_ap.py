"""GPU ap_per_class against the reference's own outputs (golden fixture) and the oracle: float64, bit for bit."""
import warnings

import numpy as np
import pytest
import torch

from oracle import ap as oa
from tests.test_oracle_golden import AP_CASES, AP_OUT

pytestmark = pytest.mark.gpu


def check(got, want):
    for nm, g, w in zip(AP_OUT, got, want):
        assert g.dtype == w.dtype and g.shape == w.shape, nm
        np.testing.assert_array_equal(g, w, err_msg=nm)


@pytest.mark.parametrize("tag", AP_CASES)
def test_ap_per_class_golden(golden, tag):
    from object_detection_amd.metrics import ap_per_class
    g = golden("ap")
    ins = [g[f"{tag}_in_{k}"] for k in ("tp", "conf", "pred_cls", "target_cls")]
    want = [g[f"{tag}_{nm}"] for nm in AP_OUT]
    check(ap_per_class(*ins), want)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in ins]          # device-resident statistics
    check(ap_per_class(*dev), want)


@pytest.mark.parametrize("seed", range(8))
def test_ap_per_class_fuzz_vs_oracle(seed):
    """Random statistics with repeated confidences (stable order), classes without labels or detections, 1-D tp."""
    from object_detection_amd.metrics import ap_per_class
    rs = np.random.RandomState(600 + seed)
    n = int(rs.choice([1, 7, 300, 5000, 70000]))
    nc = int(rs.choice([1, 3, 80, 1000]))
    niou = int(rs.choice([1, 10]))
    m = int(rs.choice([1, 20, 3000]))
    target_cls = rs.randint(0, nc, m).astype(np.float64)
    pred_cls = rs.randint(0, nc + 2, n).astype(np.float32)
    conf = rs.choice(np.linspace(0.001, 0.999, max(n // 3, 2)).astype(np.float32), n)    # many equal confidences
    tp = np.zeros((n, niou), dtype=bool)
    for c in np.unique(target_cls):
        idx = np.flatnonzero(pred_cls == c)
        hits = idx[rs.rand(len(idx)) < 0.5][:int((target_cls == c).sum())]
        for h, d in zip(hits, rs.randint(1, niou + 1, len(hits))):
            tp[h, :d] = True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oa.ap_per_class(tp, conf, pred_cls, target_cls)
    check(ap_per_class(tp, conf, pred_cls, target_cls), want)


def test_ap_per_class_argument_errors():
    from object_detection_amd.metrics import ap_per_class
    tp = np.zeros((3, 10), dtype=bool)
    conf = np.array([0.9, 0.8, 0.7], dtype=np.float32)
    cls = np.zeros(3, dtype=np.float32)
    with pytest.raises(ValueError):
        ap_per_class(tp, conf, cls, np.array([0.5]))                    # not an integer class id
    with pytest.raises(ValueError):
        ap_per_class(tp, np.array([0.1, 0.2, 1 / 3]), cls, np.array([0.0]))   # float64 confidences that are not float32 values
    with pytest.raises(RuntimeError):
        ap_per_class(tp, conf[:2], cls, np.array([0.0]))
    out = ap_per_class(tp, conf, cls, np.zeros(0))                      # no labels at all: empty results
    assert out[5].shape == (0, 10) and out[6].shape == (0,)
