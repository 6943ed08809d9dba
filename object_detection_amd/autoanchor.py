"""AutoAnchor's anchor-to-label fit metrics on the device (object_detection/yolov5/utils/autoanchor.py:35-41, 86-94).

The reference keeps these as closures inside check_anchors / kmean_anchors; here they are functions whose extra
arguments are what the closures capture (the label sizes `wh`, the threshold)."""
import numpy as np
import torch

from . import _lib as L


def _launch(k, wh, thr, want_x, want_best):
    if not isinstance(wh, torch.Tensor):
        wh = torch.as_tensor(np.asarray(wh), dtype=torch.float32).cuda()
    wh = L.dev_f32(wh, "wh").contiguous()
    k = torch.as_tensor(np.asarray(k) if not isinstance(k, torch.Tensor) else k).to(device=wh.device, dtype=torch.float32).contiguous()
    if wh.dim() != 2 or wh.shape[1] != 2 or k.dim() != 2 or k.shape[1] != 2:
        raise RuntimeError(f"anchor metric: wh {tuple(wh.shape)} and k {tuple(k.shape)} must be (N,2) and (n,2)")
    N, n = wh.shape[0], k.shape[0]
    x = torch.empty((N, n), dtype=torch.float32, device=wh.device) if want_x else None
    best = torch.empty(N, dtype=torch.float32, device=wh.device) if want_best else None
    sums = torch.empty(3, dtype=torch.float64, device=wh.device)
    with L.device_guard(wh.device):
        L.check(L.lib().dh_anchor_metric(L.ptr(wh), N, L.ptr(k), n, float(thr), L.ptr(x), L.ptr(best), L.ptr(sums), L.stream()),
                "dh_anchor_metric")
    return x, best, sums, N


def metric(k, wh):
    """kmean_anchors.metric (autoanchor.py:86-90): ratio metric x (N,n) and its best value per label (N,)."""
    x, best, _, _ = _launch(k, wh, 0.0, True, True)
    return x, best


def anchor_fitness(k, wh, thr):
    """kmean_anchors.anchor_fitness (autoanchor.py:92-94) -> 0-dim fp32 host tensor; thr is the inverted threshold (:84).
    The mean is taken over a float64 sum (the reference adds fp32 values in torch's order: equal to ~1e-7 relative)."""
    _, _, sums, N = _launch(k, wh, thr, False, False)
    return torch.tensor(np.float32(sums[0].item() / N))


def check_metric(k, wh, thr=4.0):
    """The metric of check_anchors (autoanchor.py:35-41) -> (bpr, aat) 0-dim fp32 host tensors: best possible recall and
    anchors above threshold per label, for the threshold hyp['anchor_t'] = thr."""
    _, _, sums, N = _launch(k, wh, 1 / thr, False, False)
    _, n_best, n_x = sums.tolist()
    # fp32 sums of 0/1 values divided by N, as the reference's .float().mean() (one IEEE fp32 division, on the host:
    # the counts are needed there anyway, for the caller's comparison against 0.98)
    return torch.tensor(np.float32(n_best) / np.float32(N)), torch.tensor(np.float32(n_x) / np.float32(N))
