"""Oracle: anchor-to-label fit metrics of AutoAnchor (CPU, test-only).

Restates the closures of object_detection/yolov5/utils/autoanchor.py
  :35-41  check_anchors.metric      -> (bpr, aat): best possible recall, anchors above threshold per label
  :86-90  kmean_anchors.metric      -> (x, best): ratio metric of every label against every anchor, best per label
  :92-94  kmean_anchors.anchor_fitness
as plain functions (what the closures capture becomes an argument), fp32 torch operations in the reference's order.
"""
import torch


def metric(k, wh):
    """autoanchor.py:86-90.  k (n,2) anchors, wh (N,2) label sizes -> x (N,n), best (N,)."""
    r = wh[:, None] / k[None]
    x = torch.min(r, 1 / r).min(2)[0]
    return x, x.max(1)[0]


def anchor_fitness(k, wh, thr):
    """autoanchor.py:92-94; thr is the inverted threshold (autoanchor.py:84)."""
    _, best = metric(torch.as_tensor(k, dtype=torch.float32), wh)
    return (best * (best > thr).float()).mean()


def check_metric(k, wh, thr=4.0):
    """autoanchor.py:35-41 -> (bpr, aat)."""
    r = wh[:, None] / k[None]
    x = torch.min(r, 1 / r).min(2)[0]
    best = x.max(1)[0]
    aat = (x > 1 / thr).float().sum(1).mean()
    bpr = (best > 1 / thr).float().mean()
    return bpr, aat
