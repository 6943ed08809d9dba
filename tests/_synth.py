"""Seeded synthetic inputs shared by tests/golden/make_golden.py, the tests and bench.py.

Everything is drawn from numpy's legacy ``RandomState`` (a bit-stable stream), so the
fixtures can store outputs only and the tests can rebuild the exact inputs.
"""
import math

import numpy as np


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def rand_boxes_xywh(rs, n, scale=1.0):
    xy = rs.uniform(0.1, 0.9, (n, 2)) * scale
    wh = np.exp(rs.uniform(np.log(0.02), np.log(0.5), (n, 2))) * scale
    return f32(np.concatenate([xy, wh], 1))


def synth_targets(rs, nt, bs, nc=80):
    t = np.zeros((nt, 6), dtype=np.float32)
    t[:, 0] = np.sort(rs.randint(0, bs, nt))       # collate_fn order: grouped by image (datasets.py:627-632)
    t[:, 1] = rs.randint(0, nc, nt)
    t[:, 2:4] = rs.uniform(0, 1, (nt, 2))
    t[:, 4:6] = np.exp(rs.uniform(np.log(0.01), np.log(0.6), (nt, 2)))
    return t


def synth_head(rs, bs, img, nc, nl_strides=(8, 16, 32)):
    """Raw head outputs with the bias statistics of yolo.py:196-204."""
    levels = []
    for s in nl_strides:
        n = img // s
        x = rs.normal(0, 1, (bs, 3, n, n, nc + 5))
        x[..., 4] += math.log(8 / (640 / s) ** 2)
        x[..., 5:] += math.log(0.6 / (nc - 0.999999))
        levels.append(f32(x))
    return levels


def synth_pred(rs, bs, A, nc, n_obj=6, img=640.0, jitter=6.0, k_per_obj=12):
    """Decoded-prediction-like tensor (bs,A,5+nc): sparse background + clusters of overlapping boxes."""
    pred = np.zeros((bs, A, 5 + nc), dtype=np.float32)
    pred[..., 0:2] = rs.uniform(0, img, (bs, A, 2))
    pred[..., 2:4] = np.exp(rs.uniform(np.log(8), np.log(200), (bs, A, 2)))
    pred[..., 4] = rs.uniform(0, 1, (bs, A)) ** 6                       # mostly low objectness
    pred[..., 5:] = rs.uniform(0, 1, (bs, A, nc)) ** 3
    for b in range(bs):
        for _ in range(n_obj):
            c = rs.randint(nc)
            box = np.concatenate([rs.uniform(60, img - 60, 2), rs.uniform(30, 160, 2)])
            rows = rs.choice(A, k_per_obj, replace=False)
            pred[b, rows, :4] = box + rs.normal(0, jitter, (k_per_obj, 4))
            pred[b, rows, 4] = rs.uniform(0.5, 1.0, k_per_obj)
            pred[b, rows, 5 + c] = rs.uniform(0.5, 1.0, k_per_obj)
    return pred


def dedupe_scores(pred, conf_thres):
    """Perturb obj so that every obj*cls product above half the threshold is unique within an image."""
    ncls = pred.shape[2] - 5
    for b in range(pred.shape[0]):
        for _ in range(200):
            conf = (pred[b, :, 5:] * pred[b, :, 4:5]).ravel()
            order = np.argsort(conf, kind="stable")
            same = (conf[order][1:] == conf[order][:-1]) & (conf[order][1:] > conf_thres * 0.5)
            if not same.any():
                break
            for r in sorted({int(i) // ncls for i in order[1:][same]}):   # later member of each equal pair only
                pred[b, r, 4] = np.nextafter(pred[b, r, 4], np.float32(0))
        else:
            raise RuntimeError("could not de-duplicate scores")
    return pred


def make_big_pred():
    """One image with > 30000 multi-label candidates (general.py:716-717), unique scores."""
    rs = np.random.RandomState(71)
    big = synth_pred(rs, 1, 20000, 2, n_obj=40, k_per_obj=50)
    low = big[..., 4] < 0.2
    big[..., 4][low] = rs.uniform(0.2, 0.5, int(low.sum()))
    big[..., 5:] = np.maximum(big[..., 5:], f32(rs.uniform(0.2, 0.4, big[..., 5:].shape)))
    return dedupe_scores(big, 0.001)


# The half-precision fixture (tests/golden/half.npz, tests/golden/make_golden.py gen_half): Detect cases taken from the fp32
# fixture's list (the last one is the torch.cat form, inplace=False) and NMS calls on an fp16 prediction tensor.
HALF_DETECT_TAGS = ("s3_nc80", "s3_nc7_rect", "s4_nc7", "s5_nc3_na2")

HALF_NMS_VARIANTS = {     # thresholds that fp16 represents exactly or rounds down, except multi_001 (see gen_half)
    "default": dict(conf_thres=0.25, iou_thres=0.45),
    "multi": dict(conf_thres=2.0 ** -10, iou_thres=0.6, multi_label=True),
    "multi_001": dict(conf_thres=0.001, iou_thres=0.6, multi_label=True),
    "multi_c2": dict(conf_thres=0.2, iou_thres=0.5, multi_label=True),
    "agnostic": dict(conf_thres=0.125, agnostic=True),
    "classes_multi": dict(conf_thres=0.0625, classes=[0, 2, 7], multi_label=True),
    "maxdet5": dict(conf_thres=0.0625, multi_label=True, max_det=5),
    "labels": dict(conf_thres=0.125),                                           # + the fixture's labels for images 0 and 2
}
