#!/usr/bin/env python3
"""Where does the one-workgroup NMS chain (k_select = -2) differ from the separate launches (k_select = 3840)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd.general import nms_finish, nms_launch  # noqa: E402
from tests.test_gpu_configs_full_size import dedupe_best_scores, dense_pred  # noqa: E402


def run(pred, k, **kw):
    rows, meta, cap = nms_launch(pred, k_select=k, **kw)
    host = meta.tolist()
    bs = pred.shape[0]
    res, need = nms_finish(rows, meta, cap)
    return res, need, host[2 * bs:3 * bs], host[3 * bs:]


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    if len(sys.argv) > 2 and sys.argv[2] == "filter":
        import numpy as np
        from tests._synth import synth_pred
        pred = torch.from_numpy(synth_pred(np.random.RandomState(101), 8, 25200, 80, n_obj=12, k_per_obj=40)).cuda()[:bs]
        kw = dict(conf_thres=0.02, iou_thres=0.3, multi_label=True, classes=[1, 3, 5, 79], max_det=50)
    else:
        pred = dedupe_best_scores(dense_pred(bs, 100_000, 80))
        kw = dict(conf_thres=0.25, iou_thres=0.45, max_det=300)
    a, need_a, cand_a, flags_a = run(pred, -2, **kw)
    b, need_b, cand_b, flags_b = run(pred, 3840, **kw)
    print("chain: need", need_a, "flags", flags_a, "cands", cand_a[:4])
    print("split: need", need_b, "flags", flags_b, "cands", cand_b[:4])
    if a is None or b is None:
        return
    for i, (x, y) in enumerate(zip(a, b)):
        if x.shape != y.shape or not torch.equal(x, y):
            x, y = x.cpu(), y.cpu()
            n = min(len(x), len(y))
            d = (x[:n] != y[:n]).any(1).nonzero().flatten()
            j = int(d[0]) if len(d) else n
            print(f"image {i}: {len(x)} vs {len(y)} rows, first difference at row {j}")
            print("  chain:", x[max(j - 1, 0):j + 3].tolist())
            print("  split:", y[max(j - 1, 0):j + 3].tolist())
            # is the chain's extra row suppressed by an earlier split row of the same class?
            sx = {tuple(r) for r in x.tolist()}
            sy = {tuple(r) for r in y.tolist()}
            print("  only in chain:", len(sx - sy), " only in split:", len(sy - sx))
            for r in sorted(sy - sx, key=lambda r: -r[4])[:3]:
                cls = r[5]
                same = [q for q in x.tolist() if q[5] == cls and q[4] > r[4]]
                def iou(p, q):
                    w = max(0.0, min(p[2], q[2]) - max(p[0], q[0])); h = max(0.0, min(p[3], q[3]) - max(p[1], q[1]))
                    inter = w * h
                    return inter / ((p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - inter)
                print("   missing row", r, "max IoU with the chain's earlier rows of its class:", max([iou(r, q) for q in same], default=0.0),
                      "rows of that class in chain / split:", sum(q[5] == cls for q in x.tolist()), sum(q[5] == cls for q in y.tolist()))
            break
    else:
        print("identical")


if __name__ == "__main__":
    main()
