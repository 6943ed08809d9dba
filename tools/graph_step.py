#!/usr/bin/env python3
"""The whole bench step (decode, assignment + loss forward/backward, NMS and the copy of its bookkeeping) captured as ONE
HIP graph and replayed, against the eager step.  Variants:
  seq   every launch on one stream, in the order of the eager single-stream step
  fork  inside the capture the assignment + positives' forward run on a side stream beside the decode, the NMS chain on
        another beside the loss (the graph then has three branches)
Replay modes: one instance back to back on one stream (steps strictly one after the other), or two instances with their own
buffers alternating on two streams (consecutive, independent batches overlap as far as the hardware lets them).
Prints ms per step for each combination and whether every replayed result equals the eager one.
   python tools/graph_step.py [bs] [steps]"""
import faulthandler
import os
import sys
import time

import torch

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def say(*a):
    print(*a, flush=True)


class Step:
    """One captured step with its own static buffers."""

    def __init__(self, variant, crit, levels, targets, anchors, strides, A, bs):
        from object_detection_amd.general import nms_launch
        from object_detection_amd.yolo import detect_decode
        self.pinned = torch.empty(4 * bs, dtype=torch.int32, pin_memory=True)
        self.graph = torch.cuda.CUDAGraph()
        self.cap = A
        s_pos, s_nms = torch.cuda.Stream(), torch.cuda.Stream()
        cap_stream = torch.cuda.Stream()
        cap_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(self.graph, stream=cap_stream):
            cur = torch.cuda.current_stream()
            if variant == "fork":
                s_pos.wait_stream(cur)
                with torch.cuda.stream(s_pos):
                    positives = crit.positives(levels, targets)
                self.z = detect_decode(levels, anchors, strides)
                s_nms.wait_stream(cur)
                with torch.cuda.stream(s_nms):
                    self.rows, self.meta, _ = nms_launch(self.z, 0.25, 0.45, max_det=300, cap=A)
                    self.pinned.copy_(self.meta, non_blocking=True)
                cur.wait_stream(s_pos)
                self.loss, self.items, self.grads = crit.forward_backward(levels, targets, positives=positives)
                cur.wait_stream(s_nms)
            else:
                self.z = detect_decode(levels, anchors, strides)
                self.loss, self.items, self.grads = crit.forward_backward(levels, targets)
                self.rows, self.meta, _ = nms_launch(self.z, 0.25, 0.45, max_det=300, cap=A)
                self.pinned.copy_(self.meta, non_blocking=True)
        self.done = torch.cuda.Event()

    def launch(self, stream):
        with torch.cuda.stream(stream):
            self.graph.replay()
            self.done.record()

    def finish(self):
        from object_detection_amd.general import nms_views
        self.done.synchronize()
        dets, _ = nms_views(self.rows, self.pinned.tolist(), self.cap)
        return dets


def main(bs=64, steps=500):
    dev = torch.device("cuda", 0)
    from object_detection_amd.general import non_max_suppression
    from object_detection_amd.yolo import detect_decode
    from object_detection_amd.yolo_loss import ComputeLoss
    anchors = bench.grid_anchors(dev)
    crit = ComputeLoss(bench._Model(anchors, 80))
    levels, targets = bench.synth_batch(bs, 640, 80, 10 * bs, 1000, dev)
    bench.plant_objects(levels, targets, crit)
    strides = torch.tensor([8., 16., 32.])
    A = sum(3 * (640 // s) ** 2 for s in (8, 16, 32))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):       # eager results (and the warm-up that settles the NMS path hint)
        for _ in range(3):
            z_want = detect_decode(levels, anchors, strides)
            l_want, i_want, g_want = crit.forward_backward(levels, targets)
            d_want = non_max_suppression(z_want, 0.25, 0.45, max_det=300)
    torch.cuda.synchronize()
    say("eager reference done")

    def same(step, dets):
        return (torch.equal(step.z, z_want) and torch.allclose(step.loss, l_want, rtol=1e-6)
                and all(torch.allclose(a, b, rtol=1e-5, atol=1e-9) for a, b in zip(step.grads, g_want))
                and dets is not None and all(torch.equal(a, b) for a, b in zip(dets, d_want)))

    results = {}
    for variant in ("seq", "fork"):
        inst = [Step(variant, crit, levels, targets, anchors, strides, A, bs) for _ in range(2)]
        torch.cuda.synchronize()
        say(f"{variant}: captured two instances")
        for mode, n_inst, n_str in (("one instance, one stream", 1, 1), ("two instances, one stream", 2, 1),
                                    ("two instances, two streams", 2, 2)):
            streams = [torch.cuda.Stream() for _ in range(n_str)] * (n_inst // n_str)
            ok = True
            for timed in (False, True):
                n = steps if timed else 20
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pending = []
                for i in range(n):
                    k = i % n_inst
                    if len(pending) == n_inst:      # the instance about to be replayed: its previous results are consumed first
                        dets = pending.pop(0).finish()
                    inst[k].launch(streams[k])
                    pending.append(inst[k])
                while pending:
                    step = pending.pop(0)
                    dets = step.finish()
                    ok = ok and same(step, dets)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            results[(variant, mode)] = (dt / steps * 1e3, ok)
            say(f"{variant:5s} {mode:28s}: {dt / steps * 1e3:.4f} ms/step  ({bs * steps / dt:,.0f} images/s)  equal to eager: {ok}")
        del inst
    return results


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 500)
