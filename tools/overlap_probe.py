"""Does a 548 MB zero-fill on a second stream hide the loss's latency-bound forward launches (or the other way round)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from object_detection_amd import _lib as L
from object_detection_amd import yolo_loss as YL
from object_detection_amd.yolo import detect_decode
from object_detection_amd.yolo_loss import ComputeLoss

dev = torch.device("cuda", 0)
bs = 64
anchors = bench.grid_anchors(dev)
crit = ComputeLoss(bench._Model(anchors, 80))
levels, targets = bench.synth_batch(bs, 640, 80, 10 * bs, 1000, dev)
bench.plant_objects(levels, targets, crit)
strides = torch.tensor([8., 16., 32.])
grads = [torch.empty_like(x) for x in levels]
hyp = crit._hyp_struct()
shapes = [(x.shape[2], x.shape[3]) for x in levels]
main = torch.cuda.Stream(priority=-1)
side = torch.cuda.Stream(priority=int(os.environ.get("SIDE_PRIO", "0")))
lib = L.lib()


def fill_all():
    for g in grads:
        g.zero_()


def fwd():
    A = YL._assign(shapes, targets, crit.anchors, crit.hyp['anchor_t'])
    return YL._loss_fwd(A, hyp, 80, levels)


def timed(fn, n=300):
    for _ in range(400):      # long warm-up: the runtime's first few hundred launches of a pattern are slower
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


torch.cuda.set_stream(main)
z = detect_decode(levels, anchors, strides)     # leaves the logit plane the forward reads


def seq():
    fill_all()
    fwd()


def par():
    e0 = torch.cuda.Event(); e0.record()
    side.wait_event(e0)
    torch.cuda.set_stream(side)
    fill_all()
    e1 = torch.cuda.Event(); e1.record()
    torch.cuda.set_stream(main)
    fwd()
    main.wait_event(e1)


def par_fill_main():
    e0 = torch.cuda.Event(); e0.record()
    side.wait_event(e0)
    torch.cuda.set_stream(side)
    fwd()
    e1 = torch.cuda.Event(); e1.record()
    torch.cuda.set_stream(main)
    fill_all()
    main.wait_event(e1)


print("fill alone                         %7.1f us" % timed(fill_all))
print("assign + loss forward alone        %7.1f us" % timed(fwd))
print("fill, then forward (one stream)    %7.1f us" % timed(seq))
print("fill on a side stream | forward    %7.1f us" % timed(par))
print("forward on a side stream | fill    %7.1f us" % timed(par_fill_main))
