#!/usr/bin/env python3
"""Detection-head hot-path benchmark: decode + target assignment + CIoU loss fwd/bwd + batched NMS.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic YOLOv5s head outputs that are
already resident in HBM: bs 64 per GPU @ 640 px, 3 levels, 80 classes (25 200 anchors / image),
640 targets.  Weak scaling: every rank owns its own 64 images (the path shards by image); the only
collective is one all-reduce per step of the loss kernels' per-level sums and divisors (15 doubles), from which the loss
terms of the whole batch are formed.
Rank 0 prints ONE JSON line; `value` = images/s over all ranks.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STRIDES = (8, 16, 32)
ANCHORS_PX = ((10, 13, 16, 30, 33, 23), (30, 61, 62, 45, 59, 119), (116, 90, 156, 198, 373, 326))
HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0, label_smoothing=0.0)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (≈6.3 TB/s is what a plain copy kernel reaches)


class _Det:
    def __init__(self, anchors, nc):
        self.anchors, self.nc = anchors, nc
        self.nl, self.na = anchors.shape[0], anchors.shape[1]
        self.stride = torch.tensor([float(s) for s in STRIDES])


class _Model(torch.nn.Module):
    def __init__(self, anchors, nc):
        super().__init__()
        self.model = [_Det(anchors, nc)]
        self.hyp = dict(HYP)


def grid_anchors(device):
    a = torch.tensor(ANCHORS_PX, dtype=torch.float32).view(3, 3, 2)
    return (a / torch.tensor(STRIDES, dtype=torch.float32).view(3, 1, 1)).to(device)


def synth_batch(bs, img, nc, nt, seed, device):
    """Raw head maps with the bias statistics of Detect._initialize_biases
    (object_detection/yolov5/models/yolo.py:196-204) plus targets; a few anchors per target get
    confident objectness / class logits so that NMS has real clusters to resolve."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.zeros(nt, 6)
    t[:, 0] = torch.sort(torch.randint(0, bs, (nt,), generator=gen).float())[0]
    t[:, 1] = torch.randint(0, nc, (nt,), generator=gen).float()
    t[:, 2:4] = torch.rand(nt, 2, generator=gen)
    t[:, 4:6] = torch.exp(torch.rand(nt, 2, generator=gen) * (math.log(0.6) - math.log(0.01)) + math.log(0.01))
    dgen = torch.Generator(device=device).manual_seed(seed)
    levels = []
    for s in STRIDES:
        n = img // s
        x = torch.randn((bs, 3, n, n, nc + 5), generator=dgen, device=device)
        x[..., 4] += math.log(8 / (640 / s) ** 2)
        x[..., 5:] += math.log(0.6 / (nc - 0.999999))
        levels.append(x)
    return levels, t.to(device)


def plant_objects(levels, targets, crit):
    """Raise obj / class logits on the cells the assignment selects, like a partly trained head."""
    tcls, _, indices, _ = crit.build_targets(levels, targets)
    gen = torch.Generator(device=levels[0].device).manual_seed(77)
    for l, x in enumerate(levels):
        b, a, gj, gi = indices[l]
        if b.numel():
            # one row per cell (several targets can share a cell: an indexed store with repeated indices keeps an
            # arbitrary one), so that every run plants the same values
            flat = ((b * x.shape[1] + a) * x.shape[2] + gj) * x.shape[3] + gi
            order = torch.argsort(flat, stable=True)
            keep = torch.ones_like(order, dtype=torch.bool)
            keep[1:] = flat[order][1:] != flat[order][:-1]
            sel = order[keep]
            bb, aa, jj, ii, cc = b[sel], a[sel], gj[sel], gi[sel], tcls[l][sel]
            x[bb, aa, jj, ii, 4] = 1.5 + torch.randn(sel.numel(), generator=gen, device=x.device)
            x[bb, aa, jj, ii, 5 + cc] = 2.0 + torch.randn(sel.numel(), generator=gen, device=x.device)
            x[bb, aa, jj, ii, 0:4] *= 0.3
    return sum(int(i[0].numel()) for i in indices)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_config(bs, nt, seconds_budget, val_nms=True):
    """One configuration of the workload on the CPU oracle, stage by stage: {stage: seconds per call}, reps."""
    from oracle import nms as onms
    from oracle import yolo as oy
    img, nc = 640, 80
    torch.manual_seed(0)
    levels, targets = synth_batch(bs, img, nc, nt, 4242, "cpu")
    anchors = oy.grid_anchors()
    crit = oy.YoloLossOracle(anchors, dict(HYP), nc=nc)
    # plant objects with the oracle's own assignment so the NMS sees the same kind of clusters
    for l, t in enumerate(crit.targets_for(levels, targets)):
        b, a, gj, gi = (torch.from_numpy(t[k]) for k in ("b", "a", "gj", "gi"))
        if b.numel():
            levels[l][b, a, gj, gi, 4] = 1.5 + torch.randn(b.numel())
            levels[l][b, a, gj, gi, 5 + torch.from_numpy(t["tcls"])] = 2.0 + torch.randn(b.numel())
            levels[l][b, a, gj, gi, 0:4] *= 0.3
    state = {}

    def decode():
        state["z"] = oy.detect_decode([x.detach() for x in levels], anchors, [float(s) for s in STRIDES])

    def assign():
        crit.targets_for(levels, targets)

    def loss_fwd_bwd():      # includes the assignment, as ComputeLoss.__call__ does (loss.py:120)
        p = [x.detach().requires_grad_(True) for x in levels]
        loss, _ = crit(p, targets)
        torch.autograd.grad(loss, p)

    def nms_infer():
        onms.non_max_suppression(state["z"], 0.25, 0.45, max_det=300)

    def nms_val():            # ~20 000 candidates per image: seconds per image on a CPU, so two images stand for the batch
        onms.non_max_suppression(state["z"][:2], 0.001, 0.6, multi_label=True, max_det=300)

    stages = (("decode", decode), ("assign", assign), ("loss_fwd_bwd", loss_fwd_bwd), ("nms_conf0.25_iou0.45", nms_infer))
    spent = {k: 0.0 for k, _ in stages}
    for _ in range(2):        # warm-up: first calls pay for allocator and thread-pool start
        for _, fn in stages:
            fn()
    t_start, reps = time.perf_counter(), 0
    while reps < 2 or (time.perf_counter() - t_start < seconds_budget and reps < 500):
        for k, fn in stages:
            t0 = time.perf_counter()
            fn()
            spent[k] += time.perf_counter() - t0
        reps += 1
    per = {k: v / reps for k, v in spent.items()}
    if val_nms:
        t0 = time.perf_counter()
        nms_val()
        per["nms_conf0.001_iou0.6_multilabel_per_image"] = (time.perf_counter() - t0) / 2
    return per, reps


CPU_CONFIGS = {"cfg1_bs8": (8, 80, 0.25), "cfg2_bs64": (64, 640, 0.4)}      # batch, targets, share of the time budget


def cpu_baseline(seconds_budget=12.0, configs=("cfg1_bs8", "cfg2_bs64"), val_nms=True):
    """The CPU oracle (restatement of the reference's PyTorch path, pinned to it by the golden fixtures) on bounded samples
    of the same workload, stage by stage: configuration 1 (bs 8, the reference's CPU-runnable case) and configuration 2
    (bs 64, the one the metric is quoted on).  `value` is the last configuration's images/s over the stages of the benchmark step
    (decode + loss forward/backward incl. assignment + NMS at conf 0.25 / IoU 0.45)."""
    out = {}
    for name in configs:
        bs, nt, share = CPU_CONFIGS[name]
        per, reps = _cpu_config(bs, nt, seconds_budget * share, val_nms)
        step = per["decode"] + per["loss_fwd_bwd"] + per["nms_conf0.25_iou0.45"]
        out[name] = {"images_per_s": round(bs / step, 2), "ms_per_step": round(step * 1e3, 2), "reps": reps,
                     "stage_ms": {k: round(v * 1e3, 3) for k, v in per.items()}}
    head, (hbs, hnt, _) = out[configs[-1]], CPU_CONFIGS[configs[-1]]
    return {"value": head["images_per_s"], "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "os_cpu_count": os.cpu_count(), "torch_num_threads": torch.get_num_threads(), "cpu_model": _cpu_model(),
            "sample": f"bs {hbs} @ 640px, {hnt} targets, {head['reps']} reps of decode + assign + loss fwd/bwd + NMS(0.25, 0.45) on the CPU "
                      f"oracle (PyTorch-CPU restatement with torch's intra-op threads; C greedy NMS, one image after the other as "
                      f"in the reference), {head['ms_per_step']:.1f} ms/step; stage_ms.assign is part of loss_fwd_bwd; the "
                      f"validation-threshold NMS (conf 0.001, multi-label) is timed beside the step on two images, once",
            "configs": out}


PMC_FILES = ("r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json")      # newest first


def load_pmc_traffic():
    """The committed counter summary of this command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes and the
    --kernel-trace --stats averages): (contents, path relative to the repository)."""
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f), "profiles/" + name
    return {}, None


def spawn_ranks(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, RCCL backend), wait
    for them and exit with their status.  This process never touches the GPU (counting devices does not initialise
    it), so nothing that has is ever replaced or forked."""
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("DH_BENCH_SHARE_DEVICE"):
        raise SystemExit(f"{os.path.basename(script or __file__)} --gpus {n} needs {n} devices, this machine shows {have}")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(script or __file__)] + argv, env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in pending:      # one rank failed: the others would wait in a collective forever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)     # a third of a second: box-to-box spread of a 200-step run was 5 %
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0,
                    help="time budget of the CPU baseline's repeated stages (the validation-threshold NMS sample comes on top)")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2, 3),
                    help="1: everything on one stream; 2: the NMS chain on a second stream; 3: NMS chain and loss on their own streams")
    ap.add_argument("--early-positives", action="store_true",
                    help="target assignment and the positives' forward on a side stream beside the decode (they read only "
                         "the targets and the raw maps); measured slower than keeping them on the main stream")
    ap.add_argument("--nms-first", action="store_true",
                    help="enqueue the NMS chain (side stream) before the loss (main stream) instead of after it, so that its "
                         "candidate search can start right behind the decode, beside the loss's short kernels")
    ap.add_argument("--autograd", action="store_true",
                    help="call loss = crit(p, t); autograd.grad(loss, p) instead of crit.forward_backward(p, t): same kernels, "
                         "plus the autograd engine's hop to its device thread and back (two thread wake-ups per step)")
    ap.add_argument("--pre-roll", type=int, default=2000,
                    help="untimed steps before the --warmup steps (reported as `pre_roll_steps`): a GPU that has idled through "
                         "process start-up and the CPU baseline needs a few hundred milliseconds of work before its clocks and "
                         "the allocator's pools are in the state a running loop sees; 0 disables")
    ap.add_argument("--post-roll", type=float, default=2.0,
                    help="seconds of the same steps after the timed region, reported as `sustained` (keeps the GPU busy "
                         "long enough for an outside monitor to see the run; 0 disables)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            spawn_ranks(args.gpus, sys.argv[1:])      # does not return
    elif int(os.environ["WORLD_SIZE"]) != args.gpus and not os.environ.get("DH_BENCH_FORCE_DIST"):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}")

    if os.environ.get("DH_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DH_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    # rehearsal knob for a one-GPU box: run the process group and the per-step collective with a single rank, so that the
    # RCCL code path itself (init, all_reduce on the side stream, barrier, max over ranks) is exercised
    dist_on = world > 1 or bool(os.environ.get("DH_BENCH_FORCE_DIST"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    # rehearsal knobs for a one-GPU box: all ranks on one device and a host-side backend
    # (RCCL refuses two ranks on the same GPU); never set by the real multi-GPU launch
    if os.environ.get("DH_BENCH_SHARE_DEVICE"):
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} of {world} needs device {local_rank}, this machine shows "
                         f"{torch.cuda.device_count()} (--gpus {world} needs {world} devices)")
    # the CPU baseline runs first (rank 0 of a single-GPU run), so that the GPU work is the last thing the process does
    base = cpu_baseline(seconds_budget=args.cpu_baseline_seconds) if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None
    backend = os.environ.get("DH_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when its communicator is created: keep stdout for the one JSON line by
        # pointing file descriptor 1 at stderr until the communicator exists (a first collective forces that)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)      # "nccl" is RCCL on ROCm
                warm = torch.zeros(1, device=device)
            else:
                dist.init_process_group(backend)
                warm = torch.zeros(1)
            dist.all_reduce(warm)
            if backend == "nccl":
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    from object_detection_amd.yolo import detect_decode
    from object_detection_amd.yolo_loss import ComputeLoss

    bs, img, nc, nt = args.batch, 640, 80, 10 * args.batch
    anchors = grid_anchors(device)
    crit = ComputeLoss(_Model(anchors, nc))
    levels, targets = synth_batch(bs, img, nc, nt, 1000 + rank, device)
    n_planted = plant_objects(levels, targets, crit)
    strides = torch.tensor([float(s) for s in STRIDES])
    A = sum(3 * (img // s) ** 2 for s in STRIDES)
    p = [x.requires_grad_(True) for x in levels]

    from object_detection_amd.general import nms_launch, nms_meta_to_host, nms_views, non_max_suppression
    stages = ("decode", "assign+loss_fwd+bwd", "nms", "d2h")
    ev = {s: [] for s in stages}
    cap = A

    def f_decode():
        with torch.no_grad():
            return detect_decode(levels, anchors, strides)

    def f_loss():
        if not args.autograd:      # the fused forward+backward entry: five launches, no autograd-engine thread hop
            loss, items, grads = crit.forward_backward(p, targets)
        else:
            loss, items = crit(p, targets)
            grads = torch.autograd.grad(loss, p)
        # (nl, 5) float64 sums and divisors, written by the loss kernel; the view is only built when it travels
        return loss, items, grads, (crit.last_sums if dist_on else None)

    host_s = {s: 0.0 for s in stages + ("finish",)}
    # Three HIP streams: decode on the main stream; the NMS chain (short, latency-bound kernels that read the decoded
    # tensor) on a second; assignment + loss forward/backward (bandwidth-bound, reads the raw maps and the logit plane
    # the decode left behind) on a third.  Both wait for the step's decode; the next step's decode does not wait for them.
    n_streams = args.streams
    overlap = n_streams > 1
    torch.cuda.synchronize()
    # The main stream carries the step's critical chain and is created with high priority (-1), the NMS stream with the
    # default: measured 202.5 k against 201.0 k images/s (the dense gradient pass loses less to the NMS chain beside it).
    if overlap:      # keep the legacy null stream (which the runtime orders against every other stream) out of the loop
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ.get("DH_BENCH_MAIN_PRIORITY", "-1"))))
    main_stream = torch.cuda.current_stream()
    nms_stream = torch.cuda.Stream(priority=int(os.environ.get("DH_BENCH_NMS_PRIORITY", "0"))) if overlap else main_stream
    loss_stream = torch.cuda.Stream() if n_streams == 3 else main_stream
    early = overlap and not args.autograd and args.early_positives
    nms_first = overlap and args.nms_first
    # diagnostic: how fast the main stream runs with no NMS chain beside it (the line is then marked invalid)
    skip_nms, skipped, first_call = bool(os.environ.get("DH_BENCH_SKIP_NMS")), {}, [True]
    pos_stream = torch.cuda.Stream(priority=int(os.environ.get("DH_BENCH_POS_PRIORITY", "0"))) if early else None
    set_stream = torch.cuda.set_stream
    split = {"prev_stats": None, "reduced": None, "local_of_reduced": None}

    def reduce_stats(mine):
        """all_reduce(SUM) of one step's per-level [sum(1-iou), sum cls, sum obj, rows, cells] on the current (side) stream."""
        if overlap:
            mine.record_stream(torch.cuda.current_stream())      # made on the main stream, read on this one
        mine = mine.clone() if backend == "nccl" else mine.cpu()   # (host tensor only in the gloo rehearsal)
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)               # over RCCL
        return mine

    def launch(record, detail=False):
        """Enqueue one step's GPU work and the async copy of the NMS bookkeeping; no host synchronisation.
        record: a timed step; detail: bracket the stages of this step with timing events (every 4th timed step: each
        event is a marker packet that the next kernel of the stream waits behind -- the kernel trace shows ~5 us per
        event on the critical stream -- and costs host time, too).  The other steps carry the one event the NMS stream
        needs to wait for the decode.
        The loss is enqueued before the NMS chain: it continues the critical (main) stream, the NMS chain only has to
        be done by the time its step is finished one iteration later."""
        marks = mark_pool.pop() if record else None
        detail = detail and record
        h0 = time.perf_counter()
        positives = pos_done = None
        if early:      # assignment + positives' forward beside the decode: short kernels that read ~2 MB of the raw maps
            set_stream(pos_stream)
            positives = crit.positives(p, targets)
            pos_done = torch.cuda.Event()
            pos_done.record()
            set_stream(main_stream)
        if detail: marks[0].record()
        z = f_decode()
        if detail: marks[1].record()
        h1 = time.perf_counter()
        if overlap:
            decoded = torch.cuda.Event()
            decoded.record()
            if n_streams == 3:
                loss_stream.wait_event(decoded)
                set_stream(loss_stream)
        def enqueue_loss():
            if detail: marks[6].record()
            if early:
                torch.cuda.current_stream().wait_event(pos_done)
                out = crit.forward_backward(p, targets, positives=positives) + ((crit.last_sums if dist_on else None),)
            else:
                out = f_loss()
            if detail: marks[2].record()
            return out

        def enqueue_nms():
            if overlap:
                nms_stream.wait_event(decoded)
                set_stream(nms_stream)
            if detail: marks[3].record()
            if skip_nms and not first_call[0]:      # diagnostic only (DH_BENCH_SKIP_NMS): reuse the first step's NMS results
                rows, meta = skipped["rows"], skipped["meta"]
            else:
                rows, meta, _ = nms_launch(z, 0.25, 0.45, max_det=300, cap=cap)
                skipped["rows"], skipped["meta"], first_call[0] = rows, meta, False
            if detail: marks[4].record()
            host_meta, copied = nms_meta_to_host(meta)           # the step's only device-to-host copy (4*bs ints)
            if detail: marks[5].record()
            return rows, host_meta, copied

        if nms_first:      # the NMS chain is handed to the GPU before the loss: see --nms-first
            rows, host_meta, copied = enqueue_nms()
            set_stream(loss_stream if n_streams == 3 else main_stream)
            h2 = time.perf_counter()
            loss, items, grads, stats = enqueue_loss()
            if overlap:
                set_stream(nms_stream)      # the collective below rides on the side stream
        else:
            loss, items, grads, stats = enqueue_loss()
            h2 = time.perf_counter()
            rows, host_meta, copied = enqueue_nms()
        if dist_on:
            # The collective runs on the side stream, so the next decode never waits for the other ranks.  With two
            # streams it reduces the PREVIOUS step's loss statistics: the side stream has just waited for this step's
            # decode, which follows that loss on the main stream, so no event of its own is needed on the critical
            # stream (each one costs ~5 us there); one collective per step all the same.
            mine = split["prev_stats"] if overlap else stats
            split["prev_stats"] = stats if overlap else None
            if mine is not None:
                split["local_of_reduced"] = mine
                stats = split["reduced"] = reduce_stats(mine)
        if overlap:
            set_stream(main_stream)
        h3 = time.perf_counter()
        if record:
            for s, d in zip(stages, (h1 - h0, h2 - h1, h3 - h2, 0.0)):
                host_s[s] += d
        return dict(marks=marks, z=z, rows=rows, host_meta=host_meta, copied=copied, loss=loss, grads=grads, stats=stats)

    def finish(job, record=False):
        """Host half of a step: wait for that step's copy only, then build the per-image detection views."""
        job["copied"].synchronize()
        h0 = time.perf_counter()
        dets, _ = nms_views(job["rows"], job["host_meta"].tolist(), cap)
        if dets is None:      # the fast first attempt was not enough for some image (never with this workload): blocking repeat
            dets = non_max_suppression(job["z"], 0.25, 0.45, max_det=300)
        if record:
            host_s["finish"] += time.perf_counter() - h0
        return dets

    step_wall = []

    SAMPLE_EVERY = 4      # timed steps between two that carry the stage-timing events

    def run(n, record):
        """n steps, software-pipelined: the host half of step i runs while the GPU works on step i+1."""
        all_marks, prev, dets = [], None, None
        for i in range(n):
            if record:
                step_wall.append(time.perf_counter())
            job = launch(record, i % SAMPLE_EVERY == 0)
            if prev is not None:
                dets = finish(prev, record)      # prev's tensors are released here, so the allocator reuses its blocks
            prev = job
            all_marks.append((job["marks"], i % SAMPLE_EVERY == 0))
        if dist_on and overlap and split["prev_stats"] is not None:
            # the collective of step i rides with step i+1 (see launch); the last step's goes out here, inside the timed
            # region: n steps, n collectives, every step's statistics reduced exactly once
            set_stream(nms_stream)
            nms_stream.wait_stream(main_stream)
            split["reduced"] = reduce_stats(split["prev_stats"])
            split["local_of_reduced"] = split["prev_stats"]
            split["prev_stats"] = None
            set_stream(main_stream)
        if prev is not None:
            dets = finish(prev, record)
        return all_marks, dets

    def sync_all():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # Everything the timed region needs from the host is prepared BEFORE the warm-up, so that nothing but the barrier and the
    # synchronisation the contract asks for sits between the last warm-up step and the first timed one: a GPU left idle for
    # the tens of milliseconds that creating the events and a garbage collection take drops its clocks, and a short timed
    # region (20 steps are 7 ms) would then be measured on the ramp back up (decode launches of 0.24 ms instead of 0.19 ms).
    # Timing events are created here (and recorded once, which is what allocates them).
    mark_pool = [[torch.cuda.Event(enable_timing=True) for _ in range(7)] for _ in range(args.steps)]
    for marks in mark_pool:
        for m in marks:
            m.record()
    import gc
    gc.collect()
    if not os.environ.get("DH_BENCH_KEEP_GC"):
        gc.disable()      # a serving loop does not want a cyclic-GC pause in the middle of a step
    if args.pre_roll > 0:
        run(args.pre_roll, False)
    run(args.warmup, False)
    sync_all()
    t0 = time.perf_counter()
    all_marks, dets = run(args.steps, True)
    sync_all()
    elapsed = time.perf_counter() - t0
    gc.enable()
    for marks, detail in all_marks:     # decode and loss are bracketed on the main stream, nms and the copy on the stream they ran on
        for s, (a, b) in zip(stages, ((0, 1), (6, 2), (3, 4), (4, 5))):
            if detail:
                ev[s].append(marks[a].elapsed_time(marks[b]))
    per_rank_ms = [elapsed / args.steps * 1e3]
    collective_check = None
    if dist_on:
        cdev = device if backend == "nccl" else "cpu"
        mine_t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(every, mine_t)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        elapsed = max(float(t.item()) for t in every)          # the job is as slow as its slowest rank
        # the last step's reduced vector against the sum of what each rank put in (gathered outside the timed region)
        local = split["local_of_reduced"].detach().to(cdev, torch.float64).contiguous()
        parts = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        want = torch.stack(parts).sum(0)
        got = split["reduced"].to(cdev)
        items_g = crit.items_from_sums(got)          # loss terms of the whole batch (sums over summed divisors)
        items_l = crit.items_from_sums(local)
        collective_check = {"reduced": [round(float(v), 6) for v in got.flatten().tolist()],
                            "sum_over_ranks": [round(float(v), 6) for v in want.flatten().tolist()],
                            "layout": "per level [sum(1-iou), sum class BCE, sum objectness BCE, assigned rows, cells]",
                            "global_lbox_lobj_lcls_npos": [round(float(v), 6) for v in items_g.tolist()],
                            "this_rank_lbox_lobj_lcls_npos": [round(float(v), 6) for v in items_l.tolist()],
                            "ok": bool(torch.allclose(got, want, rtol=1e-9, atol=0.0))}

    # Untimed post-roll: the same steps for a couple of seconds.  The timed region above is a few dozen milliseconds,
    # which a once-per-second utilisation monitor never sees; this one it does, and its rate is reported beside `value`.
    sustained = None
    if args.post_roll > 0:
        chunk = max(args.steps, 50)
        sync_all()
        p0, done_steps = time.perf_counter(), 0
        while True:
            run(chunk, False)
            done_steps += chunk
            go_on = torch.tensor([1.0 if time.perf_counter() - p0 < args.post_roll else 0.0],
                                 device=device if (dist_on and backend == "nccl") else "cpu")
            if dist_on:
                dist.all_reduce(go_on, op=dist.ReduceOp.MIN)      # every rank leaves the loop after the same chunk
            if float(go_on.item()) == 0.0:
                break
        sync_all()
        psec = time.perf_counter() - p0
        sustained = {"steps": done_steps, "seconds": round(psec, 3), "ms_per_step": round(psec / done_steps * 1e3, 4),
                     "images_per_s": round(bs * world * done_steps / psec, 1),
                     "note": "untimed post-roll of the same steps, this rank's clock, includes one host sync per chunk"}

    ms_per_step = elapsed / args.steps * 1e3
    value = bs * world * args.steps / elapsed
    stage_ms = {s: float(np.mean(v)) for s, v in ev.items()}

    # roofline of the dominant streaming kernel: the decode launch, 340 B read + 340 B written per anchor
    decode_bytes = 680.0 * A * bs
    decode_s = stage_ms["decode"] * 1e-3
    achieved = decode_bytes / decode_s / 1e9
    pmc, pmc_path = load_pmc_traffic()
    decode_kernel = pmc.get("decode_kernel", "k_yolo_decode_lanes")      # the launch detect_decode makes for fp32 maps
    prof_ms = pmc.get("per_launch", {}).get(decode_kernel, {}).get("avg_us", 0.0) / 1e3
    roofline = {"kernel": decode_kernel, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "measured": "HIP events around the launch on its stream, this run (every 4th timed step)",
                # NOT measured in this run: the PMC counters need rocprofv3; the figure is read from the committed summary
                "traffic": pmc.get("k_yolo_decode_bytes_per_launch"), "traffic_source": pmc_path,
                "algorithmic_bytes_per_launch": decode_bytes, "avg_launch_ms": round(stage_ms["decode"], 4),
                # the same kernel's average in the committed rocprofv3 --kernel-trace --stats run of this command (the
                # event bracket above adds the two marker packets and the dispatch latency to it), and the fraction it gives
                "rocprof_avg_launch_ms": round(prof_ms, 4) or None, "rocprof_source": pmc_path if prof_ms else None,
                "frac_rocprof": round(decode_bytes / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if prof_ms else None}

    # the measured ceiling beside the nominal one: a plain device-to-device copy moving the same bytes as one decode
    # launch (the same read + write mix), timed alone after the timed region
    if rank == 0:
        src_c = torch.empty(int(decode_bytes // 8), dtype=torch.float32, device=device)
        dst_c = torch.empty_like(src_c)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            dst_c.copy_(src_c)
        c0.record()
        for _ in range(10):
            dst_c.copy_(src_c)
        c1.record()
        torch.cuda.synchronize()
        roofline["device_copy_gbs"] = round(decode_bytes * 10 / (c0.elapsed_time(c1) * 1e-3) / 1e9, 1)
        # the same bytes through the library's plain streaming copy (one 16-byte load and store per thread): the highest
        # copy rate a kernel reaches on this memory system, i.e. the measured ceiling for the decode launch
        from object_detection_amd import _lib as L
        nbytes = src_c.numel() * 4 // 16 * 16
        best = None
        for _ in range(5):
            c0.record()
            for _ in range(10):
                L.check(L.lib().dh_stream_copy(L.ptr(src_c), L.ptr(dst_c), nbytes, L.stream()), "dh_stream_copy")
            c1.record()
            torch.cuda.synchronize()
            ms = c0.elapsed_time(c1) / 10
            best = ms if best is None else min(best, ms)
        roofline["stream_copy_gbs"] = round(2.0 * nbytes / (best * 1e-3) / 1e9, 1)
        roofline["stream_copy_ms"] = round(best, 4)
        roofline["frac_of_stream_copy"] = round(achieved / roofline["stream_copy_gbs"], 4)
        del src_c, dst_c

    if rank == 0:
        line = {
            "metric": "images/sec decode+assign+IoU-loss+NMS, bs=64 @640px; HBM GB/s vs peak",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"YOLOv5s head, bs {bs}/GPU @ {img}px, 3 levels, nc {nc}, {A} anchors/img, {nt} targets; "
                                   "decode + build_targets + CIoU/obj/cls loss fwd+bwd + NMS(conf 0.25, iou 0.45, max_det 300)",
                       "global_batch": bs * world, "parallelism": f"image-sharded x{world}",
                       "planted_positive_rows": n_planted, "detections_last_step": int(sum(d.shape[0] for d in dets))},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "host_enqueue_ms": {k: round(v / args.steps * 1e3, 4) for k, v in host_s.items()},
            "first_intervals_ms": [round(float(v) * 1e3, 3) for v in np.diff(step_wall)[:6]] if os.environ.get("DH_BENCH_DEBUG") else None,
            "host_step_interval_ms": {k: round(float(v) * 1e3, 4) for k, v in zip(
                ("p10", "p50", "p90", "max", "argmax"), list(np.percentile(np.diff(step_wall), (10, 50, 90, 100))) +
                [int(np.argmax(np.diff(step_wall))) * 1e-3])} if len(step_wall) > 2 else None,
            "ms_per_step_per_rank": [round(v, 4) for v in per_rank_ms],
            "collective": ("one all_reduce(SUM) per step of the loss kernels' per-level sums and divisors (nl x 5 float64) over RCCL "
                           "on the side stream; step i's block travels while step i+1 is enqueued, the last step's inside the "
                           "timed region") if dist_on else None,
            "collective_check": collective_check,
            "pre_roll_steps": args.pre_roll,
            "sustained": sustained,
            "loss_call": "crit(p,t) + torch.autograd.grad" if args.autograd else (
                "crit.positives(p,t) on a side stream beside the decode, then crit.forward_backward(p,t,positives=...)" if early
                else "crit.forward_backward(p,t)"),
            "streams": {1: "one stream", 2: "decode, assign+loss (high-priority stream) | nms chain on two streams (stage_ms overlap)",
                        3: "decode | nms chain | assign+loss on three streams (stage_ms overlap)"}[n_streams],
            "roofline": roofline,
        }
        if skip_nms:
            line["INVALID"] = "DH_BENCH_SKIP_NMS: the NMS chain ran in the first step only"
        if base is not None:
            line["cpu_baseline"] = base
        print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
