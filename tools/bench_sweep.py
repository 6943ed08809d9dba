#!/usr/bin/env python3
"""The inference sweep at 1280 px with validation thresholds (conf 0.001, IoU 0.6, multi-label), sharded by image: every
rank takes its own images (16 per GPU: bs 128 over 8 GPUs) from raw head outputs to detections, with no exchange between
ranks on the data path.  Launch contract and JSON line as bench.py:

  python tools/bench_sweep.py [--gpus N] [--steps K] [--warmup W] [--path fused|conv|two]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/bench_sweep.py --gpus N

--gpus N without a launcher starts the N ranks itself.  The process group (RCCL) only carries the barriers around the timed
region and the max over ranks of its duration.  path: fused = detect_nms on the head maps (default), conv = conv_nms on the
outputs of Detect's convolutions, two = detect_decode followed by non_max_suppression."""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--images", type=int, default=16, help="images per GPU")
    ap.add_argument("--img", type=int, default=1280)
    ap.add_argument("--path", default="fused", choices=("fused", "conv", "two"))
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            bench.spawn_ranks(args.gpus, sys.argv[1:], script=__file__)      # does not return
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench_sweep.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}")
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if not torch.cuda.is_available():
        raise SystemExit("bench_sweep.py needs a ROCm GPU: the product path has no CPU fallback")
    if os.environ.get("DH_BENCH_SHARE_DEVICE"):      # rehearsal on a one-GPU box (see bench.py)
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no device {local_rank} on this machine")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(os.environ.get("DH_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    from object_detection_amd.general import conv_nms, detect_nms, non_max_suppression
    from object_detection_amd.yolo import detect_decode

    gen = torch.Generator(device=dev).manual_seed(5 + rank)       # every rank its own images
    levels = []
    for s in bench.STRIDES:
        n = args.img // s
        x = torch.randn((args.images, 3, n, n, 85), generator=gen, device=dev)
        x[..., 4] += math.log(8 / (args.img / s) ** 2)             # the head's bias statistics, yolo.py:196-204
        x[..., 5:] += math.log(0.6 / (80 - 0.999999))
        levels.append(x)
    anchors = bench.grid_anchors(dev)
    strides = torch.tensor([float(s) for s in bench.STRIDES])
    kw = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)
    if args.path == "conv":
        convs = [x.permute(0, 1, 4, 2, 3).reshape(args.images, 255, x.shape[2], x.shape[3]).contiguous() for x in levels]
        step = lambda: conv_nms(convs, 3, anchors, strides, **kw)
    elif args.path == "two":
        step = lambda: non_max_suppression(detect_decode(levels, anchors, strides), **kw)
    else:
        step = lambda: detect_nms(levels, anchors, strides, **kw)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    fence()
    elapsed = torch.tensor([mine], dtype=torch.float64, device=dev)
    per_rank = [mine]
    if world > 1:
        gathered = [torch.zeros_like(elapsed) for _ in range(world)]
        dist.all_gather(gathered, elapsed)
        per_rank = [float(g) for g in gathered]
    slowest = max(per_rank)
    if rank == 0:
        anchors_per_image = sum(3 * (args.img // s) ** 2 for s in bench.STRIDES)
        print(json.dumps({
            "metric": f"images/sec raw head outputs -> detections, {args.img}px, conf 0.001 / IoU 0.6 / multi-label", "unit": "images/s",
            "value": round(world * args.images * args.steps / slowest, 1), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(slowest / args.steps * 1e3, 4), "ms_per_step_per_rank": [round(t / args.steps * 1e3, 4) for t in per_rank],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"inference sweep: {args.images} images/GPU @ {args.img}px ({anchors_per_image} anchors/img, nc 80), "
                                   f"path {args.path}", "global_batch": world * args.images, "parallelism": f"image-sharded x{world}",
                       "detections_last_step": int(sum(o.shape[0] for o in out))},
            "collective": None if world == 1 else "none on the data path; barriers and the gather of the per-rank durations only"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
