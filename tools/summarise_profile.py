#!/usr/bin/env python3
"""Turn the rocprofv3 output of `bench.py` (kernel-trace stats + the two PMC passes) into the committed summaries.

  python tools/summarise_profile.py gpurun_out/prof r1      -> profiles/r1_bench_kernel_stats.csv, profiles/r1_pmc_traffic.json
Expects <dir>/kt, <dir>/fetch, <dir>/write as written by
  rocprofv3 --kernel-trace --stats --output-format csv -d <dir>/kt    -- python3 bench.py --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/fetch       -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --pre-roll 0 --post-roll 0
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/write       -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --pre-roll 0 --post-roll 0
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import statistics as st
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    return sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)[-1]      # newest run


def short(name):
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else None


def counters(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            d[k].append(float(r["Counter_Value"]))
    return d


stats_out = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv")
shutil.copy(one("kt/*/*kernel_stats.csv"), stats_out)
fetch = counters(one("fetch/*/*counter_collection.csv"), "FETCH_SIZE")
write = counters(one("write/*/*counter_collection.csv"), "WRITE_SIZE")
avg_us = {}
for r in csv.DictReader(open(stats_out)):
    k = short(r["Name"])
    if k:
        calls = avg_us.get(k, (0, 0.0))
        avg_us[k] = (calls[0] + int(r["Calls"]), calls[1] + float(r["TotalDurationNs"]))
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --pre-roll 0 --post-roll 0 ; the "
               "same with --pmc WRITE_SIZE "
               "(separate passes); durations from rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline "
               f"(profiles/{tag}_bench_kernel_stats.csv)",
    "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB. On gfx950 FETCH_SIZE counts a wide coalesced read request at half its "
             "size, so streaming reads are corrected x2; WRITE_SIZE is exact. Strided 4-byte reads fetch one 64 B sector per row "
             "and are not doubled.",
    "per_launch": {},
}
for k in sorted(fetch):
    calls, total = avg_us.get(k, (1, 0.0))
    out["per_launch"][k] = {"FETCH_SIZE_KiB": round(st.mean(fetch[k]), 1), "WRITE_SIZE_KiB": round(st.mean(write.get(k, [0])), 1),
                            "avg_us": round(total / calls / 1e3, 1), "calls": calls}
decode_kernel = "k_yolo_decode_lanes" if "k_yolo_decode_lanes" in out["per_launch"] else "k_yolo_decode"      # whichever form ran
out["decode_kernel"] = decode_kernel
d = out["per_launch"][decode_kernel]
out["k_yolo_decode_bytes_per_launch"] = int(d["FETCH_SIZE_KiB"] * 1024 * 2 + d["WRITE_SIZE_KiB"] * 1024)
out["k_yolo_decode_algorithmic_bytes_per_launch"] = 680 * 25200 * 64
out["k_yolo_decode_note"] = ("680 B/anchor = 85 fp32 read + 85 fp32 written; on top of that the launch writes two 4 B/anchor compact "
                             "planes (decoded objectness for NMS, raw objectness logit for the loss); bs 64 x 25200 anchors")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
for k, v in out["per_launch"].items():
    print(f"{k:20s} {v['avg_us']:8.1f} us  fetch {v['FETCH_SIZE_KiB']:10.1f} KiB  write {v['WRITE_SIZE_KiB']:10.1f} KiB  calls {v['calls']}")
print("decode traffic", out["k_yolo_decode_bytes_per_launch"], "algorithmic", out["k_yolo_decode_algorithmic_bytes_per_launch"])
