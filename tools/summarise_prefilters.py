#!/usr/bin/env python3
"""Traffic and duration per launch of the candidate-search kernels at validation thresholds (bs 64 @ 640 px, conf 0.001,
IoU 0.6, multi-label), from three rocprofv3 runs of tools/pmc_val_nms.py and tools/time_configs.py:

  python tools/summarise_prefilters.py gpurun_out/r2p r2   -> profiles/r2_pmc_prefilters.json
Expects <dir>/pf_fetch, <dir>/pf_write (--pmc FETCH_SIZE / WRITE_SIZE, separate passes) and <dir>/tc (--kernel-trace --stats
of tools/time_configs.py val,dense,sweep), as tools/collect_profiles.sh writes them."""
import collections
import csv
import glob
import json
import os
import re
import statistics as st
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WANT = ("k_nms_prefilter",)


def name_of(raw):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", raw)
    if not m:
        m = re.search(r"\d+(k_[a-z0-9_]+)I(.*?)EEv", raw)             # a mangled name (fp16 instances are printed that way)
        return f"{m.group(1)}[{m.group(2)}]" if m else None
    return m.group(0)


def counters(sub, counter):
    d = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = name_of(r["Kernel_Name"])
            if k and k.startswith(WANT) and r["Counter_Name"] == counter:
                d[k].append(float(r["Counter_Value"]))
    return d


fetch, write = counters("pf_fetch", "FETCH_SIZE"), counters("pf_write", "WRITE_SIZE")
dur = {}
for f in glob.glob(os.path.join(src, "tc", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = name_of(r["Kernel_Name"])
        if k and k.startswith(WANT):
            grid = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
            dur.setdefault((k, grid), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE -- python3 tools/pmc_val_nms.py ; the same with --pmc WRITE_SIZE (separate passes); "
               "durations from rocprofv3 --kernel-trace --stats -- python3 tools/time_configs.py val,dense,sweep "
               f"(profiles/{tag}_aux_time_configs_stats.csv), launches with 64 images (grid y = 64) only",
    "units": "bytes per launch, median over the launches. fetch_bytes = FETCH_SIZE KiB x 1024 x 2: on gfx950 FETCH_SIZE counts a "
             "wide coalesced read request at half its size, and these kernels read with 16-byte (stream) or 256-byte-per-wave "
             "(conv) requests; write_bytes = WRITE_SIZE KiB x 1024, exact.",
    "algorithmic_bytes": {"tensor": 64 * 25200 * 85 * 4, "objectness_plane": 64 * 25200 * 4,
                          "note": "bs 64 x 25200 rows x 85 fp32; the decoded-tensor search also reads the 4 B/row plane the decode left"},
    "kernels": {},
}
for k in sorted(fetch):
    e = {"launches_pmc": len(fetch[k]), "fetch_bytes": int(st.median(fetch[k]) * 2048), "write_bytes": int(st.median(write.get(k, [0])) * 1024)}
    us = [v for (kk, grid), vs in dur.items() if kk == k and grid[1] == 64 for v in vs]
    if us:
        e["avg_us"] = round(st.mean(us), 1)
        e["launches_timed"] = len(us)
        e["fetch_TBps"] = round(e["fetch_bytes"] / st.mean(us) * 1e-6, 2)
    out["kernels"][k] = e
# the same kernels in the other configurations of tools/time_configs.py (no counters there): durations only
out["durations_other_grids"] = {f"{k} grid {g[0]}x{g[1]}": {"avg_us": round(st.mean(v), 1), "launches": len(v)}
                                for (k, g), v in sorted(dur.items()) if g[1] != 64}
path = os.path.join(ROOT, "profiles", f"{tag}_pmc_prefilters.json")
json.dump(out, open(path, "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, e)
