#!/usr/bin/env python3
"""Per-launch HBM traffic of the head-map kernels from two rocprofv3 counter passes over tools/time_head_maps.py:

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/fetch -- python3 tools/time_head_maps.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/write -- python3 tools/time_head_maps.py
  python tools/pmc_head_maps.py <dir> > profiles/<tag>_pmc_head_maps.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read request at half its size, so
the streaming reads of these kernels are doubled (the same correction as for the decode kernel)."""
import collections
import csv
import glob
import json
import re
import statistics as st
import sys

src = sys.argv[1]


def counters(kind, name):
    d = collections.defaultdict(list)
    f = sorted(glob.glob(f"{src}/{kind}/*/*counter_collection.csv"))[-1]
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
        if m and r["Counter_Name"] == name and ("head_maps" in m.group(1) or "prefilter_conv" in m.group(1) or m.group(1) == "k_yolo_decode"):
            d[m.group(0)].append(float(r["Counter_Value"]))
    return d


fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
out = {"units": "bytes per launch (median over the launches of tools/time_head_maps.py at bs 64 @ 640 px); fetch = FETCH_SIZE KiB x 1024 x 2",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    out["kernels"][k] = {"launches": len(fetch.get(k, [])),
                         "fetch_bytes": round(st.median(fetch[k]) * 1024 * 2) if k in fetch else None,
                         "fetch_bytes_min_max": [round(min(fetch[k]) * 2048), round(max(fetch[k]) * 2048)] if k in fetch else None,
                         "write_bytes": round(st.median(write[k]) * 1024) if k in write else None}
json.dump(out, sys.stdout, indent=1)
print()
