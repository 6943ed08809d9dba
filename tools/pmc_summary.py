#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run:  python tools/pmc_summary.py <output dir> [kernel name filter]"""
import collections
import csv
import glob
import re
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
        if m and (len(sys.argv) < 3 or sys.argv[2] in m.group(0)):
            d[m.group(0)[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print(k, " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(d[k].items())), f"(n={max(len(v) for v in d[k].values())})")
