#!/usr/bin/env python3
"""Prints the hand-written kernels of a rocprofv3 --kernel-trace --stats run:  python tools/kernel_stats.py <output dir>"""
import csv
import glob
import re
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Name"])
    if m:
        print("%-70s %6s avg %9.1f us  min %8.1f" % (m.group(0)[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
