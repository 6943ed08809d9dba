#!/usr/bin/env python3
"""Per kernel AND launch size: calls, average / minimum / maximum duration and the resources of the launch, from the
per-dispatch trace of `rocprofv3 --kernel-trace --output-format csv -d <dir> -- <program>`.

  python tools/kernel_trace_by_grid.py <dir> [out.csv] [substring ...]

rocprofv3's own --stats table averages a kernel over every size a program launched it with; a tool that sweeps sizes needs
the figures per size.  A program may also mark groups of launches (tools/pmc_kernels.py: a k_stream_copy launch whose grid
size is the group's number); the rows are then kept apart per group, with the group's description in the last column.  Only the library's kernels (k_*, dh_k_fill) are listed.  Waves per SIMD by registers = floor(512 / VGPRs
rounded up to 8), at most 8; the LDS column is bytes per workgroup."""
import collections
import csv
import glob
import os
import re
import sys


def name_of(raw):
    m = re.search(r"(?<![A-Za-z0-9_])((?:dh_)?k_[a-z0-9_]+)(<[^>]*>)?", raw)
    if m:
        return m.group(0)
    m = re.search(r"\d+(k_[a-z0-9_]+)I(.*?)EEv", raw)                 # a mangled name (some fp16 instances are printed that way)
    return f"{m.group(1)}[{m.group(2)}]" if m else None


def phases():
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from pmc_kernels import PHASES
        return PHASES
    except Exception:
        return {}


def collect(src):
    rows = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        phase = 0
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
            k = name_of(r["Kernel_Name"])
            if not k:
                continue
            wg = max(int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]), 1)
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
            if k == "k_stream_copy":          # a marker: the launches that follow belong to group `grid`
                phase = grid
                continue
            key = (phase, k, grid, wg, int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"]))
            rows.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return rows


def main():
    src = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2].endswith(".csv") else None
    subs = [a for a in sys.argv[2:] if a != out]
    rows = collect(src)
    legend = phases()
    lines = [("kernel", "workgroups", "threads_per_wg", "calls", "avg_us", "min_us", "max_us", "vgprs", "sgprs", "lds_bytes", "scratch",
              "waves_per_simd_by_vgprs", "group")]
    for (ph, k, grid, wg, vg, sg, lds, scr), us in sorted(rows.items(), key=lambda kv: (kv[0][0], kv[0][1], kv[0][2])):
        if subs and not any(s in k for s in subs):
            continue
        occ = min(8, 512 // max((vg + 7) // 8 * 8, 8))
        lines.append((k, grid, wg, len(us), f"{sum(us) / len(us):.2f}", f"{min(us):.2f}", f"{max(us):.2f}", vg, sg, lds, scr, occ,
                      f"{ph}: {legend.get(ph, '')}" if ph else ""))
    if out:
        with open(out, "w", newline="") as fh:
            csv.writer(fh).writerows(lines)
    for ln in lines:
        print("%-56s %10s %5s %6s %9s %9s %9s %5s %5s %7s %4s %3s  %s" % tuple(str(x)[:90] for x in ln))


if __name__ == "__main__":
    main()
