#!/usr/bin/env python3
"""Hardware counters per kernel AND launch size, joined over several `rocprofv3 --pmc ... --output-format csv -d <dir>` passes
of one program (a pass holds few counters, so every group of counters is a run of its own).

  python tools/pmc_by_grid.py <out.json> <dir> [<dir> ...] [--only substring,substring]

Per (kernel, workgroups): the median of every counter over the launches, the launch's resources, and what follows from them:
  fetch_bytes  = FETCH_SIZE KiB x 1024 x 2   (on gfx950 FETCH_SIZE tallies a wide coalesced read request at half its size; a
                 kernel that reads 4-byte words with a stride is over-stated by that factor: both figures are kept)
  write_bytes  = WRITE_SIZE KiB x 1024
  valu_busy    = SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES-normalised SIMD time, see `derive`
SQ_* counters that count cycles are in units of four shader cycles and are summed over all waves / SIMDs of the chip."""
import collections
import csv
import glob
import json
import os
import re
import statistics as st
import sys


def name_of(raw):
    m = re.search(r"(?<![A-Za-z0-9_])((?:dh_)?k_[a-z0-9_]+)(<[^>]*>)?", raw)
    if m:
        return m.group(0)
    m = re.search(r"\d+(k_[a-z0-9_]+)I(.*?)EEv", raw)
    return f"{m.group(1)}[{m.group(2)}]" if m else None


def derive(e):
    c = e["counters"]
    d = {}
    if "FETCH_SIZE" in c:
        d["fetch_bytes"] = int(c["FETCH_SIZE"] * 2048)
        d["fetch_bytes_uncorrected"] = int(c["FETCH_SIZE"] * 1024)
    if "WRITE_SIZE" in c:
        d["write_bytes"] = int(c["WRITE_SIZE"] * 1024)
    us = e.get("avg_us")
    if us and "fetch_bytes" in d:
        d["fetch_TBps"] = round(d["fetch_bytes"] / us * 1e-6, 2)
    if us and "write_bytes" in d:
        d["write_TBps"] = round(d["write_bytes"] / us * 1e-6, 2)
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        w = c["SQ_WAVES"]
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
            if k in c:
                d[k.lower().replace("sq_", "") + "_per_wave"] = round(c[k] / w, 1)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if k in c:
                d[k.lower().replace("sq_", "") + "_share_of_wave_cycles"] = round(c[k] / wc, 3)
    if "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] and "SQ_ACTIVE_INST_VALU" in c:
        # SQ_BUSY_CYCLES: summed over the SQs (one per CU-pair slot of every XCD's shader engines) while any wave is resident;
        # the share of it that the vector ALUs issued in is a lower bound of their utilisation
        d["valu_active_over_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"], 3)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
    return d


def main():
    out_path, args = sys.argv[1], sys.argv[2:]
    only = None
    if "--only" in args:
        i = args.index("--only")
        only = args[i + 1].split(",")
        args = args[:i] + args[i + 2:]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    meta, dur = {}, collections.defaultdict(list)
    for src in args:
        for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
            seen = set()
            phase = 0
            for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"])):
                k = name_of(r["Kernel_Name"])
                if not k:
                    continue
                wg = max(int(r["Workgroup_Size"]), 1)
                if k == "k_stream_copy":      # a marker of tools/pmc_kernels.py: the launches that follow belong to this group
                    phase = int(r["Grid_Size"]) // wg
                    continue
                if only and not any(s in k for s in only):
                    continue
                key = (phase, k, int(r["Grid_Size"]) // wg)
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[key] = dict(threads_per_wg=wg, vgprs=int(r["VGPR_Count"]), sgprs=int(r["SGPR_Count"]), lds_bytes=int(r["LDS_Block_Size"]),
                                 scratch=int(r["Scratch_Size"]))
                if r["Dispatch_Id"] not in seen:                  # one duration per dispatch (a row per counter)
                    seen.add(r["Dispatch_Id"])
                    dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {"note": "medians over the launches of the passes; durations under counter collection (serialised dispatches) are a few "
                   "percent longer than in a plain kernel trace: use the kernel-trace file for timings", "kernels": {}}
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from pmc_kernels import PHASES
    except Exception:
        PHASES = {}
    for key in sorted(vals):
        ph, k, grid = key
        e = dict(meta[key])
        if ph:
            e["group"] = f"{ph}: {PHASES.get(ph, '')}"
        e["workgroups"] = grid
        e["waves_per_simd_by_vgprs"] = min(8, 512 // max((e["vgprs"] + 7) // 8 * 8, 8))
        e["launches"] = max(len(v) for v in vals[key].values())
        e["avg_us"] = round(st.mean(dur[key]), 2)
        e["counters"] = {c: st.median(v) for c, v in sorted(vals[key].items())}
        e["derived"] = derive(e)
        out["kernels"][f"{k} @ {grid} wg" + (f" [group {ph}]" if ph else "")] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for name, e in out["kernels"].items():
        print(name, e["avg_us"], "us", e["derived"])


if __name__ == "__main__":
    main()
