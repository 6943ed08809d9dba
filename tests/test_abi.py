"""CPU-only checks of the C-ABI boundary: the library loads, and exports exactly what include/dethead.h declares."""
import ctypes
import os
import re

import pytest
import torch

from tests.conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dethead.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from object_detection_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libdethead.so has not been built (make -C object_detection_amd/csrc)"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 6
    for n in names:
        assert hasattr(handle, n), f"{n} declared in dethead.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "python binding table and header disagree"
    _lib.lib()


def test_hyp_struct_layout_matches_the_header(tmp_path):
    """The ctypes mirror of dh_yolo_hyp has the size and field offsets the C compiler gives the header's struct."""
    import subprocess
    from object_detection_amd import _lib
    fields = [n for n, _ in _lib.YoloHyp._fields_]
    src = tmp_path / "layout.c"
    prints = "".join(f'printf("{n} %zu\\n", offsetof(dh_yolo_hyp, {n}));' for n in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dethead.h"\n'
                   f'int main(void) {{ printf("size %zu\\n", sizeof(dh_yolo_hyp)); {prints} return 0; }}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(got["size"]) == ctypes.sizeof(_lib.YoloHyp)
    for n in fields:
        assert int(got[n]) == getattr(_lib.YoloHyp, n).offset, n


def test_no_cpu_fallback():
    from object_detection_amd.metrics import bbox_iou
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bbox_iou(torch.zeros(4, 3), torch.zeros(3, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "object_detection_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", ""), f"{f} mentions the oracle"


def test_only_the_checkers_import_the_oracle():
    """Outside tests/ the oracle is imported by bench.py (its cpu_baseline leg) and __graft_entry__.py (smoke, build) only:
    no tool or script may lean on it."""
    import re
    allowed = {os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")}
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "tests", "oracle", "gpurun_out", "__pycache__")]
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith(".py") and path not in allowed:
                assert not re.search(r"^\s*(from|import)\s+oracle\b", open(path).read(), re.M), f"{path} imports the oracle"


def test_bench_cpu_baseline_leg_runs_on_the_oracle():
    """bench.py's cpu_baseline leg is the oracle on a bounded sample; it must work without a GPU."""
    import bench
    out = bench.cpu_baseline(seconds_budget=0.2, configs=("cfg1_bs8",), val_nms=False)      # (the default also times bs 64)
    assert out["kind"] == "port" and out["unit"] == "images/s" and out["value"] > 0 and out["cores"] >= 1
    assert "bs 8" in out["sample"] and out["os_cpu_count"] >= 1 and out["cpu_model"]
    stages = out["configs"]["cfg1_bs8"]["stage_ms"]
    assert set(stages) == {"decode", "assign", "loss_fwd_bwd", "nms_conf0.25_iou0.45"} and all(v > 0 for v in stages.values())


def test_host_side_metric_helpers_match_the_oracle():
    """compute_ap and fitness are host arithmetic by interface (numpy curves in, numbers out)."""
    import numpy as np
    from object_detection_amd.metrics import compute_ap, fitness
    from oracle import ap as oa
    rs = np.random.RandomState(0)
    recall = np.sort(rs.rand(50))
    precision = np.sort(rs.rand(50))[::-1]
    got, want = compute_ap(recall, precision), oa.compute_ap(recall, precision)
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    x = rs.rand(4, 7)
    assert np.array_equal(fitness(x), oa.fitness(x))
