"""The benchmark's one JSON line, from a fresh single-GPU process: every field of the contract is there and the
numbers in it agree with each other."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_single_gpu_bench_line_holds_the_contract():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "3", "--pre-roll", "50",
                          "--post-roll", "0", "--cpu-baseline-seconds", "0.5"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]                      # stdout carries the JSON line and nothing else
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 3
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "synthetic" in d["data"]
    assert "bs=64" in d["metric"] and "workload" in d["config"] and "model" not in d["config"]
    # value is images over the timed wall clock
    assert d["value"] == pytest.approx(64 * 1000.0 / d["ms_per_step"], rel=2e-3)
    assert 0.2 < d["ms_per_step"] < 5.0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4) and 0.3 < r["frac"] < 1.0
    # achieved = the decode launch's algorithmic bytes over its measured duration
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=2e-3)
    assert r["algorithmic_bytes_per_launch"] == 680.0 * 25200 * 64
    assert r["stream_copy_gbs"] > r["achieved"] * 0.9               # the kernel sits near the measured copy ceiling
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["unit"] == "images/s" and c["value"] > 0 and c["cores"] >= 1
    assert d["value"] > 50 * c["value"]
