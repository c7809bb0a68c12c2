"""bench.py's output contract on a small workload: ONE JSON line with the driver's keys, the `roofline` object of the dominant
kernel and (N=1) a `cpu_baseline` object.  Runs the real script in a child process (it owns the GPU context for its lifetime)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def run_bench(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "bench.py prints exactly one line on stdout"
    return json.loads(lines[0])


def check_common(j):
    for k, t in REQUIRED.items():
        assert k in j, k
        assert isinstance(j[k], t) or (t is float and isinstance(j[k], int)), (k, type(j[k]))
    assert "vs_baseline" in j and j["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert j["higher_is_better"] is True and j["unit"] == "images/sec" and j["data"] == "synthetic" and j["dtype"] == "bf16"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s")  # "valu": the exact Ward update is bound by vector-ALU issue
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert j["value"] > 0 and j["ms_per_step"] > 0


def test_small_job_with_cpu_baseline():
    j = run_bench("--total-images", "2048", "--steps", "2", "--warmup", "1")
    check_common(j)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["metric"].startswith("images/sec")
    assert abs(j["value"] - 2048 * 1e3 / j["ms_per_step"]) / j["value"] < 1e-3  # whole-job rate over the timed steps
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    st = j["stages_ms_last_step"]
    assert st["embed_ms"] > 0 and st["merge_ms"] > 0 and st["dist_ms"] > 0


def test_embed_only_without_cpu_baseline():
    j = run_bench("--embed-only", "--total-images", "1024", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    check_common(j)
    assert "cpu_baseline" not in j or j["cpu_baseline"] is None
    assert j["roofline"]["bound"] == "mfma" and "embed only" in j["config"]["workload"]
