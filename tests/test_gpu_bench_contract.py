"""bench.py prints ONE JSON line with the contract's keys (driver-side parser), in both phases."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "1",
                          "--size", "96"] + list(extra), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_phase1_with_cpu_baseline():
    d = run()
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["metric"] == "train_Mpixels_per_sec" and d["unit"] == "Mpixels/s" and d["n_gpus"] == 1 and d["steps"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert d["value"] > 0 and abs(d["value"] - 96 * 96 * 1e-6 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c and "c_oracle" in c
    assert "workload" in d["config"] and "model" not in d["config"] and "K=59" in d["config"]["workload"]
    # per-kernel table (MFMA- and HBM-class), projection report and the phase-2 sub-record travel in the same line
    rows = d["kernels"]["rows"]
    assert any(x.get("bound") == "mfma" for x in rows) and any(x.get("bound") == "hbm" for x in rows)
    assert all(abs(x["frac"] - x["achieved"] / (r["peak"] if x["bound"] == "mfma" else 8000.0)) < 1e-3 for x in rows if "frac" in x)
    assert len(d["projection"]["true_shape"]) == 3 and d["projection"]["nominal_shape"]["M"] == 262144
    assert d["phase2"]["value"] > 0 and "configs[2]" in d["phase2"]["workload"]


def test_bench_line_phase2():
    d = run("--phase", "seenmask", "--no-cpu-baseline")
    assert KEYS <= set(d) and d["value"] > 0 and "configs[2]" in d["config"]["workload"]
