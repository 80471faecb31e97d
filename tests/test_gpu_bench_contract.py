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


@pytest.mark.parametrize("arch", ["fcn32s", "fcn8s"])
def test_bench_two_ranks_code_path_on_one_gpu(arch):
    """the N > 1 branch of bench.py exactly as the driver launches it (python -m torch.distributed.run ... bench.py --gpus 2),
    both ranks on device 0 over gloo (SZN_TEST_ONE_GPU=1): barrier + max-over-ranks timing, rank-0-only JSON line, whole-job
    value = 2 x the per-rank pixels (fcn8s: the autograd path with engine.allreduce_param_grads)"""
    env = dict(os.environ, SZN_TEST_ONE_GPU="1")
    port = 29900 + os.getpid() % 1000 + (500 if arch == "fcn8s" else 0)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                          "--warmup", "1", "--batch", "1", "--size", "96", "--arch", arch], capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 96 * 96 * 1e-6 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    assert "cpu_baseline" not in d and "roofline" in d


def test_bench_plain_gpus2_relaunches_itself_under_torchrun():
    """`python bench.py --gpus 2` with no launcher around it (how the driver called bench.py at N = 1): bench.py re-executes itself
    under torch.distributed.run with two ranks (here both on device 0 over gloo) and the line carries the `comm` record of N > 1:
    step time with the exchange off / fp32 wire / bf16 wire / sharded optimizer and the per-bucket issue / wait times"""
    env = dict(os.environ, SZN_TEST_ONE_GPU="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1",
                          "--size", "96"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["grad_wire"] == "bf16"
    c = d["comm"]
    assert c["world"] == 2 and set(c["ms_per_step"]) == {"off", "fp32", "bf16", "bf16-sharded"}
    assert all(isinstance(v, float) and v > 0 for v in c["ms_per_step"].values()), c["ms_per_step"]
    assert set(c["exposed_comm_ms"]) == {"fp32", "bf16", "bf16-sharded"}
    b = c["buckets"]["bf16"]
    assert b["steps"] >= 2 and len(b["buckets"]) >= 2 and all(r["wait_ms"] >= 0 and r["issued_at_ms"] >= 0 for r in b["buckets"])
    assert b["buckets"][0]["bucket"] == "fc7" or b["buckets"][0]["issued_at_ms"] <= b["buckets"][-1]["issued_at_ms"]


def test_train_cli_two_ranks_on_one_gpu(fast_tmp):
    """train.py under torchrun, 2 ranks (device 0, gloo): shared log directory, DistributedSampler shards, sharded validation with
    all-reduced histograms -- cfg 4 (20-d pascal embeddings, phase 1 only; the phase-2 reduction is engine.allreduce_param_grads,
    covered in tests/test_ddp_gloo.py)"""
    import glob
    env = dict(os.environ, SZN_TEST_ONE_GPU="1")
    port = 30900 + os.getpid() % 1000
    d = fast_tmp
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "train.py"), "-c", "4", "-ve", "1", "-dir", d,
                          "-n", "ddp", "--synthetic", "4", "64", "64", "--workers", "0"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    logs = glob.glob(os.path.join(d, "logs", "ddp_CFG_4_*"))
    assert len(logs) == 1                                       # ONE log directory for the job
    rows = open(os.path.join(logs[0], "train_log.csv")).read().strip().split("\n")
    assert len(rows) == 1 + 2                                   # 4 images / 2 ranks = 2 iterations per rank, rank 0 logs
    assert len(open(os.path.join(logs[0], "val_log.csv")).read().strip().split("\n")) == 2
    assert os.path.exists(os.path.join(logs[0], "checkpoint"))
