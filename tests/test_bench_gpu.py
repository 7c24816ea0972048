"""GPU: bench.py keeps its contract (one JSON line, the required keys, roofline + cpu_baseline objects) at N = 1, and the
N > 1 code path (one process per rank, barrier, max over ranks, per-rank reports) runs - exercised here with two ranks that
SHARE the box's single GPU over gloo (BUSCA_BENCH_BACKEND=gloo, a test mode; real runs use RCCL with one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline", "cpu_baseline"}


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_contract_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-variants",
                        "--cpu-seconds", "2", "--latency-samples", "50"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and d["higher_is_better"] is True
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    assert abs(rf["steps_per_launch"] - 20 / 3) < 1e-9                     # 7 + 7 + 6: the roofline counts the steps really processed
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "steps/s"
    assert abs(d["value"] - 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-6


def test_bench_two_ranks_code_path():
    env = dict(os.environ, BUSCA_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "8", "--no-variants",
                        "--cpu-seconds", "0", "--latency-samples", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and len(d["ranks"]) == 2 and {x["rank"] for x in d["ranks"]} == {0, 1}
    assert all(x["busca_version"] >= 1000 and x["steps"] == 16 for x in d["ranks"])
    assert abs(d["value"] - 2 * 16 / (d["ms_per_step"] * 16 / 1e3)) / d["value"] < 1e-6       # whole-job steps / slowest rank's time
    assert "TEST MODE" in d["config"]["parallelism"]
