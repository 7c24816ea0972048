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


MAX_LINE = 6000        # the driver keeps about 8 KB of stdout; round 5's 24 KB line did not parse


def _last_json(out, launcher_noise=False):
    """stdout must be exactly ONE line, short enough for the driver to keep whole, and it must parse.  (Under torch.distributed.run the gloo test backend's
    C++ prints its own "[Gloo] Rank ..." lines on the children's stdout: those runs must hold exactly one JSON line.)"""
    lines = out.strip().splitlines()
    if launcher_noise:        # (the two ranks' "[Gloo] Rank ..." prints interleave arbitrarily, empty lines included: everything that is not a JSON object is the launcher's)
        lines = [l for l in lines if l.startswith("{")]
    assert len(lines) == 1 and lines[0].startswith("{"), out[-2000:]
    assert len(lines[0]) <= MAX_LINE, len(lines[0])
    return json.loads(lines[0])


def _check_contract(d, steps, warmup):
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "steps/s"       # the reference's own arithmetic on the primary line
    assert "workload" in d["config"] and "model" not in d["config"] and d["higher_is_better"] is True
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms", "steps_per_launch", "algorithmic_bytes_per_step"} <= set(rf)
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "steps/s" and "sample" in cb
    assert abs(d["value"] - steps / (d["ms_per_step"] * steps / 1e3)) / d["value"] < 1e-4
    return rf


def test_bench_exact_driver_command(tmp_path):
    """The command the driver runs at round end, variants and all: one parseable stdout line of at most MAX_LINE characters carrying the
    contract keys, `roofline`, `cpu_baseline` and the compact `variants`; everything else is in the detail file."""
    det = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--detail", det],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    rf = _check_contract(d, 20, 5)
    assert rf["steps_per_launch"] == 20 and d["config"]["launches_in_timed_region"] == 1
    v = d["variants"]
    assert {"x3", "f16", "x3_whole_rounds", "f32_whole_rounds"} <= set(v)
    for k, e in v.items():
        assert set(e) == {"value", "dtype", "frac"} and e["value"] > 0 and 0 < e["frac"] < 1, (k, e)
    assert v["x3"]["dtype"] == "x3" and v["f16"]["dtype"] == "f16"
    full = json.load(open(det))
    assert full["value"] == pytest.approx(d["value"], rel=1e-4) and "ranks" in full and "assoc_e2e" in full and "hbm_kernels" in full
    assert "frac_of_f32_mfma_peak" not in json.dumps(full)
    assert full["variants"]["x3"]["roofline"]["peak"] == pytest.approx(2500.0 / 3)


def test_bench_contract_single_gpu(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-variants",
                        "--cpu-seconds", "2", "--latency-samples", "50", "--detail", str(tmp_path / "d.json")], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    rf = _check_contract(d, 20, 5)
    assert rf["steps_per_launch"] == 20 and d["config"]["launches_in_timed_region"] == 1     # K steps = ONE launch (whole rounds + the token-split tail)
    full = json.load(open(str(tmp_path / "d.json")))["roofline"]
    assert rf["traffic"] is not None or "no PMC run" in full["traffic_note"]          # never a neighbouring shape's number


def _check_two_ranks(d, full):
    assert d["n_gpus"] == 2 and full["n_gpus"] == 2
    d = full
    assert len(d["ranks"]) == 2 and {x["rank"] for x in d["ranks"]} == {0, 1}
    assert all(x["busca_version"] >= 1000 and x["steps"] == 16 and "gfx950" in x["build"] for x in d["ranks"])
    assert abs(d["value"] - 2 * 16 / (d["ms_per_step"] * 16 / 1e3)) / d["value"] < 1e-6       # whole-job steps / slowest rank's time
    assert "TEST MODE" in d["config"]["parallelism"]


def test_bench_two_ranks_code_path(tmp_path):
    """Launched the way the driver does for N > 1 (python -m torch.distributed.run ... bench.py --gpus 2)."""
    env = dict(os.environ, BUSCA_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "8", "--no-variants",
                        "--cpu-seconds", "0", "--latency-samples", "0", "--split-steps", "0", "--detail", str(tmp_path / "d.json")],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    _check_two_ranks(_last_json(r.stdout, True), json.load(open(str(tmp_path / "d.json"))))


def test_bench_self_launches_ranks_and_splits_cfg5(tmp_path):
    """A plain `python bench.py --gpus 2` starts its own two ranks (child processes; the parent never touches the GPU) and relays ONE
    JSON line.  The cfg5 split leg (512 lost x 64 proposals x d512, tracks of one step split over the ranks, host gather) must
    return logits BIT-IDENTICAL to the one-rank run of the same step: a track's result does not depend on which rank, row tile
    or batch it was computed in."""
    env = dict(os.environ, BUSCA_BENCH_BACKEND="gloo")
    common = ["--steps", "16", "--warmup", "8", "--no-variants", "--cpu-seconds", "0", "--latency-samples", "0", "--split-steps", "3"]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--detail", str(tmp_path / "d2.json")] + common,
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    d2 = _last_json(r2.stdout, True)
    f2 = json.load(open(str(tmp_path / "d2.json")))
    _check_two_ranks(d2, f2)
    assert d2["configs"]["cfg5_split"]["n_gpus"] == 2 and d2["configs"]["cfg5_split"]["value"] > 0
    s2 = f2["configs"]["cfg5_split"]
    assert s2["n_gpus"] == 2 and s2["track_slices"] == [[0, 256], [256, 512]] and s2["value"] > 0 and s2["scaling"] == "strong"
    # one rank: the split leg only runs with the variants enabled, so ask for it through a 1-rank launcher run instead
    env1 = dict(env)
    r1 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--detail", str(tmp_path / "d1.json")] + common,
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env1)
    assert r1.returncode == 0, r1.stderr[-2000:]
    _last_json(r1.stdout, True)
    s1 = json.load(open(str(tmp_path / "d1.json")))["configs"]["cfg5_split"]
    assert s1["n_gpus"] == 1 and s1["track_slices"] == [[0, 512]]
    assert s1["logits_sha256"] == s2["logits_sha256"] and s1["argmax_sha256"] == s2["argmax_sha256"]


def test_bench_refuses_more_ranks_than_gpus():
    """Without the test backend, --gpus N on a box with fewer GPUs must fail loudly - never fall back to fewer ranks."""
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k != "BUSCA_BENCH_BACKEND"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "1", "--no-variants",
                        "--cpu-seconds", "0", "--latency-samples", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
