"""CPU: the ONE stdout line of bench.py (round 5's grew to 24 KB and the driver's parser kept only its tail).  `contract_line` must carry the contract keys, `roofline`,
`cpu_baseline`, compact `variants` / `configs`, stay below MAX_LINE whatever the legs add, and parse."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _result(nvar=6, ncfg=14, blob=0):
    rf = {"bound": "mfma", "achieved": 96.3015123456, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.6122151234, "traffic": 456941000.0, "kernel": "dt_fused_kernel " + "x" * 200,
          "kernel_avg_ms": 1.41733, "kernel_launches_per_call": 1.0, "kernel_ms_per_call": 1.41733, "steps_per_launch": 20.0, "algorithmic_bytes_per_step": 10833800,
          "flops_per_call": 136491171840.0, "traffic_note": "n" * 500, "launch_geometry": {"workgroups": 704}}
    leg = {"value": 34716.123456, "unit": "steps/s", "dtype": "x3", "roofline": dict(rf), "note": "n" * 300}
    return {"metric": "BUSCA association steps/sec", "value": 13216.0123456, "unit": "steps/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.0756657123,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dtype_note": "d" * 400, "p50_latency_ms": 0.231,
            "config": {"workload": "cfgN DT-step: 32 lost x 16 proposals x d256", "lost": 32, "proposals": 16, "d": 256, "seq_len": 11, "parallelism": "p" * 80},
            "roofline": rf, "cpu_baseline": {"value": 98.9, "unit": "steps/s", "cores": 16, "kind": "port", "host_cpus": 256, "sample": "s" * 200},
            "variants": {"v%d" % i: dict(leg) for i in range(nvar)}, "configs": {"c%d" % i: dict(leg, n_gpus=2 if i == 0 else 1) for i in range(ncfg)},
            "full_step": dict(leg), "full_step_f32": {"error": "boom"}, "assoc_e2e": {"blob": "b" * blob}, "hbm_kernels": {"blob": "b" * blob}, "ranks": [{"build": "f" * 200}] * 8}


def test_contract_line_is_short_and_complete(bench):
    line = bench.contract_line(_result(blob=20000), "/somewhere/bench_detail.json")
    assert "\n" not in line and len(line) <= bench.MAX_LINE <= 6000
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms", "steps_per_launch", "algorithmic_bytes_per_step"} <= set(d["roofline"])
    assert "workload" in d["config"] and d["detail"] == "bench_detail.json" and d["vs_baseline"] is None
    assert all(set(v) == {"value", "dtype", "frac"} for v in d["variants"].values())
    assert d["configs"]["c0"]["n_gpus"] == 2 and "n_gpus" not in d["configs"]["c1"] and d["configs"]["full_step"]["dtype"] == "x3"
    assert "assoc_e2e" not in d and "hbm_kernels" not in d and "ranks" not in d and "dtype_note" not in d          # the bulk lives in the detail file
    assert d["value"] == pytest.approx(13216.0123456, rel=1e-5) and d["roofline"]["frac"] == pytest.approx(0.612215, rel=1e-5)


def test_contract_line_sheds_optional_blocks_rather_than_outgrow_the_parser(bench):
    """Whatever later rounds add to the legs: optional blocks are dropped before the line may exceed MAX_LINE; the contract keys never are."""
    big = _result(nvar=60, ncfg=200)
    line = bench.contract_line(big, None)
    assert len(line) <= bench.MAX_LINE
    d = json.loads(line)
    assert "roofline" in d and "cpu_baseline" in d and "configs" not in d and "detail" not in d
    hopeless = _result()
    hopeless["roofline"]["kernel"] = "k" * 7000
    with pytest.raises(RuntimeError, match="contract line"):
        bench.contract_line(hopeless, None)


def test_emit_writes_the_detail_file_and_one_line(bench, tmp_path, capsys):
    res = _result(blob=5000)
    bench.emit(res, str(tmp_path / "detail.json"))
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and json.loads(out)["detail"] == "detail.json"
    full = json.load(open(str(tmp_path / "detail.json")))
    assert full["assoc_e2e"]["blob"] == "b" * 5000 and full["value"] == res["value"]
