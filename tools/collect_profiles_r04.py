#!/usr/bin/env python3
"""Copy the judged summaries of a tools/gpu_profiles_r04.sh run (gpurun_out/r04/prof) into profiles/r04_* and fold the PMC traffic into profiles/pmc_traffic.json."""
import json
import os
import shutil

P = "gpurun_out/r04/prof/"
d = json.load(open("profiles/pmc_traffic.json"))
for k, v in json.load(open(P + "pmc_dt_entries.json")).items():
    v["round"] = 4
    d[k] = v
tot = [l for l in open(P + "reid_x3_512_pmc_traffic.txt") if l.startswith("TOTAL")][0].split()
ent = d.setdefault("reid_x3_n512", {"correction": "read = FETCH_SIZE KiB x 2 (gfx950 wide-read undercount, MI355X_MICROARCH.md HBM section); write = WRITE_SIZE KiB as reported",
                                    "source": "tools/pmc_traffic.sh (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, --kernel-trace only), python3 tools/reid_bench.py 512 2 x3; profiles/r04_reid_x3_512_pmc_traffic.txt", "round": 4})
ent.update({"hbm_bytes_per_pass": (float(tot[3]) + float(tot[4])) * 1e6, "read_bytes": float(tot[3]) * 1e6, "write_bytes": float(tot[4]) * 1e6, "kernel_us_per_pass": float(tot[2])})
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
cp = {"bench_steps20.json": "r04_bench_steps20.json", "decision_agreement.json": "r04_decision_agreement.json", "reid_x3_512_timeline.txt": "r04_reid_x3_512_timeline.txt",
      "reid_x3_88_timeline.txt": "r04_reid_x3_88_timeline.txt", "reid_x3_512.stats.txt": "r04_reid_x3_512_kernel_stats.txt", "reid_x3_88.stats.txt": "r04_reid_x3_88_kernel_stats.txt",
      "reid_f16_512.stats.txt": "r04_reid512_kernel_stats.txt", "reid_f16_88.stats.txt": "r04_reid88_kernel_stats.txt", "dt_f32_steps20.stats.txt": "r04_dt_f32_steps20_kernel_stats.txt",
      "dt_f16_inflight16.stats.txt": "r04_dt_f16_inflight16_kernel_stats.txt", "dtl_cfg5_f16.stats.txt": "r04_dtl_cfg5_f16_kernel_stats.txt", "dtl_cfg4_f16.stats.txt": "r04_dtl_cfg4_f16_kernel_stats.txt",
      "dtl_cfg4_f32.stats.txt": "r04_dtl_cfg4_f32_kernel_stats.txt", "reid_x3_512_sq_counters.txt": "r04_reid_x3_512_sq_counters.txt", "dtl_cfg5_sq_counters.txt": "r04_dtl_cfg5_sq_counters.txt",
      "reid_x3_512_pmc_traffic.txt": "r04_reid_x3_512_pmc_traffic.txt", "pmc_dt_entries.json": "r04_pmc_dt_entries.json",
      "dt_f32_steps20/t_kernel_stats.csv": "r04_dt_f32_steps20_rocprof_kernel_stats.csv", "dt_f16_inflight16/t_kernel_stats.csv": "r04_dt_f16_inflight16_rocprof_kernel_stats.csv"}
for a, b in cp.items():
    shutil.copy(P + a, "profiles/" + b)
for a, b in (("gpurun_out/r04/bench_default.json", "profiles/r04_bench_default.json"), ("gpurun_out/r04/gpu_tests.txt", "profiles/r04_gpu_tests.txt")):
    if os.path.exists(a):
        shutil.copy(a, b)
for f in ("profiles/r04_bench_default.json", "profiles/r04_bench_steps20.json"):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(r["value"]), round(r["roofline"]["frac"], 3), "kernel_avg_ms", round(r["roofline"]["kernel_avg_ms"], 3))
    for k in ("full_step", "full_step_x3", "full_step_f32", "full_step_x3_tracker_like_candidates", "full_step_tracker_like_candidates"):
        print("  ", k, round(r[k]["ms_per_step"], 2), round(r[k]["roofline"]["frac"], 3))
    for k, v in r["assoc_e2e"].items():
        if isinstance(v, dict) and "p50_assoc_latency_ms" in v:
            print("  ", k, round(v["p50_assoc_latency_ms"], 2), round(v.get("p50_crop_ms", 0), 2))
    for k, v in r["configs"].items():
        if "roofline" in v:
            print("   cfg", k, round(v["ms_per_step"], 3), round(v["roofline"]["frac"], 3), v["roofline"].get("traffic"))
    for k, v in r.get("variants", {}).items():
        if "roofline" in v:
            print("   var", k, round(v["value"]), round(v["roofline"]["frac"], 3))
