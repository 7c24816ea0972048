#!/usr/bin/env python3
"""Copy the judged summaries of a tools/gpu_profiles_r06.sh run (gpurun_out/r06/prof) into profiles/r06_* and fold the PMC traffic into profiles/pmc_traffic.json."""
import json
import os
import shutil

P = "gpurun_out/r06/prof/"
d = json.load(open("profiles/pmc_traffic.json"))
if os.path.exists(P + "reid_x3_512_pmc_traffic.txt"):
    tot = [l for l in open(P + "reid_x3_512_pmc_traffic.txt") if l.startswith("TOTAL")][0].split()
    ent = d.setdefault("reid_x3_n512", {})
    ent.update({"correction": "read = FETCH_SIZE KiB x 2 (gfx950 wide-read undercount, MI355X_MICROARCH.md HBM section); write = WRITE_SIZE KiB as reported",
                "source": "tools/pmc_traffic.sh (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, --kernel-trace only), python3 tools/reid_bench.py 512 2 x3; profiles/r06_reid_x3_512_pmc_traffic.txt",
                "round": 6, "hbm_bytes_per_pass": (float(tot[3]) + float(tot[4])) * 1e6, "read_bytes": float(tot[3]) * 1e6, "write_bytes": float(tot[4]) * 1e6, "kernel_us_per_pass": float(tot[2])})
if os.path.exists(P + "pmc_dt/entries.json"):       # Decision-Transformer launch shapes (tools/pmc_dt_traffic.sh): this round's measurement replaces the entry
    for k, v in json.load(open(P + "pmc_dt/entries.json")).items():
        if k in d and d[k].get("round", 0) < 6:
            d[k + "_r%02d" % d[k].get("round", 5)] = d[k]          # keep the previous round's entry beside it
        v["round"] = 6
        d[k] = v
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
cp = {"bench_steps20.json": "r06_bench_steps20.json", "bench_default.json": "r06_bench_default.json",
      "bench_detail_steps20.json": "r06_bench_detail_steps20.json", "bench_detail_default.json": "r06_bench_detail_default.json",
      "reid_x3_512_timeline.txt": "r06_reid_x3_512_timeline.txt", "reid_x3_88_timeline.txt": "r06_reid_x3_88_timeline.txt", "reid_x3_40_timeline.txt": "r06_reid_x3_40_timeline.txt",
      "reid_x3_512.stats.txt": "r06_reid_x3_512_kernel_stats.txt", "reid_x3_88.stats.txt": "r06_reid_x3_88_kernel_stats.txt", "reid_x3_40.stats.txt": "r06_reid_x3_40_kernel_stats.txt",
      "dt_f32_steps20.stats.txt": "r06_dt_f32_steps20_kernel_stats.txt", "dt_x3_steps20.stats.txt": "r06_dt_x3_steps20_kernel_stats.txt",
      "dt_f32_steps20/t_kernel_stats.csv": "r06_dt_f32_steps20_rocprof_kernel_stats.csv", "dt_x3_steps20/t_kernel_stats.csv": "r06_dt_x3_steps20_rocprof_kernel_stats.csv",
      "dtl_cfg5_f16.stats.txt": "r06_dtl_cfg5_f16_kernel_stats.txt", "dtl_cfg4_x3.stats.txt": "r06_dtl_cfg4_x3_kernel_stats.txt", "dtl_cfg4_f32.stats.txt": "r06_dtl_cfg4_f32_kernel_stats.txt",
      "dtl_cfg5_f16_sq_counters.txt": "r06_dtl_cfg5_f16_sq_counters.txt", "dtl_cfg4_x3_sq_counters.txt": "r06_dtl_cfg4_x3_sq_counters.txt",
      "hbm_kernels.stats.txt": "r06_hbm_kernels_kernel_stats.txt", "hbm_kernels_pmc_traffic.txt": "r06_hbm_kernels_pmc_traffic.txt", "hbm_kernels.log": "r06_hbm_kernels_table.json",
      "reid_x3_512_sq_counters.txt": "r06_reid_x3_512_sq_counters.txt", "reid_x3_512_pmc_traffic.txt": "r06_reid_x3_512_pmc_traffic.txt"}
for a, b in cp.items():
    if os.path.exists(P + a):
        shutil.copy(P + a, "profiles/" + b)
    else:
        print("missing", a)
for a, b in (("gpurun_out/r06_gpu_tests.txt", "profiles/r06_gpu_tests.txt"), ("gpurun_out/r06_smoke.txt", "profiles/r06_smoke.txt"),
             ("gpurun_out/r06_x3_tail_ab.txt", "profiles/r06_x3_tail_ab.txt")):
    if os.path.exists(a):
        shutil.copy(a, b)
