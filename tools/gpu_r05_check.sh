#!/bin/bash
# Round-5 check (run through gpurun): geometry / harness / bench tests, the driver's bench command, x3 ReID timelines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_geometry_gpu.py tests/test_harness_gpu.py tests/test_bench_gpu.py tests/test_dt_tiled_gpu.py -q -x > $O/tests_a.log 2>&1; tail -4 $O/tests_a.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; tail -c 600 $O/bench_steps20.err
python3 - <<'PY'
import json
r = json.load(open("gpurun_out/r05/bench_steps20.json"))
print("headline", r["value"], r["roofline"]["frac"])
for k in ("full_step", "full_step_f16", "full_step_f32"):
    print(k, r.get(k, {}).get("ms_per_step"))
print({k: v.get("p50_assoc_latency_ms") for k, v in r["assoc_e2e"].items() if isinstance(v, dict)})
print({k: v.get("p50_crop_ms") for k, v in r["assoc_e2e"].items() if isinstance(v, dict)})
print(json.dumps(r.get("hbm_kernels"), indent=1))
print({k: (v.get("ms_per_step"), v.get("roofline", {}).get("frac"), v.get("roofline", {}).get("kernel_launches_per_call")) for k, v in r["configs"].items()})
PY
for N in 512 150; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/reid_x3_$N -o t -- python3 tools/reid_bench.py $N 3 x3 > $O/reid_x3_$N.log 2>&1
python3 tools/kstats.py $O/reid_x3_$N > $O/reid_x3_${N}_kernel_stats.txt
python3 tools/timeline.py $(find $O/reid_x3_$N -name "*kernel_trace.csv" | head -1) "conv_x3_kernel<2, 2, 2, 4, 2, 7" -v > $O/reid_x3_${N}_timeline.txt 2>/dev/null
done
find $O -name "*.csv" -size +4M -delete
tail -3 $O/reid_x3_512.log
