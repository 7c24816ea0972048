# rocprofv3 evidence for the round (run through gpurun): kernel-trace stats of the bench legs + ReID passes, SQ counters of the DT and ReID
# kernels, HBM traffic of the ReID passes.  Usage: bash tools/gpu_profiles.sh [outdir]     (summaries are then copied into profiles/ by hand)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/prof}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dt_f32 -o t -- python3 bench.py --precision f32 --steps 20 --warmup 5 --cpu-seconds 0 --latency-samples 0 --no-variants --split-steps 0 > $O/dt_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dt_f32_k640 -o t -- python3 bench.py --precision f32 --steps 640 --warmup 64 --cpu-seconds 0 --latency-samples 0 --no-variants --split-steps 0 > $O/dt_f32_k640.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dt_f16 -o t -- python3 bench.py --precision f16 --inflight 16 --steps 320 --warmup 32 --cpu-seconds 0 --latency-samples 0 --no-variants --split-steps 0 > $O/dt_f16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/reid512 -o t -- python3 tools/reid_bench.py 512 3 > $O/reid512.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/reid88 -o t -- python3 tools/reid_bench.py 88 3 > $O/reid88.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/reid_f32_88 -o t -- python3 tools/reid_bench.py 88 3 f32 > $O/reid_f32_88.log 2>&1
for d in dt_f32 dt_f32_k640 dt_f16 reid512 reid88 reid_f32_88; do
  python3 - $O/$d > $O/$d.stats.txt <<PY
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    a = agg.setdefault(r["Kernel_Name"][:88], [0, 0, 10**18, 0])
    t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[0] += 1; a[1] += t; a[2] = min(a[2], t); a[3] = max(a[3], t)
tot = sum(v[1] for v in agg.values()) or 1
print("%-88s %7s %13s %11s %6s %10s %10s" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-88s %7d %13d %11d %6.2f %10d %10d" % (k, v[0], v[1], v[1] // v[0], 100.0 * v[1] / tot, v[2], v[3]))
PY
done
for N in 512 88 8; do python3 tools/timeline.py $(find $O/reid$N -name "*kernel_trace.csv" 2>/dev/null | head -1) stem_ -v > $O/timeline_$N.txt 2>/dev/null; done
bash tools/pmc_reid_sq.sh 512 $O/sq_reid512 > /dev/null 2>&1
bash tools/pmc_reid_sq.sh 88 $O/sq_reid88 > /dev/null 2>&1
for P in f32 f16; do bash tools/pmc_dt.sh $P $O/pmc_dt_$P > /dev/null 2>&1; python3 profiles/pmc_summary.py $O/pmc_dt_$P dt_fused > $O/pmc_dt_$P.txt; done
bash tools/pmc_traffic.sh $O/pmc_reid512 python3 tools/reid_bench.py 512 2 > /dev/null 2>&1
python3 profiles/pmc_traffic_summary.py $O/pmc_reid512 4 > $O/reid512_traffic.txt
bash tools/pmc_traffic.sh $O/pmc_reid88 python3 tools/reid_bench.py 88 2 > /dev/null 2>&1
python3 profiles/pmc_traffic_summary.py $O/pmc_reid88 4 > $O/reid88_traffic.txt
find $O -name "*.csv" -size +6M -delete
ls $O
