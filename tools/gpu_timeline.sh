# per-launch timelines of one ReID pass at several batch sizes: tools/gpu_timeline.sh <outdir> <n> [<n> ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$1; shift; mkdir -p $O
for N in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tl$N -o t -- python3 tools/reid_bench.py $N 3 > $O/tl$N.log 2>&1
  python3 tools/timeline.py $(find $O/tl$N -name "*kernel_trace.csv" | head -1) stem_ -v > $O/timeline_$N.txt 2>&1
  find $O/tl$N -name "*.csv" -size +8M -delete
  tail -1 $O/tl$N.log
done
