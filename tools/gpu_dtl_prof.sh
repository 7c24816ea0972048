# kernel-trace stats of the layer-wise Decision-Transformer path: tools/gpu_dtl_prof.sh <outdir> B P d prec
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$1; shift; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dtl -o t -- python3 tools/dt_cfg_bench.py "$@" 5 > $O/dtl.log 2>&1
tail -1 $O/dtl.log
python3 - "$O" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/dtl/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print("%-90s %5s calls %10.1f us avg %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
find $O -name "*.csv" -size +8M -delete
