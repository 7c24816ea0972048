#!/bin/bash
# PC sampling of the fused Decision-Transformer kernel (gpurun box). Usage: tools/pcsample_dt.sh <outdir> <method: stochastic|host_trap> [B=256] [prec=f16]
# Round 3: both methods answer "Given PC sampling configuration is not supported on any of the agents" for the unprivileged user of this pool;
# kept for a box where it is allowed.
export TMPDIR=/tmp
OUT=$1; M=${2:-stochastic}; B=${3:-256}; PR=${4:-f16}
mkdir -p $OUT
if [ "$M" = stochastic ]; then UNIT="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else UNIT="--pc-sampling-unit time --pc-sampling-interval 1"; fi
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M $UNIT --output-format csv -d $OUT -o pcs -- python3 tools/dt_cfg_bench.py $B 16 256 $PR 400 > $OUT/pcs.log 2>&1
echo rc=$?
tail -5 $OUT/pcs.log
find $OUT -type f | head; du -sh $OUT
