#!/bin/bash
# layer-wise Decision Transformer: parity tests + A/B of the fused feed-forward block (BUSCA_DTL_FFN) on the BASELINE shapes + kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/dtl/$1; mkdir -p $O
python -m pytest tests/test_dt_tiled_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/tests.txt
for f in "2 1" "2 0" "0 0"; do set -- $f; for cfg in "512 64 512 f16" "128 32 512 f16" "128 32 512 f32" "256 32 512 f32" "128 32 256 f16"; do echo -n "DTL_FFN=$1 DTL_ATTN=$2 "; BUSCA_DTL_FFN=$1 BUSCA_DTL_ATTN=$2 python tools/dt_cfg_bench.py $cfg 10; done; done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -o t -- python3 tools/dt_cfg_bench.py 512 64 512 f16 5 > /dev/null 2>&1
python tools/kstats.py $O/prof_cfg5 > $O/cfg5_kernel_stats.txt; head -16 $O/cfg5_kernel_stats.txt
find $O -name "*.csv" -size +6M -delete
