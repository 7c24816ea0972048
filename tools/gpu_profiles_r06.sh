#!/bin/bash
# Round-6 evidence (run through gpurun): kernel-trace stats of the bench's primary line (exact f32) + x3 variant, the x3 ReID passes, the layer-wise DT shapes
# (cfg5 f16, cfg4 x3 / f32), the HBM-bound geometry kernels; SQ counters + HBM traffic of the x3 ReID pass and of the layer-wise DT.
# Usage: bash tools/gpu_profiles_r06.sh [outdir] [part]      part: a = traces + bench lines, b = ReID counters + traffic, c = HBM traffic of the Decision-Transformer
#        launch shapes, e = SQ counters of the layer-wise DT (cfg5 f16, cfg4 x3)      (default: a, b, c, e)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/r06/prof}; PART=${2:-abce}; mkdir -p $O
tr() { rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -o t -- "${@:2}" > $O/$1.log 2>&1; python3 tools/kstats.py $O/$1 > $O/$1.stats.txt; }
sq() { # name, program...
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/sq_$1 -o p1 -- "${@:2}" > $O/sq_$1.p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq_$1 -o p2 -- "${@:2}" > $O/sq_$1.p2.log 2>&1
python3 profiles/pmc_kernel_table.py $O/sq_$1 > $O/$1_sq_counters.txt 2>&1
}
if [[ $PART == *a* ]]; then
tr dt_f32_steps20 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --latency-samples 0 --no-variants --split-steps 0
tr dt_x3_steps20 python3 bench.py --precision x3 --steps 20 --warmup 5 --cpu-seconds 0 --latency-samples 0 --no-variants --split-steps 0
tr reid_x3_512 python3 tools/reid_bench.py 512 3 x3
tr reid_x3_88 python3 tools/reid_bench.py 88 3 x3
tr reid_x3_40 python3 tools/reid_bench.py 40 3 x3
tr dtl_cfg5_f16 python3 tools/dt_cfg_bench.py 512 64 512 f16 5
tr dtl_cfg4_x3 python3 tools/dt_cfg_bench.py 256 32 512 x3 5
tr dtl_cfg4_f32 python3 tools/dt_cfg_bench.py 256 32 512 f32 5
tr hbm_kernels python3 tools/hbm_kernels_bench.py
for N in 512 88 40; do python3 tools/timeline.py $(find $O/reid_x3_$N -name "*kernel_trace.csv" | head -1) "conv_x3_kernel<2, 2, 2, 4, 2, 7" -v > $O/reid_x3_${N}_timeline.txt 2>/dev/null; done
python3 bench.py --steps 20 --warmup 5 --detail $O/bench_detail_steps20.json > $O/bench_steps20.json 2> $O/bench_steps20.err
python3 bench.py --detail $O/bench_detail_default.json > $O/bench_default.json 2> $O/bench_default.err
fi
if [[ $PART == *e* ]]; then
sq dtl_cfg5_f16 python3 tools/dt_cfg_bench.py 512 64 512 f16 3
sq dtl_cfg4_x3 python3 tools/dt_cfg_bench.py 256 32 512 x3 3
fi
if [[ $PART == *c* ]]; then
bash tools/pmc_dt_traffic.sh $O/pmc_dt > $O/pmc_dt.log 2>&1
fi
if [[ $PART == *b* ]]; then
sq reid_x3_512 python3 tools/reid_bench.py 512 2 x3
bash tools/pmc_traffic.sh $O/pmc_reid_x3_512 python3 tools/reid_bench.py 512 2 x3 > /dev/null 2>&1
python3 profiles/pmc_traffic_summary.py $O/pmc_reid_x3_512 4 > $O/reid_x3_512_pmc_traffic.txt
bash tools/pmc_traffic.sh $O/pmc_hbm_kernels python3 tools/hbm_kernels_bench.py > /dev/null 2>&1
python3 profiles/pmc_traffic_summary.py $O/pmc_hbm_kernels 21 > $O/hbm_kernels_pmc_traffic.txt
fi
find $O -name "*.csv" -size +4M -delete
ls $O | head -80
