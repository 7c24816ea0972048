#!/bin/bash
# Round-5 evidence (run through gpurun): kernel-trace stats of the bench legs + x3 ReID passes + layer-wise DT shapes + the HBM-bound geometry kernels,
# SQ counters and HBM traffic of the x3 ReID pass, HBM traffic of the geometry kernels.
# Usage: bash tools/gpu_profiles_r05.sh [outdir] [part]      part: a = traces + bench, b = counters, c = HBM traffic of the Decision-Transformer launch shapes, d = SQ counters of the one-kernel Decision Transformer (default: a, b, c)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=${1:-gpurun_out/r05/prof}; PART=${2:-abc}; mkdir -p $O
tr() { rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -o t -- "${@:2}" > $O/$1.log 2>&1; python3 tools/kstats.py $O/$1 > $O/$1.stats.txt; }
if [[ $PART == *a* ]]; then
tr dt_x3_steps20 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --latency-samples 0 --no-variants --split-steps 0
tr dt_f32_steps20 python3 bench.py --precision f32 --steps 20 --warmup 5 --cpu-seconds 0 --latency-samples 0 --no-variants --split-steps 0
tr reid_x3_512 python3 tools/reid_bench.py 512 3 x3
tr reid_x3_352 python3 tools/reid_bench.py 352 3 x3
tr reid_x3_88 python3 tools/reid_bench.py 88 3 x3
tr reid_x3_40 python3 tools/reid_bench.py 40 3 x3
tr dtl_cfg5_f16 python3 tools/dt_cfg_bench.py 512 64 512 f16 5
tr dtl_cfg4_f32 python3 tools/dt_cfg_bench.py 256 32 512 f32 5
tr hbm_kernels python3 tools/hbm_kernels_bench.py
for N in 512 352 88 40; do python3 tools/timeline.py $(find $O/reid_x3_$N -name "*kernel_trace.csv" | head -1) "conv_x3_kernel<2, 2, 2, 4, 2, 7" -v > $O/reid_x3_${N}_timeline.txt 2>/dev/null; done
python3 bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
fi
if [[ $PART == *d* ]]; then
# SQ counters of the one-kernel Decision Transformer, whole rounds (2 048 tracks), x3 and f32
for PR in x3 f32; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/sq_dt_$PR -o p1 -- python3 tools/dt_cfg_bench.py 2048 16 256 $PR 4 > $O/sq_dt_p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq_dt_$PR -o p2 -- python3 tools/dt_cfg_bench.py 2048 16 256 $PR 4 > $O/sq_dt_p2.log 2>&1
python3 profiles/pmc_kernel_table.py $O/sq_dt_$PR > $O/dt_${PR}_sq_counters.txt 2>&1
done
fi
if [[ $PART == *c* ]]; then
bash tools/pmc_dt_traffic.sh $O/pmc_dt > $O/pmc_dt.log 2>&1
fi
if [[ $PART == *b* ]]; then
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/sq_reid_x3_512 -o p1 -- python3 tools/reid_bench.py 512 2 x3 > $O/sq_p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq_reid_x3_512 -o p2 -- python3 tools/reid_bench.py 512 2 x3 > $O/sq_p2.log 2>&1
python3 profiles/pmc_kernel_table.py $O/sq_reid_x3_512 > $O/reid_x3_512_sq_counters.txt 2>&1
bash tools/pmc_traffic.sh $O/pmc_reid_x3_512 python3 tools/reid_bench.py 512 2 x3 > /dev/null 2>&1
python3 profiles/pmc_traffic_summary.py $O/pmc_reid_x3_512 4 > $O/reid_x3_512_pmc_traffic.txt
bash tools/pmc_traffic.sh $O/pmc_hbm_kernels python3 tools/hbm_kernels_bench.py > /dev/null 2>&1
python3 profiles/pmc_traffic_summary.py $O/pmc_hbm_kernels 21 > $O/hbm_kernels_pmc_traffic.txt
fi
find $O -name "*.csv" -size +4M -delete
ls $O | head -80
