#!/bin/bash
# SQ counter passes over the ReID extractor (two passes of 8 SQ counters each; run through gpurun). Usage: tools/pmc_reid_sq.sh <n_crops> <outdir>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=$1; OUT=$2; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT -o p1 -- python3 tools/reid_bench.py $N 2 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT -o p2 -- python3 tools/reid_bench.py $N 2 > $OUT/p2.log 2>&1
python3 profiles/pmc_kernel_table.py $OUT > $OUT/table.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
tail -2 $OUT/p2.log
