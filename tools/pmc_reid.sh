#!/bin/bash
# PMC passes over the ReID extractor (run on the GPU box through gpurun). Usage: tools/pmc_reid.sh <n_crops> <outdir>
export TMPDIR=/tmp
N=$1; OUT=$2; mkdir -p $OUT
CMD="python3 tools/reid_bench.py $N 2"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT WRITE_SIZE TCC_MISS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o p3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES --kernel-trace --output-format csv -d $OUT -o p5 -- $CMD > $OUT/p5.log 2>&1
ls $OUT
