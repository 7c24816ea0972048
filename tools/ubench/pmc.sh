#!/bin/bash
# SQ counters of a ubench binary: tools/ubench/pmc.sh <outdir> <binary> [args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$1; shift; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT -o p1 -- "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT -o p2 -- "$@" > $OUT/p2.log 2>&1
python3 profiles/pmc_kernel_table.py $OUT > $OUT/table.txt 2>&1
find $OUT -name "*.csv" -size +8M -delete
cat $OUT/table.txt
