#!/bin/bash
# x3 conv harness: experiments + SQ counters of selected cases.  Usage: tools/gpu_x3_exp.sh <tag> <cases...>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/x3/$TAG; mkdir -p $O
tools/ubench/x3_conv_bench "$@" > $O/bench.txt 2>&1
cat $O/bench.txt | grep -v amdgpu.ids
if [ -n "$PMC" ]; then
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O -o p1 -- tools/ubench/x3_conv_bench $PMC > $O/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O -o p2 -- tools/ubench/x3_conv_bench $PMC > $O/p2.log 2>&1
python3 profiles/pmc_kernel_table.py $O > $O/table.txt 2>&1
find $O -name "*.csv" -size +8M -delete
cat $O/table.txt
fi
