#!/bin/bash
# HBM traffic counters per kernel (run on the GPU box through gpurun): FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2), each with --kernel-trace only.
# Usage: tools/pmc_traffic.sh <outdir> <program and args...>      e.g.  tools/pmc_traffic.sh gpurun_out/pmc_reid512 python3 tools/reid_bench.py 512 2
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -o fetch -- "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT -o write -- "$@" > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- "$@" > $OUT/trace.log 2>&1
ls $OUT
