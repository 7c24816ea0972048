# rocprofv3 evidence for the round: kernel-trace stats of the default bench legs + ReID passes, PMC of the DT kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dt_f32 -o t -- python3 bench.py --precision f32 --steps 160 --warmup 16 --cpu-seconds 0 --latency-samples 0 --no-variants > $O/dt_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dt_f16 -o t -- python3 bench.py --precision f16 --inflight 16 --steps 320 --warmup 32 --cpu-seconds 0 --latency-samples 0 --no-variants > $O/dt_f16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/reid512 -o t -- python3 tools/reid_bench.py 512 3 > $O/reid512.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/reid88 -o t -- python3 tools/reid_bench.py 88 3 > $O/reid88.log 2>&1
for P in f32 f16; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/pmc_$P -o p1 -- python3 bench.py --precision $P --steps 80 --warmup 8 --cpu-seconds 0 --latency-samples 0 --no-variants > $O/pmc_$P.p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$P -o p2 -- python3 bench.py --precision $P --steps 80 --warmup 8 --cpu-seconds 0 --latency-samples 0 --no-variants > $O/pmc_$P.p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_$P -o p3 -- python3 bench.py --precision $P --steps 80 --warmup 8 --cpu-seconds 0 --latency-samples 0 --no-variants > $O/pmc_$P.p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_$P -o p4 -- python3 bench.py --precision $P --steps 80 --warmup 8 --cpu-seconds 0 --latency-samples 0 --no-variants > $O/pmc_$P.p4.log 2>&1
done
tools/pmc_traffic.sh $O/pmc_reid512 python3 tools/reid_bench.py 512 2 > /dev/null
find $O -name "*.csv" -size +20M -delete
ls $O
