#!/bin/bash
# first light of the split-fp16 ReID flavour: parity tests, then timing next to the other two flavours, then a kernel-trace profile
mkdir -p gpurun_out/x3
python -m pytest tests/test_reid_gpu.py -x -q -k "f32_mode or golden_reference or negative_batchnorm or weighted_statistics_equal" -s 2>&1 | tail -25 > gpurun_out/x3/tests.txt
cat gpurun_out/x3/tests.txt
for n in 88 512; do for p in x3 f32 f16; do python tools/reid_bench.py $n 5 $p; done; done 2>&1 | tee gpurun_out/x3/bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/x3/prof512 -o t -- python3 $GRAFT_REPO_ROOT/tools/reid_bench.py 512 3 x3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/kstats.py gpurun_out/x3/prof512 > gpurun_out/x3/prof512_stats.txt; python3 tools/timeline.py $(find gpurun_out/x3/prof512 -name "*kernel_trace.csv" | head -1) preprocess -v > gpurun_out/x3/timeline_512.txt 2>/dev/null
find gpurun_out/x3 -name "*.csv" -size +6M -delete; head -30 gpurun_out/x3/prof512_stats.txt
