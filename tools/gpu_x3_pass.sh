#!/bin/bash
# x3 flavour: pass-level timing under schedule knobs + kernel-trace profile.  Usage: tools/gpu_x3_pass.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/x3/$1; mkdir -p $O
tools/ubench/x3_conv_bench 0 1 3 7 11 14 18 2>&1 | grep -v amdgpu.ids > $O/conv_bench.txt; cat $O/conv_bench.txt
for ml in 15 3 1 0; do echo "== BUSCA_REID_X3_MERGE_LAYERS=$ml"; BUSCA_REID_X3_MERGE_LAYERS=$ml python tools/reid_bench.py 512 5 x3; BUSCA_REID_X3_MERGE_LAYERS=$ml python tools/reid_bench.py 88 5 x3; done 2>&1 | grep -v amdgpu.ids | tee $O/bench.txt
python -m pytest tests/test_reid_gpu.py -x -q -k "f32_mode or golden_reference" 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof512 -o t -- python3 tools/reid_bench.py 512 3 x3 > /dev/null 2>&1
python tools/kstats.py $O/prof512 > $O/prof512_stats.txt; python3 tools/timeline.py $(find $O/prof512 -name "*kernel_trace.csv" | head -1) preprocess -v > $O/timeline_512.txt 2>/dev/null
find $O -name "*.csv" -size +6M -delete; head -24 $O/prof512_stats.txt
