#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/x3/$1; mkdir -p $O
tools/ubench/x3_conv_bench 12 13 14 19 20 2>&1 | grep -v amdgpu.ids > $O/conv_bench.txt; cat $O/conv_bench.txt
for h in 512 0; do for n in 40 88 150 352 512; do echo -n "X3_HALF=$h "; BUSCA_REID_X3_HALF=$h python tools/reid_bench.py $n 5 x3; done; done 2>&1 | grep -v amdgpu.ids | tee $O/bench.txt
python -m pytest tests/test_reid_gpu.py -x -q -k "f32_mode or golden_reference or large_batch_schedule or weighted_statistics_equal or negative" 2>&1 | tail -3
