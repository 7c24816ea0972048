#!/bin/bash
# Round-5 ReID check (run through gpurun): the persistent-tail test, then in-process A/B of reid_x3_ptail at several batch sizes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_reid_gpu.py -q -x -k "persistent or x3_large_batch or golden or weighted" > $O/reid_tests.log 2>&1; tail -5 $O/reid_tests.log
for n in 512 150 88 40 22; do python3 tools/reid_ab.py reid_x3_ptail 0 512 $n x3 4; done 2>&1 | tee $O/ptail_ab.txt
