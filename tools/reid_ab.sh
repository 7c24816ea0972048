#!/bin/bash
# In-situ A/B of ReID schedule knobs (run through gpurun): tools/reid_ab.sh "<n list>" "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]
# Each configuration is timed twice, interleaved, at every batch size (tools/reid_bench.py N 5).
cd $GRAFT_REPO_ROOT
NS=$1; shift
for rep in 1 2; do
  for cfg in "" "$@"; do
    for n in $NS; do
      printf "%-40s " "[${cfg:-default}]"
      env $cfg python3 tools/reid_bench.py $n 5 2>&1 | grep "reid n="
    done
  done
done
