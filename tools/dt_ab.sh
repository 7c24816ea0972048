#!/bin/bash
# Decision-Transformer kernel A/B on ONE GPU box: busca_amd/libbusca_base.so (a build of the previous kernel) against the current
# library, interleaved twice; cfgN f16 with one and two tracks per workgroup, cfgN f32, cfgR both.  tools/dt_ab.sh [stamps]
cd $GRAFT_REPO_ROOT
LIBS=${LIBS:-"busca_amd/libbusca_base.so busca_amd/libbusca_hip.so"}
for rep in 1 2; do
  for lib in $LIBS; do
    [ -f $lib ] || continue
    for a in "256 16 256 f16 300" "512 16 256 f16 300" "256 16 256 f32 50" "256 5 512 f16 100" "256 5 512 f32 20"; do
      printf "%-28s " "[$(basename $lib)]"; BUSCA_HIP_LIB=$PWD/$lib python3 tools/dt_cfg_bench.py $a
    done
  done
done
if [ -n "$1" ]; then
  for lib in $LIBS; do
    [ -f $lib ] || continue
    echo "[$lib]"; BUSCA_HIP_LIB=$PWD/$lib python3 tools/dt_prof.py f16 256 2>&1 | grep " w1" | tail -1 | cut -c1-700
    BUSCA_HIP_LIB=$PWD/$lib python3 tools/dt_prof.py f16 512 2>&1 | grep " w1" | tail -1 | cut -c1-700
  done
fi
