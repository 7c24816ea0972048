cd $GRAFT_REPO_ROOT
for mask in 0 1 2 4 8 10 14 15; do for cfg in "512 64 512 f16" "128 32 512 f16" "256 32 512 f32"; do
echo -n "mask=$mask  "; BUSCA_DTL_RT_MASK=$mask python tools/dt_cfg_bench.py $cfg 10 2>&1 | tail -1
done; done
