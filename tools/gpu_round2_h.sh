cd $GRAFT_REPO_ROOT
for lib in "$GRAFT_REPO_ROOT/busca_amd/libbusca_hip.so" "$GRAFT_REPO_ROOT/busca_amd/libbusca_unroll.so" "$GRAFT_REPO_ROOT/busca_amd/libbusca_hip.so" "$GRAFT_REPO_ROOT/busca_amd/libbusca_unroll.so"; do
echo "== lib: [$lib]"
BUSCA_HIP_LIB=$lib python bench.py --precision f32 --inflight 8 --steps 1600 --warmup 160 --cpu-seconds 0 --latency-samples 200 --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f32', round(d['value']), 'steps/s  frac', round(d['roofline']['frac'],4), 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'p50', round(d['p50_latency_ms'],4))"
done
