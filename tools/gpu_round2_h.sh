cd $GRAFT_REPO_ROOT
for lib in "$GRAFT_REPO_ROOT/busca_amd/libbusca_hip.so" "$GRAFT_REPO_ROOT/busca_amd/libbusca_relu.so" "$GRAFT_REPO_ROOT/busca_amd/libbusca_hip.so" "$GRAFT_REPO_ROOT/busca_amd/libbusca_relu.so"; do
echo "== lib: [$lib]"
for P in f32 f16; do
F=8; [ $P = f16 ] && F=16
BUSCA_HIP_LIB=$lib python bench.py --precision $P --inflight $F --steps 1600 --warmup 160 --cpu-seconds 0 --latency-samples 200 --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$P', round(d['value']), 'steps/s  frac', round(d['roofline']['frac'],4), 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'p50', round(d['p50_latency_ms'],4))"
done
for n in ; do BUSCA_HIP_LIB=$lib python tools/reid_bench.py $n 6 2>/dev/null | tail -1; done
done
