cd $GRAFT_REPO_ROOT
for P in f32 f16; do F=8; [ $P = f16 ] && F=16
python bench.py --precision $P --inflight $F --steps 1600 --warmup 160 --cpu-seconds 0 --latency-samples 200 --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$P', round(d['value']), 'steps/s  frac', round(d['roofline']['frac'],4), 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'p50', round(d['p50_latency_ms'],4))"
done
python -m pytest tests/test_dt_gpu.py tests/test_associate_gpu.py -m gpu -q 2>&1 | tail -2
