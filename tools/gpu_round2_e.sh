cd $GRAFT_REPO_ROOT
for F in 8 16 32; do
BUSCA_DT_OCC2=1 BUSCA_DT_NTRK=1 python bench.py --precision f16 --inflight $F --steps 1600 --warmup 160 --cpu-seconds 0 --latency-samples 0 --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('occ2 F=$F', round(d['value']), 'steps/s  frac', round(d['roofline']['frac'],4), 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4))"
done
BUSCA_DT_OCC2=1 BUSCA_DT_NTRK=1 python -m pytest tests/test_dt_gpu.py -m gpu -q -x -k "golden and d256" 2>&1 | tail -3
