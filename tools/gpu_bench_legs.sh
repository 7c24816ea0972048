cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench
python tools/cfg4_step.py 3 f16 2>&1 | tail -2
python tools/cfg4_step.py 3 f32 2>&1 | tail -2
python -m pytest tests/test_properties_gpu.py -m gpu -q -x -k cfg4_full 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/bench/bench_20.json 2> gpurun_out/bench/bench_20.err; tail -c 300 gpurun_out/bench/bench_20.err
python bench.py > gpurun_out/bench/bench_default.json 2> gpurun_out/bench/bench_default.err
