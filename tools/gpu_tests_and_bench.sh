cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tests
( time python -m pytest tests -m gpu -q ) > gpurun_out/tests/pytest.log 2>&1
tail -15 gpurun_out/tests/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/tests/bench_20.json 2> gpurun_out/tests/bench_20.err
tail -c 400 gpurun_out/tests/bench_20.err
