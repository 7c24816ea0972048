cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a/pytest.log 2>&1
tail -5 gpurun_out/r2a/pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_20.json 2> gpurun_out/r2a/bench_20.err
tail -c 600 gpurun_out/r2a/bench_20.err
python tools/reid_bench.py 512 5 > gpurun_out/r2a/reid.log 2>&1; python tools/reid_bench.py 88 10 >> gpurun_out/r2a/reid.log 2>&1; python tools/reid_bench.py 8 20 >> gpurun_out/r2a/reid.log 2>&1
cat gpurun_out/r2a/reid.log
tools/pmc_traffic.sh gpurun_out/r2a/pmc_reid512 python3 tools/reid_bench.py 512 2 > /dev/null
tools/pmc_traffic.sh gpurun_out/r2a/pmc_reid88 python3 tools/reid_bench.py 88 2 > /dev/null
python profiles/pmc_traffic_summary.py gpurun_out/r2a/pmc_reid512 4 > gpurun_out/r2a/reid512_traffic.txt
python profiles/pmc_traffic_summary.py gpurun_out/r2a/pmc_reid88 4 > gpurun_out/r2a/reid88_traffic.txt
tail -3 gpurun_out/r2a/reid512_traffic.txt
# keep the merge small
find gpurun_out/r2a -name "*.csv" -size +20M -delete
