cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
python -m pytest tests/test_ecc.py -m gpu -q 2>&1 | tail -3
python tools/reid_sweep.py BUSCA_REID_HALO_MIN default,32,64,96,128 > gpurun_out/r2d/halo_min.txt 2>&1
cat gpurun_out/r2d/halo_min.txt
python tools/reid_sweep.py BUSCA_REID_GRAM_MIN default,8192,16384,32768 > gpurun_out/r2d/gram_min.txt 2>&1
cat gpurun_out/r2d/gram_min.txt
python tools/reid_sweep.py BUSCA_REID_SPLITK_BLOCKS default,128,256,512,768 8,22,40,88,160 > gpurun_out/r2d/splitk.txt 2>&1
cat gpurun_out/r2d/splitk.txt
python tools/reid_sweep.py BUSCA_REID_DIRECT_ROWS default,128,256,1024,2048 > gpurun_out/r2d/direct.txt 2>&1
cat gpurun_out/r2d/direct.txt
