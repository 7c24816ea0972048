cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
( time python -m pytest tests -m gpu -q -x tests/test_ecc.py tests/test_harness_gpu.py tests/test_associate_gpu.py ) > gpurun_out/r2c/pytest.log 2>&1
tail -8 gpurun_out/r2c/pytest.log
python - <<'PY' > gpurun_out/r2c/ecc_time.txt 2>&1
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from test_ecc import _pair
from busca_amd import tracking
im1, im2, M = _pair(H=1080, W=1920, th=0.004, tx=5.5, ty=-2.25, seed=3)
a, b = torch.from_numpy(im1).cuda(), torch.from_numpy(im2).cuda()
for _ in range(3): tracking.find_transform_ecc(a, b)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): cc, W = tracking.find_transform_ecc(a, b)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print("ecc 1080p: %.3f ms per call, %d iterations, cc %.5f" % (dt * 1e3, tracking.find_transform_ecc.last_iterations, cc))
PY
cat gpurun_out/r2c/ecc_time.txt
