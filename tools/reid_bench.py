#!/usr/bin/env python3
"""Time the ReID extractor alone: python tools/reid_bench.py [n_crops] [iters] [f16|f32]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from busca_amd import _lib, synth  # noqa: E402
from busca_amd.reid import ReIDEncoderHIP  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prec = sys.argv[3] if len(sys.argv) > 3 else "f16"
ctx = _lib.Context(0)
m = ReIDEncoderHIP(ctx, synth.reid_state_dict(3), precision=prec)
crops = torch.from_numpy(synth.randint_u8(1, "c", (n, 384, 128, 3))).cuda()
for _ in range(2):
    f = m.forward(crops)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    f = m.forward(crops)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print("reid n=%d (%s): %.3f ms/forward, %.1f TFLOP/s (8.01 GFLOP/crop), %.0f crops/s" % (n, prec, dt * 1e3, n * 8.01e9 / dt / 1e12, n / dt))
