#!/usr/bin/env python3
"""ReID pass with and without multiplicities (busca_reid_forward_w): python tools/reid_weighted_bench.py [n ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from busca_amd import _lib, synth
from busca_amd.reid import ReIDEncoderHIP
ctx = _lib.Context(0)
m = ReIDEncoderHIP(ctx, synth.reid_state_dict(3))
for n in [int(a) for a in sys.argv[1:]] or [150]:
    crops = torch.from_numpy(synth.randint_u8(1, "c", (n, 384, 128, 3))).cuda()
    w = np.ones(n, np.float32); w[::2] = 5
    for name, kw in (("plain", {}), ("weighted", {"weights": w})):
        for _ in range(2): m.forward(crops, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m.forward(crops, **kw)
        torch.cuda.synchronize()
        print("n=%d %s: %.3f ms" % (n, name, (time.perf_counter() - t0) / 5 * 1e3))
