#!/usr/bin/env python3
"""A/B of one ReID schedule option inside ONE process, alternating between the two values (box-to-box and clock-state differences cancel):
python tools/reid_ab.py <option> <value_a> <value_b> [n_crops] [flavour] [rounds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from busca_amd import _lib, synth  # noqa: E402
from busca_amd.reid import ReIDEncoderHIP  # noqa: E402

opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 512
prec = sys.argv[5] if len(sys.argv) > 5 else "x3"
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 8
ctx = _lib.Context(0)
m = ReIDEncoderHIP(ctx, synth.reid_state_dict(3), precision=prec)
crops = torch.from_numpy(synth.randint_u8(1, "c", (n, 384, 128, 3))).cuda()
ts = {va: [], vb: []}
for r in range(rounds + 1):
    for v in (va, vb):
        ctx.set_option(opt, v)
        m.forward(crops)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m.forward(crops)
        torch.cuda.synchronize()
        if r:
            ts[v].append((time.perf_counter() - t0) / 3 * 1e3)
for v in (va, vb):
    print("%s=%d n=%d (%s): median %.3f ms, min %.3f ms" % (opt, v, n, prec, float(np.median(ts[v])), min(ts[v])))
