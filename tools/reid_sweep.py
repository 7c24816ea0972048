#!/usr/bin/env python3
"""Sweep one ReID schedule knob over batch sizes: python tools/reid_sweep.py ENV_NAME v1,v2,... [n1,n2,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.reid import ReIDEncoderHIP
name, vals = sys.argv[1], sys.argv[2].split(",")
ns = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "8,22,40,88,160,352,512").split(",")]
ctx = _lib.Context(0)
sd = synth.reid_state_dict(3)
crops = {n: torch.from_numpy(synth.randint_u8(1, "c", (n, 384, 128, 3))).cuda() for n in ns}
print("%-28s" % name + "".join("%9d" % n for n in ns))
for v in vals:
    if v == "default": os.environ.pop(name, None)
    else: os.environ[name] = v
    m = ReIDEncoderHIP(ctx, sd)
    row = []
    for n in ns:
        for _ in range(3): m.forward(crops[n])
        torch.cuda.synchronize(); t = time.perf_counter()
        it = 20 if n <= 160 else 6
        for _ in range(it): m.forward(crops[n])
        torch.cuda.synchronize(); row.append((time.perf_counter() - t) / it * 1e3)
    print("%-28s" % v + "".join("%9.3f" % x for x in row), flush=True)
