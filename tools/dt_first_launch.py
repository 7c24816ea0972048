#!/usr/bin/env python3
"""First-launch cost of the flavours a step count dispatches to: warm-up with 160 tracks, then the first 640-track call, then steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP
prec = sys.argv[1] if len(sys.argv) > 1 else "x3"
ctx = _lib.Context(0)
m = DecisionTransformerHIP(ctx, synth.dt_state_dict(3, d=256, ff=512), activation="relu", precision=prec)
def inputs(B): return {k: torch.from_numpy(v).cuda() for k, v in synth.dt_inputs(3, B, 11, 16).items()}
def t(i, n=1):
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): m.forward(i["mem_feat"], i["can_feat"], i["mem_boxes"], i["can_boxes"])
    torch.cuda.synchronize(); return (time.perf_counter() - a) * 1e3 / n
i160, i640, i2048 = inputs(160), inputs(640), inputs(2048)
print("%s: B=160 first %.3f then %.3f | B=640 first %.3f second %.3f then %.3f | B=2048 first %.3f then %.3f" % (prec, t(i160), t(i160, 10), t(i640), t(i640), t(i640, 20), t(i2048), t(i2048, 10)))
