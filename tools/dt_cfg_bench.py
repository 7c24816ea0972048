#!/usr/bin/env python3
"""One BASELINE shape through busca_dt_forward: python tools/dt_cfg_bench.py B P d precision [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP
B, P, d, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
ctx = _lib.Context(0)
m = DecisionTransformerHIP(ctx, synth.dt_state_dict(7, d=d, ff=2 * d), precision=prec)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.dt_inputs(7, B, 11, P).items()}
m.reserve(B, 11, P)
for _ in range(3): m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(iters): m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / iters
T = 11 + 2 * (P + 2); ff = 2 * d
fl = 2 * B * (11 + P) * 512 * d + 4 * (2 * B * T * d * 3 * d + 4 * B * T * T * d + 2 * B * T * d * d + 4 * B * T * d * ff) + 2 * B * (P + 2) * d
print("B=%d P=%d d=%d %s: %.3f ms/step, %.1f TFLOP/s" % (B, P, d, prec, dt * 1e3, fl / dt / 1e12))
