#!/usr/bin/env python3
"""Time busca_dt_forward on an arbitrary shape: python tools/dt_bench.py B P d precision [iters] [tiled]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP
B, P, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else "f16"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
if len(sys.argv) > 6 and sys.argv[6] == "tiled":
    os.environ["BUSCA_DT_TILED"] = "1"
L = 11
ctx = _lib.Context(0)
m = DecisionTransformerHIP(ctx, synth.dt_state_dict(7, d, 2 * d), precision=prec)
inp = {k: torch.from_numpy(v).cuda() for k, v in synth.dt_inputs(7, B, L, P).items()}
for _ in range(3):
    m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
fl = bench.dt_step_flops(B, L, P, d, 2 * d)
print("B=%d P=%d d=%d T=%d %s%s: %.3f ms  %.1f TFLOP/s  (%.1f GFLOP)" % (B, P, d, L + 2 * (P + 2), prec, " tiled" if os.environ.get("BUSCA_DT_TILED") else "", dt * 1e3, fl / dt / 1e12, fl / 1e9))
