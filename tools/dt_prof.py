#!/usr/bin/env python3
"""Debug: per-phase cycle stamps of workgroup 0 (run with BUSCA_DT_PROF=1). python tools/dt_prof.py f16 256"""
import os, sys
os.environ["BUSCA_DT_PROF"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = _lib.Context(0)
m = DecisionTransformerHIP(ctx, synth.dt_state_dict(7, 256, 512), precision=prec)
inp = synth.dt_inputs(7, B, 11, 16)
for _ in range(3):
    m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
torch.cuda.synchronize()
