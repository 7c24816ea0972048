#!/usr/bin/env python3
"""The 128-track tail of a 640-track (20 steps x 32 lost) launch, one workgroup per track against the token-split tail (dt_split 1: one tile per workgroup,
2: one tile index of two tracks per workgroup): python tools/dt_x3_tail_ab.py [x3|f32] [B ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP
ctx = _lib.Context(0)
prec = sys.argv[1] if len(sys.argv) > 1 else "x3"
Bs = [int(x) for x in sys.argv[2:]] or [640, 512, 256, 384, 896]
L, P, d = 11, 16, 256
sd = synth.dt_state_dict(3, d=d, ff=2 * d)
m = DecisionTransformerHIP(ctx, sd, activation="relu", fake_bbox_f64=True, precision=prec)
def t(inp, n=30):
    for _ in range(3): m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): o = m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n, o["logits"].cpu().numpy()
for B in Bs:
    inp = {k: torch.from_numpy(v).cuda() for k, v in synth.dt_inputs(3, B, L, P).items()}
    res = {}
    for mode in (0, 1, 2, -1):
        ctx.set_option("dt_split", mode)
        ms, lg = t(inp)
        res[mode] = (ms, lg, ctx.get_option("last_dt_split"), ctx.get_option("last_dt_grid"))
    ctx.set_option("dt_split", -1)
    same = all(np.array_equal(res[0][1], res[k][1]) for k in res)
    print("%s B=%d: " % (prec, B) + ", ".join("split=%d: %.3f ms (%d split, grid %d)" % (k, v[0], v[2], v[3]) for k, v in res.items()) + (", bit-identical" if same else ", MISMATCH"), flush=True)
assert ctx.get_option("dt_status") == 0
