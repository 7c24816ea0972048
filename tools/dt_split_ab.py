#!/usr/bin/env python3
"""Token-split tail of the fused Decision-Transformer kernel, A/B in one process (option dt_split: 0 = one workgroup per track, 1 / 2 = every track split with
one / two tracks per workgroup, -1 = the default policy): python tools/dt_split_ab.py [f32|f16] [d] [P]"""
import numpy as np, torch, sys, os, time
sys.path.insert(0, os.getcwd())
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP
ctx = _lib.Context(0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
L, P, d = 11, int(sys.argv[3]) if len(sys.argv) > 3 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 256
sd = synth.dt_state_dict(3, d=d, ff=2 * d)
m = DecisionTransformerHIP(ctx, sd, activation="relu", fake_bbox_f64=True, precision=prec)
def inputs(B):
    inp = synth.dt_inputs(3, B, L, P)
    return {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
def t(inp, n=20):
    for _ in range(3): m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for B in (32, 64, 85, 86, 100, 128):
    inp = inputs(B)
    ctx.set_option("dt_split", 1); a = t(inp); ctx.set_option("dt_split", 2); a2 = t(inp); g = ctx.get_option("last_dt_grid")
    ctx.set_option("dt_split", 0); b = t(inp)
    print("B=%3d: all split %.3f ms, pairs %.3f ms, one workgroup per track %.3f ms" % (B, a, a2, b), flush=True)
for B in (300, 384, 640, 896):
    inp = inputs(B)
    ctx.set_option("dt_split", -1); a = t(inp); g = ctx.get_option("last_dt_split")
    ctx.set_option("dt_split", 0); b = t(inp)
    print("B=%3d: default (%d tracks split) %.3f ms, no split %.3f ms" % (B, g, a, b), flush=True)
