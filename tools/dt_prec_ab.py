#!/usr/bin/env python3
"""Decision-Transformer flavours side by side (f32 exact, x3 float32-equivalent split-fp16, f16): time per launch and distance from the f32 flavour /
the oracle: python tools/dt_prec_ab.py [P] [d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP

ctx = _lib.Context(0)
L, P, d = 11, int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 256
sd = synth.dt_state_dict(3, d=d, ff=2 * d)
models = {p: DecisionTransformerHIP(ctx, sd, activation="relu", fake_bbox_f64=True, precision=p) for p in ("f32", "x3", "f16")}
def inputs(B):
    inp = synth.dt_inputs(3, B, L, P)
    return inp, {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
def t(m, i, n=20):
    for _ in range(3): m.forward(i["mem_feat"], i["can_feat"], i["mem_boxes"], i["can_boxes"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): m.forward(i["mem_feat"], i["can_feat"], i["mem_boxes"], i["can_boxes"])
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
raw, i = inputs(64)
outs = {p: {k: v.cpu().numpy() for k, v in m.forward(i["mem_feat"], i["can_feat"], i["mem_boxes"], i["can_boxes"], want_hidden=True).items()} for p, m in models.items()}
from oracle import dt as odt
ref = odt.dt_forward(sd, odt.DTConfig(d=d, ff=2 * d), **raw, return_all=True)
for p in ("f32", "x3", "f16"):
    print("%s: logits vs oracle %.2e, probs %.2e; vs the f32 flavour logits %.2e hidden %.2e" % (p, np.abs(outs[p]["logits"] - ref["logits"].numpy()).max(),
          np.abs(outs[p]["probs"] - ref["probs"].numpy()).max(), np.abs(outs[p]["logits"] - outs["f32"]["logits"]).max(), np.abs(outs[p]["hidden"] - outs["f32"]["hidden"]).max()))
for B in (32, 128, 256, 512, 640, 2048):
    _, i = inputs(B)
    row = []
    for sp in (0, -1):
        ctx.set_option("dt_split", sp)
        row.append("  ".join("%s %.3f" % (p, t(m, i)) for p, m in models.items()))
    print("B=%4d ms per launch: no split [%s]   default [%s]" % (B, row[0], row[1]), flush=True)
