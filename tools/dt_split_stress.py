#!/usr/bin/env python3
"""Stress of the token-split exchange: many back-to-back forwards of changing track counts and flavours on one context, forced and default split, outputs
compared with the unsplit flavour every time; dt_status must stay 0.  python tools/dt_split_stress.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ctx = _lib.Context(0)
rng = np.random.default_rng(0)
models = {}
for prec in ("f32", "x3"):
    for d, P in ((256, 16), (512, 5), (64, 24)):
        models[(prec, d, P)] = DecisionTransformerHIP(ctx, synth.dt_state_dict(d + P, d=d, ff=2 * d), activation="relu", precision=prec)
inputs = {}
def inp(B, P):
    if (B, P) not in inputs:
        inputs[(B, P)] = {k: torch.from_numpy(v).cuda() for k, v in synth.dt_inputs(B + P, B, 11, P).items()}
    return inputs[(B, P)]
t0 = time.time(); bad = 0
keys = list(models)
for it in range(iters):
    prec, d, P = keys[rng.integers(len(keys))]
    B = int(rng.choice([1, 2, 7, 31, 32, 33, 85, 86, 100, 128, 129, 170, 300, 385, 640]))
    m, i = models[(prec, d, P)], inp(B, P)
    ctx.set_option("dt_split", 0)
    ref = m.forward(i["mem_feat"], i["can_feat"], i["mem_boxes"], i["can_boxes"])["logits"].clone()
    mode = int(rng.choice([-1, 1, 2]))
    ctx.set_option("dt_split", mode)
    outs = [m.forward(i["mem_feat"], i["can_feat"], i["mem_boxes"], i["can_boxes"])["logits"] for _ in range(3)]     # back to back: flags and parity buffers reused at once
    ok = all(torch.equal(o, ref) for o in outs)
    st = ctx.get_option("dt_status")
    if not ok or st != 0:
        bad += 1
        print("MISMATCH it=%d %s d=%d P=%d B=%d mode=%d status=%d split=%d" % (it, prec, d, P, B, mode, st, ctx.get_option("last_dt_split")))
ctx.set_option("dt_split", -1)
torch.cuda.synchronize()
print("%d iterations, %d mismatches, %.1f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
