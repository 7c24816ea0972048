#!/usr/bin/env python3
"""Fused feed-forward block of the layer-wise path against the two-kernel form (option dtl_ffn): max |delta| of logits / hidden."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP
ctx = _lib.Context(0)
ctx.set_option("dt_tiled", 1)
for d, B, P, nl in ((256, 32, 16, 1), (256, 32, 16, 4), (256, 8, 16, 1), (256, 32, 5, 1), (512, 32, 5, 4)):
    for prec in ("f16", "f32"):
        m = DecisionTransformerHIP(ctx, synth.dt_state_dict(7, d=d, ff=2 * d, nlayers=nl), precision=prec)
        inp = {k: torch.from_numpy(v).cuda() for k, v in synth.dt_inputs(7, B, 11, P).items()}
        res = {}
        for f in (1, 0):
            ctx.set_option("dtl_ffn", f)
            o = m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"], want_hidden=True)
            torch.cuda.synchronize()
            res[f] = {k: v.cpu().numpy() for k, v in o.items()}
        dh = np.abs(res[1]["hidden"] - res[0]["hidden"])
        bad = dh > 1e-3
        print("d=%d B=%d P=%d layers %d %s: max |d hidden| %.3e, %d of %d elements off by > 1e-3; bad features %s; bad tokens %s; bad tracks %s" % (
            d, B, P, nl, prec, dh.max(), bad.sum(), bad.size, np.unique(np.nonzero(bad)[2])[:12], np.unique(np.nonzero(bad)[1])[:12], np.unique(np.nonzero(bad)[0])[:12]))
ctx.set_option("dtl_ffn", 1)
