"""Single-step latency of the two Decision-Transformer paths (one launch on B of the 256 CUs against 23 chip-wide launches): python tools/dt_latency_paths.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.dt import DecisionTransformerHIP
ctx = _lib.Context(0)
for d, P in ((256, 16), (512, 5)):
    for prec in ("f32", "f16"):
        m = DecisionTransformerHIP(ctx, synth.dt_state_dict(7, d=d, ff=2 * d), precision=prec)
        for B in (8, 32, 64, 128):
            inp = {k: torch.from_numpy(v).cuda() for k, v in synth.dt_inputs(7, B, 11, P).items()}
            m.reserve(B, 11, P)
            res = []
            for tiled in (0, 1):
                ctx.set_option("dt_tiled", tiled)
                for _ in range(5): m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
                torch.cuda.synchronize()
                ts = []
                for _ in range(50):
                    t = time.perf_counter(); m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"]); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
                res.append(np.median(ts) * 1e3)
            ctx.set_option("dt_tiled", 0)
            print("d=%d P=%d %s B=%3d: fused %.3f ms, layer-wise %.3f ms" % (d, P, prec, B, res[0], res[1]), flush=True)
