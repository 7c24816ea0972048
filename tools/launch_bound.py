"""Is a small ReID batch bound by the host's launch rate?  Compares the time busca_reid_forward takes to RETURN with
the time until the stream has drained."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from busca_amd import _lib, synth
from busca_amd.reid import ReIDEncoderHIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = _lib.Context(0)
m = ReIDEncoderHIP(ctx, synth.reid_state_dict(3))
crops = torch.from_numpy(synth.randint_u8(1, "c", (n, 384, 128, 3))).cuda()
for _ in range(3): m.forward(crops)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.forward(crops); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
enq.sort(); tot.sort()
print("n=%d: enqueue p50 %.3f ms, until drained p50 %.3f ms" % (n, enq[10] * 1e3, tot[10] * 1e3))
