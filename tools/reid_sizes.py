"""Smoke the ReID extractor over a range of batch sizes (edge cases of every schedule switch): finite, unit-norm, deterministic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.reid import ReIDEncoderHIP
ctx = _lib.Context(0)
m = ReIDEncoderHIP(ctx, synth.reid_state_dict(3))
big = torch.from_numpy(synth.randint_u8(1, "c", (64, 384, 128, 3))).cuda()
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 7, 15, 16, 17, 21, 22, 43, 63, 95, 96, 97, 188, 189, 255, 864, 1500]:
    crops = big.repeat((n + 63) // 64, 1, 1, 1)[:n].contiguous()
    crops[:, :8] = torch.arange(n, device="cuda", dtype=torch.uint8).view(n, 1, 1, 1)   # make the images differ
    a = m.forward(crops).cpu().numpy(); b = m.forward(crops).cpu().numpy()
    ok = np.isfinite(a).all() and np.allclose(np.linalg.norm(a, axis=1), 1, atol=1e-3) and np.array_equal(a, b)
    print("n=%5d finite+unit+deterministic=%s" % (n, ok), flush=True)
    assert ok
