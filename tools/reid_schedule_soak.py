"""Cross-schedule soak of the ReID extractor: the default schedule (K-split / weights-direct / half-image halo / Gram ... picked per
launch) against the plain tiled schedule at batch sizes around every switch-over; both flavours round stored tensors
identically, so the features must agree to the tolerance of the schedule tests (5e-3, cos >= 0.9998).
  python tools/reid_schedule_soak.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.reid import ReIDEncoderHIP

PLAIN = {"BUSCA_REID_KWAVE_BLOCKS": "0", "BUSCA_REID_HALO_HALF": "0", "BUSCA_REID_GRAM": "0", "BUSCA_REID_FUSE_C1": "0"}
sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 9, 12, 13, 24, 25, 31, 33, 47, 49, 64, 65, 86, 127, 129, 171, 191, 193, 257, 342, 400]
ctx = _lib.Context(0)
sd = synth.reid_state_dict(3)
for k in PLAIN: os.environ.pop(k, None)
dflt = ReIDEncoderHIP(ctx, sd)
worst = 0.0
base = synth.randint_u8(11, "soak", (64, 24, 8, 3)).astype(np.float32)
for n in sizes:
    rep = np.concatenate([base] * ((n + 63) // 64))[:n] + np.arange(n, dtype=np.float32).reshape(n, 1, 1, 1) * 0.37
    up = np.repeat(np.repeat(rep, 16, axis=1), 16, axis=2)
    noise = synth.randint_u8(n, "noise", (n, 384, 128, 3)).astype(np.float32) - 128
    crops = torch.from_numpy(np.clip(up + 0.25 * noise, 0, 255).astype(np.uint8)).cuda()
    for k in PLAIN: os.environ.pop(k, None)
    dflt = ReIDEncoderHIP(ctx, sd)
    a = dflt.forward(crops).cpu().numpy()
    os.environ.update(PLAIN)
    plain = ReIDEncoderHIP(ctx, sd)
    b = plain.forward(crops).cpu().numpy()
    err, cos = float(np.abs(a - b).max()), float((a * b).sum(1).min())
    worst = max(worst, err)
    print("n=%4d  max|d| %.2e  min cos %.6f  %s" % (n, err, cos, "ok" if err <= 5e-3 and cos >= 0.9998 else "FAIL"), flush=True)
    assert np.isfinite(a).all() and err <= 5e-3 and cos >= 0.9998
print("worst %.2e" % worst)
