"""Repeatability soak of the ReID extractor: every pass of the same batch must return the SAME BITS (the statistics chain hands
results between workgroups through arrival counters - a race there shows up as a run-to-run difference).  Plain and weighted
(deduplicated) batches at sizes on both sides of every schedule switch, two streams alternating.
  python tools/reid_repeat_soak.py [reps=30] [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.reid import ReIDEncoderHIP

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sizes = [int(a) for a in sys.argv[2:]] or [1, 8, 24, 40, 88, 150, 257, 352, 512]
ctx = _lib.Context(0)
m = ReIDEncoderHIP(ctx, synth.reid_state_dict(3))
side = torch.cuda.Stream()
bad = 0
for n in sizes:
    base = synth.randint_u8(n, "rs", (n, 24, 8, 3)).astype(np.float32)
    up = np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)
    noise = synth.randint_u8(n + 1, "rsn", (n, 384, 128, 3)).astype(np.float32) - 128
    crops = torch.from_numpy(np.clip(up + 0.25 * noise, 0, 255).astype(np.uint8)).cuda()
    wts = (1 + (np.arange(n) % 7)).astype(np.float32)
    first = first_w = None
    for r in range(reps):
        if r % 2:
            with torch.cuda.stream(side):
                a = m.forward(crops)
                w = m.forward(crops, weights=wts) if n > 1 else None
            side.synchronize()
        else:
            a = m.forward(crops)
            w = m.forward(crops, weights=wts) if n > 1 else None
            torch.cuda.synchronize()
        a = a.cpu().numpy(); w = None if w is None else w.cpu().numpy()
        if first is None:
            first, first_w = a, w
            assert np.isfinite(a).all()
        else:
            if not np.array_equal(a, first): bad += 1; print("n=%d rep %d: plain pass differs by %.3e" % (n, r, np.abs(a - first).max()))
            if w is not None and not np.array_equal(w, first_w): bad += 1; print("n=%d rep %d: weighted pass differs by %.3e" % (n, r, np.abs(w - first_w).max()))
    print("n=%4d  %d passes%s identical" % (n, reps, "" if first_w is None else " (+ weighted)"), flush=True)
print("differences:", bad)
sys.exit(1 if bad else 0)
