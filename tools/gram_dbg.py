import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd import _lib, synth
from busca_amd.reid import ReIDEncoderHIP
ctx = _lib.Context(0)
sd = synth.reid_state_dict(3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
def smooth(seed, n):
    base = synth.randint_u8(seed, "crops", (n, 24, 8, 3)).astype(np.float32)
    up = np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)
    noise = synth.randint_u8(seed, "noise", (n, 384, 128, 3)).astype(np.float32) - 128
    return np.clip(up + 0.25 * noise, 0, 255).astype(np.uint8)
crops = smooth(300 + n, n) if len(sys.argv) <= 2 else synth.randint_u8(5, "c", (n, 384, 128, 3))
ref = ReIDEncoderHIP(ctx, sd, precision="f32").forward(crops).cpu().numpy()
out = {}
for mode in ("0", "2", "1"):
    os.environ["BUSCA_REID_GRAM"] = mode
    out[mode] = ReIDEncoderHIP(ctx, sd).forward(crops).cpu().numpy()
    print("mode", mode, "vs f32 flavour: max |d| %.5f cos min %.6f" % (np.abs(out[mode] - ref).max(), (out[mode] * ref).sum(1).min()))
for mode in ("2", "1"):
    print("mode", mode, "max |d| vs direct", np.abs(out[mode] - out["0"]).max(), "cos min", (out[mode] * out["0"]).sum(1).min())
