"""Oracle (CPU checker, test infrastructure only): the ReID crop feature extractor of
/root/reference/busca/reid/resnet.py (ResNet-50 Bottleneck [3,4,6,3], pool='max', red=4) as driven by
/root/reference/busca/network.py:510-575 - i.e. with every BatchNorm in TRAIN mode (batch statistics)
at inference (network.py:553-556).  Plain torch CPU functional ops, float32, no reference modules.
"""
import numpy as np
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)


def _bn(x, sd, name):
    # nn.BatchNorm2d defaults, training=True -> biased batch variance, eps 1e-5 (resnet.py:153-154)
    return F.batch_norm(x, None, None, sd[name + ".weight"], sd[name + ".bias"], True, 0.1, 1e-5)


def _bottleneck(x, sd, p, stride, has_ds):
    """resnet.py:108-128."""
    out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2"))
    out = _bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
    if has_ds:
        x = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1")
    return F.relu(out + x)


@torch.no_grad()
def reid_forward(sd, x, return_stages=False, dtype=torch.float32):
    """x: [n,3,384,128] float32 (RGB, CHW, normalised as network.py:470-478,397) -> [n,512] L2-normalised
    features (resnet.py:266-322 with output_option='plain').  One call == one BN batch.
    dtype=torch.float64 evaluates the same network in double precision (used once, in the build container, to tell the float32
    round-off of the reference's own CPU kernels from a defect: tests/golden/make_golden.py reid_cfg4_f64)."""
    sd = {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in sd.items()}
    x = torch.as_tensor(x).to(dtype)
    stages = {}
    x = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    stages["stem"] = x
    for li, nblk in enumerate(LAYERS):
        for b in range(nblk):
            stride = 2 if (li > 0 and b == 0) else 1
            x = _bottleneck(x, sd, "layer%d.%d." % (li + 1, b), stride, b == 0)
        stages["layer%d" % (li + 1)] = x
    fc7 = torch.flatten(F.adaptive_max_pool2d(x, 1), 1)
    stages["pool"] = fc7
    fc7 = F.linear(fc7, sd["red.weight"], sd["red.bias"])
    feats = F.normalize(fc7, p=2, dim=1)
    return (feats, stages) if return_stages else feats


def crops_to_reid_input(crops_u8_bgr):
    """u8 [n,384,128,3] BGR -> float32 [n,3,384,128] RGB normalised (network.py:470-478, 397).
    The reference builds this tensor as `batch[..., [2, 1, 0]].permute(0, 1, 4, 2, 3)` and `.view(-1, C, H, W)`
    (network.py:397,188): an NHWC buffer seen as NCHW, i.e. torch's channels_last memory format, which the conv /
    batch-norm kernels then keep for the whole ResNet.  The same strides are produced here; with them the oracle's
    features are BIT-IDENTICAL to the reference's on the same host (a contiguous NCHW input differs by ~1e-5)."""
    from .geometry import normalize_bgr
    x = torch.from_numpy(normalize_bgr(np.asarray(crops_u8_bgr))).float()
    return x[..., [2, 1, 0]].permute(0, 3, 1, 2)
