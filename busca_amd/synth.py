"""Portable synthetic weights and inputs for tests, fixtures and bench.

Everything here is derived from a counter-based splitmix64 hash, so the same
(seed, name) pair yields bit-identical float32 arrays on any numpy version and
on any machine.  Golden fixtures therefore only store seeds plus the
reference's outputs, never weights.

The state-dict key space mirrors the reference's (busca/network.py:45-94,
busca/custom_layers.py:9-22, busca/reid/resnet.py:137-193).
"""
import zlib
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream_key(seed, name):
    h = np.uint64(zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        k = (np.uint64(seed & 0xFFFFFFFF) << np.uint64(32)) ^ h
        return _splitmix64(np.array([k], dtype=np.uint64))[0]


def uniform(seed, name, shape, lo=-1.0, hi=1.0):
    """float32 array of `shape`, uniform in [lo, hi) with 24 random bits per value."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = _stream_key(seed, name)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key
    bits = _splitmix64(idx) >> np.uint64(40)  # 24 bits
    u = bits.astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, name, shape, std=1.0):
    """float32 approx-normal (sum of 4 uniforms, variance-matched); portable, not exact Gaussian."""
    acc = np.zeros(int(np.prod(shape)) if len(shape) else 1, dtype=np.float64)
    for k in range(4):
        acc += uniform(seed, "%s#n%d" % (name, k), (acc.size,)).astype(np.float64)
    acc *= np.sqrt(3.0 / 4.0) * std  # var(U(-1,1)) = 1/3 -> sum of 4 has var 4/3
    return acc.astype(np.float32).reshape(shape)


def randint_u8(seed, name, shape):
    n = int(np.prod(shape))
    key = _stream_key(seed, name)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key
    return (_splitmix64(idx) >> np.uint64(56)).astype(np.uint8).reshape(shape)


# ----------------------------------------------------------------------------------------------
# Decision-Transformer weights (reference key names, busca/network.py:45-94)
# ----------------------------------------------------------------------------------------------

def dt_state_dict(seed, d=256, ff=None, nlayers=4, E=512, flavour="MEM-SEP-CAN-BAD", gain=1.0):
    """OrderedDict name -> float32 ndarray for the non-ReID part of the reference model."""
    ff = 2 * d if ff is None else ff
    sd = OrderedDict()

    def lin(name, out_f, in_f, g=1.0):
        a = g * gain / np.sqrt(in_f)
        sd[name + ".weight"] = uniform(seed, name + ".weight", (out_f, in_f), -a, a)
        sd[name + ".bias"] = uniform(seed, name + ".bias", (out_f,), -0.1, 0.1)

    def ln(name):
        sd[name + ".weight"] = 1.0 + 0.2 * uniform(seed, name + ".weight", (d,))
        sd[name + ".bias"] = 0.1 * uniform(seed, name + ".bias", (d,))

    if "SEP" in flavour:
        sd["sep_token"] = normal(seed, "sep_token", (d,))
    sd["non_token"] = normal(seed, "non_token", (d,))
    if "BAD" in flavour:
        sd["bad_token"] = normal(seed, "bad_token", (d,))
    for i in range(nlayers):
        p = "transformer_encoder.layers.%d." % i
        a = 1.7 * gain / np.sqrt(d)
        sd[p + "self_attn.in_proj_weight"] = uniform(seed, p + "in_proj_weight", (3 * d, d), -a, a)
        sd[p + "self_attn.in_proj_bias"] = uniform(seed, p + "in_proj_bias", (3 * d,), -0.1, 0.1)
        lin(p + "self_attn.out_proj", d, d, 1.7)
        lin(p + "linear1", ff, d, 1.7)
        lin(p + "linear2", d, ff, 1.7)
        ln(p + "norm1")
        ln(p + "norm2")
    lin("encoder", d, E, 1.7)
    ln("decoder.0")
    sd["decoder.1.weight"] = uniform(seed, "decoder.1.weight", (1, d), -0.3, 0.3)
    sd["decoder.1.bias"] = uniform(seed, "decoder.1.bias", (1,), -0.1, 0.1)
    return sd


# ----------------------------------------------------------------------------------------------
# ReID ResNet-50 weights (torchvision naming, busca/reid/resnet.py:137-264; red=4 -> 2048->512)
# ----------------------------------------------------------------------------------------------

RESNET50_LAYERS = (3, 4, 6, 3)


def reid_conv_specs():
    """List of (name, cout, cin, k, stride, pad) for the 53 convs in forward order, plus BN names."""
    specs = [("conv1", 64, 3, 7, 2, 3, "bn1")]
    inplanes = 64
    for li, (nblk, planes) in enumerate(zip(RESNET50_LAYERS, (64, 128, 256, 512))):
        stride = 1 if li == 0 else 2
        for b in range(nblk):
            p = "layer%d.%d." % (li + 1, b)
            s = stride if b == 0 else 1
            specs.append((p + "conv1", planes, inplanes, 1, 1, 0, p + "bn1"))
            specs.append((p + "conv2", planes, planes, 3, s, 1, p + "bn2"))
            specs.append((p + "conv3", planes * 4, planes, 1, 1, 0, p + "bn3"))
            if b == 0:
                specs.append((p + "downsample.0", planes * 4, inplanes, 1, s, 0, p + "downsample.1"))
            inplanes = planes * 4
    return specs


def reid_state_dict(seed, with_fc=False):
    """float32 ReID weights: kaiming-like convs, BN affine near (1, 0), red Linear 2048->512."""
    sd = OrderedDict()
    for (name, cout, cin, k, _s, _p, bn) in reid_conv_specs():
        fan_out = cout * k * k
        a = np.sqrt(3.0) * np.sqrt(2.0 / fan_out)
        sd[name + ".weight"] = uniform(seed, "reid." + name, (cout, cin, k, k), -a, a)
        sd[bn + ".weight"] = 1.0 + 0.2 * uniform(seed, "reid." + bn + ".w", (cout,))
        sd[bn + ".bias"] = 0.2 * uniform(seed, "reid." + bn + ".b", (cout,))
    a = 1.0 / np.sqrt(2048.0)
    sd["red.weight"] = uniform(seed, "reid.red.w", (512, 2048), -a, a)
    sd["red.bias"] = uniform(seed, "reid.red.b", (512,), -a, a)
    if with_fc:
        sd["fc.weight"] = uniform(seed, "reid.fc.w", (299, 512), -0.04, 0.04)
        sd["fc.bias"] = uniform(seed, "reid.fc.b", (299,), -0.04, 0.04)
    return sd


# ----------------------------------------------------------------------------------------------
# Synthetic association inputs (SURVEY.md section 8d "Synthetic inputs")
# ----------------------------------------------------------------------------------------------

F32_MIN = float(np.finfo(np.float32).min)


def missing_ltrb_f32():
    """The padded-candidate sentinel after the reference's ltwh->ltrb step, as float32 values.

    busca/tracking.py:11-12 builds [f32min, f32min, -f32min/100, -f32min/100] (ltwh);
    busca/network.py:389,393 casts it to float32 and adds l,t to w,h.
    """
    ltwh = np.array([F32_MIN, F32_MIN, -F32_MIN / 100.0, -F32_MIN / 100.0], dtype=np.float64).astype(np.float32)
    out = ltwh.copy()
    out[2:] += out[:2]
    return out


def dt_inputs(seed, B, L=11, P=16, E=512, sentinel_every=16):
    """Features + ltrb boxes for one association step (all float32).

    Returns dict(mem_feat[B,L,E], can_feat[B,P,E], mem_boxes[B,L,4], can_boxes[B,P,4]).
    One sentinel-padded candidate per `sentinel_every` tracks exercises the E3 path.
    """
    def feats(name, n):
        x = normal(seed, name, (B, n, E)).astype(np.float64)
        x /= np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), 1e-12)
        return x.astype(np.float32)

    cx = uniform(seed, "cx", (B,), 0.0, 1920.0)
    cy = uniform(seed, "cy", (B,), 0.0, 1080.0)
    h = uniform(seed, "h", (B,), 40.0, 400.0)
    w = h * uniform(seed, "ar", (B,), 0.3, 0.5)

    def boxes(name, n, jitter, walk):
        j = normal(seed, name + ".j", (B, n, 2)) * jitter * w[:, None, None]
        if walk:
            j = np.cumsum(j[:, ::-1, :], axis=1)[:, ::-1, :]
            j = j - j[:, -1:, :]  # the walk ends at the reference box
        s = np.exp(0.15 * normal(seed, name + ".s", (B, n, 2)))
        if walk:
            s[:, -1, :] = 1.0
        bw = w[:, None] * s[..., 0]
        bh = h[:, None] * s[..., 1]
        bx = cx[:, None] + j[..., 0]
        by = cy[:, None] + j[..., 1]
        return np.stack([bx - bw / 2, by - bh / 2, bx + bw / 2, by + bh / 2], axis=-1).astype(np.float32)

    out = dict(
        mem_feat=feats("mem_feat", L),
        can_feat=feats("can_feat", P),
        mem_boxes=boxes("mem_boxes", L, 0.15, True),
        can_boxes=boxes("can_boxes", P, 0.5, False),
    )
    if sentinel_every:
        miss = missing_ltrb_f32()
        for b in range(0, B, sentinel_every):
            out["can_boxes"][b, P - 1] = miss
            out["can_feat"][b, P - 1] = 0.0
    return out
