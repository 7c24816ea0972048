"""Reference state_dict  <->  flat float32 weight blobs for libbusca_hip.so.

Key names are the reference's (busca/network.py:45-94); the order of the blob is
the one documented in include/busca_hip.h."""
import numpy as np


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def dt_blob_keys(nlayers):
    keys = ["encoder.weight", "encoder.bias", "sep_token", "non_token", "bad_token"]
    for i in range(nlayers):
        p = "transformer_encoder.layers.%d." % i
        keys += [p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", p + "self_attn.out_proj.weight",
                 p + "self_attn.out_proj.bias", p + "linear1.weight", p + "linear1.bias", p + "linear2.weight",
                 p + "linear2.bias", p + "norm1.weight", p + "norm1.bias", p + "norm2.weight", p + "norm2.bias"]
    keys += ["decoder.0.weight", "decoder.0.bias", "decoder.1.weight", "decoder.1.bias"]
    return keys


def dt_blob(state_dict, nlayers):
    """Concatenate the Decision-Transformer tensors of a reference state_dict into the C-ABI blob.  A model without the BAD token
    (input_flavour without "-BAD", network.py:66-69) has no `bad_token`: its slot is filled with zeros (BUSCA_LAYOUT_NO_BAD ignores it)."""
    missing = [k for k in dt_blob_keys(nlayers) if k not in state_dict and k != "bad_token"]
    if missing:
        raise KeyError("state_dict lacks %s" % missing[:4])
    d = _np(state_dict["encoder.weight"]).shape[0]
    return np.concatenate([_np(state_dict[k]).ravel() if k in state_dict else np.zeros(d, np.float32) for k in dt_blob_keys(nlayers)])


def dt_dims(state_dict):
    """(d, ff, nlayers, E) from tensor shapes."""
    d, E = _np(state_dict["encoder.weight"]).shape
    ff = _np(state_dict["transformer_encoder.layers.0.linear1.weight"]).shape[0]
    nl = 0
    while "transformer_encoder.layers.%d.linear1.weight" % nl in state_dict:
        nl += 1
    return d, ff, nl, E


def encoding_luts(d):
    """The three per-axis fp16 LUTs that the reference's 211x211x61xd table (busca/encodings.py:23-32)
    factors into: interleaved sin/cos of pos / 10000^(2j/c), c = 2*ceil(d/6), float32 torch CPU math
    (the ops the third-party positional_encodings 6.0.3 module runs) then fp16.  Returns uint16 bit
    patterns [211,c], [211,c], [61,c] and c."""
    import torch
    c = int(np.ceil(d / 6) * 2)
    c += c % 2
    inv_freq = 1.0 / (10000 ** (torch.arange(0, c, 2).float() / c))

    def lut(n):
        s = torch.arange(n, dtype=torch.float32)[:, None] * inv_freq[None, :]
        e = torch.stack((s.sin(), s.cos()), dim=-1).flatten(-2, -1).to(torch.float16)
        return np.ascontiguousarray(e.numpy().view(np.uint16))

    return lut(211), lut(211), lut(61), c


def reid_conv_names():
    """(conv key, bn key) in forward order - the order of the ReID blob (busca/reid/resnet.py:169-252)."""
    names = [("conv1", "bn1")]
    for li, nblk in enumerate((3, 4, 6, 3)):
        for b in range(nblk):
            p = "layer%d.%d." % (li + 1, b)
            names += [(p + "conv1", p + "bn1"), (p + "conv2", p + "bn2"), (p + "conv3", p + "bn3")]
            if b == 0:
                names.append((p + "downsample.0", p + "downsample.1"))
    return names


def reid_blob(state_dict, prefix=""):
    """ReID blob: per conv (forward order) weight[Cout,Cin,k,k], bn.weight, bn.bias; then red.weight, red.bias.
    `prefix` is 'reid_encoder.model.' for a full BUSCA checkpoint.  BN running statistics and the unused
    classifier `fc` are not part of the blob (train-mode BN never reads them, network.py:553-556)."""
    parts = []
    for conv, bn in reid_conv_names():
        parts += [_np(state_dict[prefix + conv + ".weight"]).ravel(), _np(state_dict[prefix + bn + ".weight"]).ravel(),
                  _np(state_dict[prefix + bn + ".bias"]).ravel()]
    parts += [_np(state_dict[prefix + "red.weight"]).ravel(), _np(state_dict[prefix + "red.bias"]).ravel()]
    return np.concatenate(parts)


def dt_unblob(blob, d, ff, nlayers, E=512):
    """Inverse of dt_blob: flat float32 blob -> OrderedDict with the reference's key names and torch shapes."""
    from collections import OrderedDict
    shapes = {"encoder.weight": (d, E), "encoder.bias": (d,), "sep_token": (d,), "non_token": (d,), "bad_token": (d,),
              "decoder.0.weight": (d,), "decoder.0.bias": (d,), "decoder.1.weight": (1, d), "decoder.1.bias": (1,)}
    for i in range(nlayers):
        p = "transformer_encoder.layers.%d." % i
        shapes.update({p + "self_attn.in_proj_weight": (3 * d, d), p + "self_attn.in_proj_bias": (3 * d,),
                       p + "self_attn.out_proj.weight": (d, d), p + "self_attn.out_proj.bias": (d,),
                       p + "linear1.weight": (ff, d), p + "linear1.bias": (ff,), p + "linear2.weight": (d, ff), p + "linear2.bias": (d,),
                       p + "norm1.weight": (d,), p + "norm1.bias": (d,), p + "norm2.weight": (d,), p + "norm2.bias": (d,)})
    out, off = OrderedDict(), 0
    blob = np.asarray(blob, dtype=np.float32)
    for k in dt_blob_keys(nlayers):
        n = int(np.prod(shapes[k]))
        out[k] = blob[off:off + n].reshape(shapes[k]).copy()
        off += n
    assert off == blob.size
    return out


X3_OPERAND_LIMIT = 1023.5          # reid_x3.hip.inc: staged activations are clamped to |x| * 2^6 <= 65504 (the fp16 range)


def x3_activation_bound(state_dict, prefix="", sigmas=48.0):
    """Upper bound of what the split-fp16 (x3) ReID flavour stages as an MFMA operand, from the checkpoint alone.  Every conv input is either
    relu(gamma * z + beta) with z the batch-normalised producer output (|z| <= `sigmas` standard deviations: batch statistics bound the largest
    outlier by sqrt(pixels), real activations stay far below) or a residual-stream value = a sum of such terms (one per bottleneck of the layer plus
    the downsample branch).  Returns (bound, where): the largest such bound over the network and the BatchNorm it comes from.  A bound above
    X3_OPERAND_LIMIT means the kernel's clamp could clip activations SILENTLY (round-4 advisor finding); ReIDEncoderHIP warns and names `f32`."""
    worst, where = 0.0, None
    layer_sum = {}
    for k, v in state_dict.items():
        if not (k.startswith(prefix) and k.endswith(".weight")):
            continue
        name = k[len(prefix):-len(".weight")]
        if not (name == "bn1" or ".bn" in name or name.endswith("downsample.1")):
            continue
        b = state_dict.get(prefix + name + ".bias")
        if b is None:
            continue
        g, b = np.abs(_np(v)), np.abs(_np(b))
        bound = float((g * sigmas + b).max())
        if bound > worst:
            worst, where = bound, name
        if name.endswith("bn3") or name.endswith("downsample.1"):        # terms of a layer's residual stream
            layer = name.split(".")[0]
            layer_sum[layer] = layer_sum.get(layer, 0.0) + bound
    for layer, tot in layer_sum.items():
        if tot > worst:
            worst, where = tot, layer + " residual stream"
    return worst, where
