"""Oracle (CPU checker, test infrastructure only): the Decision-Transformer forward of
/root/reference/busca/network.py:176-244 with the ReID stage replaced by given 512-d features,
restated with plain torch CPU functional ops (float32), no reference modules.

Also used (and only there, besides tests/smoke) as bench.py's `cpu_baseline` ("port").
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import encoding as enc


class DTConfig:
    def __init__(self, d=256, ff=None, nhead=4, nlayers=4, E=512, fake_f64=True, activation="relu", flavour="MEM-SEP-CAN-BAD",
                 encode_sep_as_ref=True):
        self.d, self.nhead, self.nlayers, self.E = d, nhead, nlayers, E
        assert flavour in enc.FLAVOURS, flavour
        self.flavour, self.encode_sep_as_ref = flavour, encode_sep_as_ref     # network.py:103-165, encodings.py:112-146
        self.ff = 2 * d if ff is None else ff
        self.fake_f64 = fake_f64
        # EFFECTIVE activation of the reference is ReLU whatever the YAML says: TransformerEncoder clones
        # the layer with copy.deepcopy (custom_layers.py:44-45,52), deepcopy calls
        # TransformerEncoderLayer.__setstate__ (:24-27), the nn.Module activation lives in `_modules`, not
        # in `__dict__`, so `'activation' not in state` is true and every clone gets `F.relu` in its
        # instance dict, which shadows the registered nn.GELU.  Verified against the imported reference
        # (tests/golden/make_golden.py); "gelu" is kept only for a reference with that quirk fixed.
        self.activation = activation


def _t(sd):
    if getattr(sd, "_oracle_prepared", False):
        return sd
    return {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in sd.items()}


class _Prepared(dict):
    _oracle_prepared = True


def prepare(sd):
    """Convert a state dict to float32 torch tensors once (so timing loops do not re-convert)."""
    return _Prepared(_t(sd))


def assemble_tokens(sd, mem_e, can_e, flavour="MEM-SEP-CAN-BAD"):
    """network.py:103-165: [MEM*L, pair_1 .. pair_P, pair(NON) [, pair(BAD)]] with pair = (SEP, CAN) for MEM-SEP-CAN* and (CAN, SEP)
    for MEM-CAN-SEP*; the learned tokens are appended unscaled (:128-130)."""
    B, P, d = can_e.shape
    sep = sd["sep_token"].view(1, 1, d).expand(B, 1, d)
    cands = [can_e[:, i:i + 1] for i in range(P)] + [sd["non_token"].view(1, 1, d).expand(B, 1, d)]
    if "BAD" in flavour:
        cands.append(sd["bad_token"].view(1, 1, d).expand(B, 1, d))
    toks = [mem_e]
    for c in cands:
        toks += [sep, c] if "MEM-SEP-CAN" in flavour else [c, sep]
    return torch.cat(toks, dim=1)


def can_positions(L, P, flavour="MEM-SEP-CAN-BAD"):
    """network.py:142,154: rows of the candidate tokens (incl. NON [, BAD]) in the assembled sequence."""
    n = P + (2 if "BAD" in flavour else 1)
    return [L + 2 * j + (1 if "MEM-SEP-CAN" in flavour else 0) for j in range(n)]


def mha(x, w_in, b_in, w_out, b_out, nhead):
    """torch.nn.MultiheadAttention (custom_layers.py:12,32-34; batch_first, no mask, eval) spelled out:
    q is pre-scaled by 1/sqrt(head_dim), weights = softmax(q k^T), per-head weights are returned."""
    B, T, d = x.shape
    hd = d // nhead
    qkv = F.linear(x, w_in, b_in)
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(B, T, nhead, hd).transpose(1, 2) * math.sqrt(1.0 / float(hd))
    k = k.view(B, T, nhead, hd).transpose(1, 2)
    v = v.view(B, T, nhead, hd).transpose(1, 2)
    att = torch.softmax(torch.matmul(q, k.transpose(-2, -1)), dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(B, T, d)
    return F.linear(o, w_out, b_out), att


def encoder_layer(x, sd, p, nhead, activation="relu"):
    """custom_layers.py:30-41: post-norm, LayerNorm eps 1e-5; activation see DTConfig (ReLU in effect)."""
    act = F.relu if activation == "relu" else F.gelu
    d = x.shape[-1]
    a, att = mha(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
                 sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], nhead)
    x = F.layer_norm(x + a, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    h = F.linear(act(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                 sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    x = F.layer_norm(x + h, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    return x, att


@torch.no_grad()
def dt_forward(sd, cfg, mem_feat, can_feat, mem_boxes, can_boxes, luts=None, return_all=False):
    """Features [B,L,E],[B,P,E] + ltrb boxes [B,L,4],[B,P,4] -> logits [B,P+2] ([B,P+1] without BAD; pre-softmax, network.py:244).

    return_all -> dict(logits, probs, argmax, hidden[B,T,d], att[list of B,h,T,T], bucket_ids[B,T,3])."""
    sd = _t(sd)
    mem_feat = torch.as_tensor(mem_feat, dtype=torch.float32)
    can_feat = torch.as_tensor(can_feat, dtype=torch.float32)
    mem_boxes = torch.as_tensor(mem_boxes, dtype=torch.float32)
    can_boxes = torch.as_tensor(can_boxes, dtype=torch.float32)
    B, L, _ = mem_feat.shape
    P = can_feat.shape[1]
    d = cfg.d
    scale = float(np.sqrt(d))                                   # network.py:203-204
    mem_e = F.linear(mem_feat, sd["encoder.weight"], sd["encoder.bias"]) * scale
    can_e = F.linear(can_feat, sd["encoder.weight"], sd["encoder.bias"]) * scale
    flavour = getattr(cfg, "flavour", "MEM-SEP-CAN-BAD")
    x = assemble_tokens(sd, mem_e, can_e, flavour)              # [B, T, d]
    ids = enc.token_bucket_ids(mem_boxes, can_boxes, fake_f64=cfg.fake_f64, flavour=flavour,
                               encode_sep_as_ref=getattr(cfg, "encode_sep_as_ref", True))
    if luts is None:
        luts = enc.build_luts(d)
    x = x + enc.encoding_rows(luts, ids[..., 0], ids[..., 1], ids[..., 2], d)  # encodings.py:87-88
    atts = []
    for i in range(cfg.nlayers):
        x, att = encoder_layer(x, sd, "transformer_encoder.layers.%d." % i, cfg.nhead, cfg.activation)
        atts.append(att)
    pos = can_positions(L, P, flavour)                          # network.py:142 CAN rows (incl. NON, BAD)
    out = x[:, pos]
    out = F.layer_norm(out, (d,), sd["decoder.0.weight"], sd["decoder.0.bias"], 1e-5)
    logits = F.linear(out, sd["decoder.1.weight"], sd["decoder.1.bias"])[:, :, 0]
    if not return_all:
        return logits
    probs = torch.softmax(logits, dim=-1)                        # network.py:96,403
    return dict(logits=logits, probs=probs, argmax=probs.argmax(dim=-1), hidden=x, att=atts, bucket_ids=ids)
