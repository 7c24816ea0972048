#!/usr/bin/env python3
"""Generate the committed golden vectors by running THE REFERENCE ITSELF (imported from
/root/reference, build container only) on seeded inputs.  Run from the repo root:

    python tests/golden/make_golden.py            # all sets
    python tests/golden/make_golden.py dt geom    # selected sets

The reference needs three import shims here: `cv2` and
`positional_encodings` stubs from oracle/ref_shims/, and a no-network patch of
busca.reid.resnet.load_state_dict_from_url.  Weights and inputs come from the portable PRNG in
busca_amd/synth.py, so the fixtures hold only seeds/shapes and the reference's OUTPUTS.
Nothing of the reference (source or bytecode) is copied; /root/reference never travels.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from busca_amd import synth  # noqa: E402


def import_reference():
    """Put the shims first on sys.path, then the reference; patch out the ImageNet download."""
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shims"))
    import busca.reid.resnet as ref_resnet
    ref_resnet.load_state_dict_from_url = lambda *a, **k: {}
    import busca.network as ref_network
    import busca.tracking as ref_tracking
    import busca.encodings as ref_encodings
    return ref_network, ref_tracking, ref_encodings


def ref_args(d, ff, flavour="MEM-SEP-CAN-BAD"):
    return types.SimpleNamespace(
        num_layer=4, nhead=4, dim_embedding=512, trans_dim=d, ff_size=ff, activation="gelu", dropout_p=0.1,
        input_flavour=flavour, output_flavour="CAN", encode_separator_as_reference=True,
        encode_special_tokens=False, reid_weights_file="no", device=torch.device("cpu"))


F32_MIN = float(np.finfo(np.float32).min)


def set_fake_dtype(model, f64):
    """Emulate the reference's pinned numpy 1.23.5 (float64 sentinel) or keep numpy>=2 (float32)."""
    if f64:
        model.pos_encoder.distant_fake_bbox = torch.tensor(
            [F32_MIN, F32_MIN, -F32_MIN / 100.0, -F32_MIN / 100.0], dtype=torch.float64)
    else:
        m = np.float32(F32_MIN)
        model.pos_encoder.distant_fake_bbox = torch.from_numpy(
            np.array([m, m, -m / np.float32(100.0), -m / np.float32(100.0)], dtype=np.float32))


_MODELS = {}


def build_ref_model(ref_network, d, ff):
    """One reference BUSCA per (d, ff): building it materialises the 211x211x61xd table (slow, GBs)."""
    key = (d, ff)
    if key not in _MODELS:
        m = ref_network.BUSCA(ref_args(d, ff)).eval()
        _MODELS[key] = m
    return _MODELS[key]


def load_dt_weights(model, sd):
    full = model.state_dict()
    for k, v in sd.items():
        assert k in full and tuple(full[k].shape) == tuple(v.shape), k
        full[k] = torch.from_numpy(np.asarray(v))
    model.load_state_dict(full)


def run_ref_dt(model, inp):
    """Drive BUSCA.forward with precomputed 512-d features."""
    B, L, E = inp["mem_feat"].shape
    P = inp["can_feat"].shape[1]
    feats = [torch.from_numpy(inp["mem_feat"]).reshape(B * L, E), torch.from_numpy(inp["can_feat"]).reshape(B * P, E)]
    calls = []

    def fake_reid(x):
        calls.append(1)
        return None, feats[len(calls) - 1]

    model.reid_encoder.forward = fake_reid
    dummy_m = torch.zeros(B, L, 3, 1, 1)
    dummy_c = torch.zeros(B, P, 3, 1, 1)
    with torch.no_grad():
        logits = model.forward(dummy_m, dummy_c, memory_bboxes=torch.from_numpy(inp["mem_boxes"]),
                               candidates_bboxes=torch.from_numpy(inp["can_boxes"]), return_att=True, return_logits=True)
        probs = model.softmax(logits)
    return dict(logits=logits.numpy(), probs=probs.numpy(), argmax=probs.argmax(-1).numpy(),
                can_hidden=model.logits.numpy(), mem_hidden_mean=model.mem_logits.numpy(),
                att=np.stack([a.numpy() for a in model.attentions]))


# ------------------------------------------------------------------------------------------------
# sets
# ------------------------------------------------------------------------------------------------

def make_dt(ref):
    ref_network, _, _ = ref
    cases = [
        # name, d, ff, B, L, P, seed
        ("dt_d64_b4_p5", 64, 128, 4, 11, 5, 11),
        ("dt_d64_b3_p16", 64, 128, 3, 11, 16, 12),
        ("dt_d256_b8_p16", 256, 512, 8, 11, 16, 7),     # BASELINE configs[0]
        ("dt_d256_b32_p16", 256, 512, 32, 11, 16, 8),   # north-star shape
    ]
    for name, d, ff, B, L, P, seed in cases:
        model = build_ref_model(ref_network, d, ff)
        sd = synth.dt_state_dict(seed, d=d, ff=ff)
        load_dt_weights(model, sd)
        inp = synth.dt_inputs(seed, B, L, P, sentinel_every=4 if B <= 8 else 16)
        out = {}
        for mode, f64 in (("f64", True), ("f32", False)):
            set_fake_dtype(model, f64)
            r = run_ref_dt(model, inp)
            for k, v in r.items():
                if k == "att" and B * P > 64:
                    continue  # keep the fixtures small
                out["%s_%s" % (k, mode)] = v
        np.savez_compressed(os.path.join(OUT, name + ".npz"), d=d, ff=ff, B=B, L=L, P=P, seed=seed, **out)
        print("wrote", name, {k: v.shape for k, v in out.items() if k.startswith("logits")})


def main():
    which = set(sys.argv[1:]) or {"dt", "enc", "geom", "assoc", "reid"}
    ref = import_reference()
    if "dt" in which:
        make_dt(ref)
    for name in ("enc", "geom", "assoc", "reid"):
        fn = globals().get("make_" + name)
        if name in which and fn is not None:
            fn(ref)


if __name__ == "__main__":
    main()
