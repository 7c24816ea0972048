#!/usr/bin/env python3
"""Generate the committed golden vectors by running THE REFERENCE ITSELF (imported from
/root/reference, build container only) on seeded inputs.  Run from the repo root:

    python tests/golden/make_golden.py            # all sets
    python tests/golden/make_golden.py dt geom    # selected sets
    python tests/golden/make_golden.py reid_cfg4                              # round-3 set: 1 408-crop BatchNorm batches (~30 GB RAM, minutes)
    python tests/golden/make_golden.py dt512 assoc512 assoc_select reid_big   # round-2 sets (dt512/assoc512 need ~20 GB RAM;
                                                                              # not part of the default "all")

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
    _run_dt_cases(ref_network, cases)


def make_dt512(ref):
    """cfgR, the shipped model shape (config/*/*/*.yml: d=512, ff=1024, P=5).  Building the reference model at d=512
    materialises its 211x211x61x512 table (~17 GB of transient host RAM, about a minute)."""
    _run_dt_cases(ref[0], [("dt_d512_b32_p5", 512, 1024, 32, 11, 5, 9)])


def _run_dt_cases(ref_network, cases):
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


DT_FLAVOUR_CASES = [
    # name, input_flavour, encode_separator_as_reference, encode_special_tokens, B, L, P, seed   (d = 64, ff = 128)
    ("sep_can", "MEM-SEP-CAN", True, False, 3, 11, 5, 31),
    ("can_sep_bad", "MEM-CAN-SEP-BAD", True, False, 3, 11, 5, 32),
    ("can_sep", "MEM-CAN-SEP", True, False, 4, 6, 16, 33),
    ("sep_can_bad_sepcan", "MEM-SEP-CAN-BAD", False, False, 3, 11, 5, 34),
    ("can_sep_bad_sepcan", "MEM-CAN-SEP-BAD", False, False, 3, 4, 7, 35),
    ("sep_can_sepcan", "MEM-SEP-CAN", False, False, 3, 11, 5, 36),
]


def make_dt_flavours(ref):
    """The non-shipped token layouts of network.py:103-165 / encodings.py:112-146, from the REFERENCE itself (d = 64).  Also records
    what the reference does with the CLS-* flavours and with encode_special_tokens (see the notes saved in the file)."""
    ref_network, _, _ = ref
    out = {}
    for name, flavour, sep_ref, enc_special, B, L, P, seed in DT_FLAVOUR_CASES:
        a = ref_args(64, 128, flavour)
        a.encode_separator_as_reference = sep_ref
        a.encode_special_tokens = enc_special
        model = ref_network.BUSCA(a).eval()
        sd = synth.dt_state_dict(seed, d=64, ff=128, flavour=flavour)
        load_dt_weights(model, sd)
        inp = synth.dt_inputs(seed, B, L, P, sentinel_every=4)
        for mode, f64 in (("f64", True), ("f32", False)):
            set_fake_dtype(model, f64)
            r = run_ref_dt(model, inp)
            for k, v in r.items():
                out["%s/%s_%s" % (name, k, mode)] = v
        out[name + "/meta"] = np.array([B, L, P, seed, int(sep_ref), int(enc_special)])
        out[name + "/flavour"] = np.array(flavour)
        print("flavour case", name, flavour, r["logits"].shape)
        del model
    notes = []
    # CLS-*: PositionalEncoding._get_temporal_ids (encodings.py:161) replaces the index TENSOR by the int 0, torch.clamp then fails
    try:
        a = ref_args(64, 128, "CLS-MEM-SEP-CAN-BAD")
        model = ref_network.BUSCA(a).eval()
        run_ref_dt(model, synth.dt_inputs(1, 2, 4, 3))
        notes.append("CLS-MEM-SEP-CAN-BAD: ran")
    except Exception as e:  # noqa: BLE001
        notes.append("CLS-MEM-SEP-CAN-BAD: %s: %s" % (type(e).__name__, str(e)[:160]))
    # encode_special_tokens with dim_embedding != trans_dim: the tokens get dim_embedding entries and torch.cat fails
    try:
        a = ref_args(64, 128, "MEM-SEP-CAN-BAD")
        a.encode_special_tokens = True
        model = ref_network.BUSCA(a).eval()
        run_ref_dt(model, synth.dt_inputs(1, 2, 4, 3))
        notes.append("encode_special_tokens, E=512 d=64: ran")
    except Exception as e:  # noqa: BLE001
        notes.append("encode_special_tokens, E=512 d=64: %s: %s" % (type(e).__name__, str(e)[:160]))
    out["notes"] = np.array(notes)
    for n in notes:
        print("note:", n)
    np.savez_compressed(os.path.join(OUT, "flavours_dt.npz"), **out)


DT_GEOMETRY_CASES = [
    # name, d, ff, nhead, B, L, P, seed : head counts / feed-forward widths other than the shipped nhead = 4, ff = 2 d
    ("d64_h2_ff256", 64, 256, 2, 3, 11, 5, 41),
    ("d64_h4_ff64", 64, 64, 4, 3, 6, 16, 42),
    ("d256_h8_ff256", 256, 256, 8, 2, 11, 5, 43),
    ("d256_h2_ff1024", 256, 1024, 2, 2, 5, 7, 44),
    ("d256_h16_ff512", 256, 512, 16, 2, 11, 5, 45),
]


def make_dt_geometry(ref):
    """network.py:84-86 builds the encoder from args.nhead / args.ff_size: cases the one-kernel path is not built for."""
    ref_network, _, _ = ref
    out = {}
    for name, d, ff, nhead, B, L, P, seed in DT_GEOMETRY_CASES:
        a = ref_args(d, ff)
        a.nhead = nhead
        model = ref_network.BUSCA(a).eval()
        sd = synth.dt_state_dict(seed, d=d, ff=ff)
        load_dt_weights(model, sd)
        inp = synth.dt_inputs(seed, B, L, P, sentinel_every=4)
        set_fake_dtype(model, True)
        r = run_ref_dt(model, inp)
        for k, v in r.items():
            out["%s/%s" % (name, k)] = v
        out[name + "/meta"] = np.array([d, ff, nhead, B, L, P, seed])
        print("geometry case", name, r["logits"].shape, r["att"].shape)
        del model
    np.savez_compressed(os.path.join(OUT, "geometry_dt.npz"), **out)


class FakeTrack:
    """Track protocol of associate_embeddings."""
    def __init__(self, tlwh_hist, images, scale=1.0):
        self.tlwh_mem = [np.asarray(b, dtype=np.float64) for b in tlwh_hist]
        self.images_mem = list(images)
        self.scale = scale
        self.tlwh = self.tlwh_mem[-1]

    @property
    def tlbr(self):
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r


def smooth_crops(seed, n):
    """Smooth-ish random u8 crops [n,384,128,3] (pure noise would make every crop statistically identical)."""
    base = synth.randint_u8(seed, "crops", (n, 24, 8, 3)).astype(np.float32)
    up = np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)
    noise = synth.randint_u8(seed, "noise", (n, 384, 128, 3)).astype(np.float32) - 128
    return np.clip(up + 0.25 * noise, 0, 255).astype(np.uint8)


def assoc_scene(seed, hist_lens, n_det, with_kalman):
    """Deterministic fake tracks / detections (u8 crops from the portable PRNG)."""
    def rng_boxes(name, n):
        return np.stack([synth.uniform(seed, name + "x", (n,), 50, 1500), synth.uniform(seed, name + "y", (n,), 50, 800),
                         synth.uniform(seed, name + "w", (n,), 30, 120), synth.uniform(seed, name + "h", (n,), 80, 300)], 1).astype(np.float64)
    tracks = []
    for t, hl in enumerate(hist_lens):
        base = rng_boxes("trk%d" % t, 1)[0]
        hist = [base + np.array([2.0 * i, 1.0 * i, 0.3 * i, 0.5 * i]) for i in range(hl)]
        tracks.append(FakeTrack(hist, list(smooth_crops(seed * 100 + t, hl)), scale=1.0 + 0.25 * (t % 2)))
    det_boxes = rng_boxes("det", max(n_det, 1))[:n_det]
    for i in range(min(n_det, len(tracks))):          # put some detections near tracks
        det_boxes[i] = tracks[i].tlwh_mem[-1] + np.array([5.0, -3.0, 2.0, 4.0])
    det_imgs = smooth_crops(seed * 100 + 50, max(n_det, 1))
    dets = [FakeTrack([det_boxes[i]], [det_imgs[i]], scale=1.0) for i in range(n_det)]
    kal = []
    if with_kalman:
        kimgs = smooth_crops(seed * 100 + 60, len(tracks))
        kal = [FakeTrack([tr.tlwh_mem[-1] + np.array([1.0, 1.0, 0.0, 0.0])], [kimgs[i]], scale=tr.scale) for i, tr in enumerate(tracks)]
    return tracks, dets, kal


def make_enc(ref):
    """Bucket indices for seeded + adversarial boxes in both dtype modes, and rows of the real table vs LUT."""
    _, _, ref_enc = ref
    out = {}
    for d in (12, 64):
        pe = ref_enc.PositionalEncoding(d, input_flavour="MEM-SEP-CAN-BAD", dropout=0.1, encode_sep_as_ref=True,
                                        batch_first=True, device="cpu")
        idx = np.array([[0, 0, 0], [210, 210, 60], [87, 100, 10], [2, 105, 32], [179, 210, 34], [100, 210, 34], [105, 1, 59]])
        out["pe_idx_d%d" % d] = idx
        out["pe_rows_d%d" % d] = np.stack([pe.pe[i, j, k].numpy().view(np.uint16) for i, j, k in idx])
        if d == 64:
            inp = synth.dt_inputs(31, 24, 11, 16, sentinel_every=6)
            mb, cb = inp["mem_boxes"].copy(), inp["can_boxes"].copy()
            # adversarial: candidates that sit exactly on / next to the reference box, degenerate boxes
            cb[1, 0] = mb[1, -1]
            cb[1, 1] = mb[1, -1] + np.array([1e-3, 0, 1e-3, 0], np.float32)
            cb[2, 0] = np.array([100, 100, 100, 100], np.float32)      # zero-area box
            cb[2, 1] = mb[2, -1] * 4.0
            mb[3, 0] = mb[3, -1]
            out["ids_mem_boxes"], out["ids_can_boxes"] = mb, cb
            mem = torch.zeros(24, 11, d)
            can = torch.zeros(24, 2 * 18, d)
            for mode, f64 in (("f64", True), ("f32", False)):
                m = types.SimpleNamespace(pos_encoder=pe)
                set_fake_dtype(m, f64)
                mbt, cbt = torch.from_numpy(mb), torch.from_numpy(cb)
                fakes = pe._insert_fake_bboxes(can=can, can_bboxes=cbt, ref_bbox=mbt[:, -1:, :].clone(), num_candidates=18, encode_sep_as_ref=True)
                mt, ct = pe._get_temporal_ids(mem=mem, can=can, num_candidates=18)
                (mxy, msz), (cxy, csz) = pe._get_spatial_ids(mem_bboxes=mbt, can_bboxes=fakes)
                ids = torch.stack([torch.cat([mxy, cxy], 1), torch.cat([msz, csz], 1), torch.cat([mt, ct], 1)], -1)
                out["ids_" + mode] = ids.numpy().astype(np.int32)
        del pe
    np.savez_compressed(os.path.join(OUT, "enc.npz"), **out)
    print("wrote enc", {k: v.shape for k, v in out.items()})


def make_enc_big(ref):
    """Bucket indices of 512 tracks x 47 tokens (24k tokens, both dtype modes) computed by the reference: the volume test
    for the kernel's index arithmetic (an index is right or wrong, there is no tolerance)."""
    _, _, ref_enc = ref
    d = 12
    pe = ref_enc.PositionalEncoding(d, input_flavour="MEM-SEP-CAN-BAD", dropout=0.1, encode_sep_as_ref=True,
                                    batch_first=True, device="cpu")
    B, L, P = 512, 11, 16
    inp = synth.dt_inputs(77, B, L, P, sentinel_every=16)
    mb, cb = inp["mem_boxes"], inp["can_boxes"]
    out = {"B": B, "L": L, "P": P, "seed": 77}
    mem, can = torch.zeros(B, L, d), torch.zeros(B, 2 * (P + 2), d)
    for mode, f64 in (("f64", True), ("f32", False)):
        m = types.SimpleNamespace(pos_encoder=pe)
        set_fake_dtype(m, f64)
        mbt, cbt = torch.from_numpy(mb), torch.from_numpy(cb)
        fakes = pe._insert_fake_bboxes(can=can, can_bboxes=cbt, ref_bbox=mbt[:, -1:, :].clone(), num_candidates=P + 2, encode_sep_as_ref=True)
        mt, ct = pe._get_temporal_ids(mem=mem, can=can, num_candidates=P + 2)
        (mxy, msz), (cxy, csz) = pe._get_spatial_ids(mem_bboxes=mbt, can_bboxes=fakes)
        ids = torch.stack([torch.cat([mxy, cxy], 1), torch.cat([msz, csz], 1), torch.cat([mt, ct], 1)], -1)
        out["ids_" + mode] = ids.numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "enc_big.npz"), **out)
    print("wrote enc_big", out["ids_f64"].shape)


def make_geom(ref):
    ref_network, ref_tracking, _ = ref
    out = {}
    a = np.stack([synth.uniform(5, "ax", (40,), 0, 1900), synth.uniform(5, "ay", (40,), 0, 1000)], 1).astype(np.float64)
    a = np.concatenate([a, a + np.stack([synth.uniform(5, "aw", (40,), 10, 200), synth.uniform(5, "ah", (40,), 20, 400)], 1)], 1)
    b = np.stack([synth.uniform(6, "bx", (77,), 0, 1900), synth.uniform(6, "by", (77,), 0, 1000)], 1).astype(np.float64)
    b = np.concatenate([b, b + np.stack([synth.uniform(6, "bw", (77,), 10, 200), synth.uniform(6, "bh", (77,), 20, 400)], 1)], 1)
    out["a"], out["b"] = a, b
    out["center"] = ref_tracking.center_distance(a, b)
    out["center_w"] = ref_tracking.center_distance(a, b, weight_size=True)
    out["missing_ltrb"] = ref_tracking.missing_candidate_bbox(flavour="ltrb")
    out["missing_ltwh"] = ref_tracking.missing_candidate_bbox(flavour="ltwh")
    # memory sampling (network.py:247-279)
    gm = ref_network.BUSCA._get_track_mem
    rows = []
    for n_hist in (1, 3, 10, 11, 12, 15, 40, 101):
        for seq_len in (1, 5, 11):
            for broader in (True, False):
                trk = types.SimpleNamespace(images_mem=list(range(n_hist)), tlwh_mem=[np.zeros(4)] * n_hist, scale=1.0)
                mem, _ = gm(None, trk, seq_len, broader)
                rows.append([n_hist, seq_len, int(broader)] + list(mem) + [-1] * (11 - len(mem)))
    out["track_mem"] = np.array(rows, dtype=np.int64)
    # cutout geometry + mean fill without the (third-party) resize
    fr = synth.randint_u8(3, "frame", (540, 960, 3))
    boxes = np.array([[100.3, 50.2, 180.9, 300.7], [-20.5, -30.0, 60.2, 200.0], [900.0, 400.0, 1000.0, 600.0],
                      [10.0, 10.0, 138.0, 394.0], [5.5, 5.5, 6.2, 6.1], [400.0, 100.0, 1000.0, 539.5]], dtype=np.float32)
    out["cut_boxes"] = boxes
    for i, bx in enumerate(boxes):
        cut = ref_tracking._cutout_with_pad(fr, bx)
        out["cut_shape_%d" % i] = np.array(cut.shape)
        out["cut_sum_%d" % i] = np.array([int(cut.astype(np.int64).sum()), int(cut[0, 0, 0]), int(cut[-1, -1, 2])])
    np.savez_compressed(os.path.join(OUT, "geom.npz"), **out)
    print("wrote geom")


def load_reid_weights(enc, seed):
    full = enc.model.state_dict()
    for k, v in synth.reid_state_dict(seed).items():
        assert tuple(full[k].shape) == tuple(v.shape), k
        full[k] = torch.from_numpy(v)
    enc.model.load_state_dict(full)


REID_BIG_CASES = ((96, 1096), (200, 1200))


def make_reid_big(ref):
    """Batches large enough for the default large-batch schedule of the HIP extractor (Gram statistics from ~21 crops,
    halo-resident 3x3 convs and fused tails from 96): the reference's own ReID_Encoder features."""
    make_reid(ref, cases=REID_BIG_CASES, fname="reid_big.npz")


REID_CFG4_N, REID_CFG4_SEED = 1408, 2408            # BASELINE configs[3]: 128 lost x 11 memory crops = one 1 408-crop BatchNorm batch
REID_CFG4_DUP = (1408, 55, 2409)                     # 1 408 candidate slots over 55 distinct crops: the MOT20 duplication ratio 4 096 / 160


def cfg4_dup_indices():
    """Which distinct crop fills each slot of the duplicated batch (every distinct crop at least once, then seeded draws)."""
    slots, distinct, seed = REID_CFG4_DUP
    extra = synth.randint_u8(seed, "dup", (slots - distinct, 2)).astype(np.int64)
    return np.concatenate([np.arange(distinct), (extra[:, 0] * 256 + extra[:, 1]) % distinct])


def make_reid_cfg4(ref):
    """cfg4-sized BatchNorm batches through the reference's own ReID_Encoder (about 11 TFLOP each on the CPU; needs ~30 GB of RAM):
    (a) 1 408 distinct crops; (b) 1 408 slots filled from 55 distinct crops, expanded on the CPU exactly as the reference's
    associate_embeddings would build the batch - the features of the 55 distinct crops (first occurrence) are kept."""
    ref_network = ref[0]
    enc = ref_network.ReID_Encoder(num_classes=299, device=torch.device("cpu"), pretrained_path="no",
                                   use_domain_adaptation=True, trainable=False, use_checkpointing=False)
    load_reid_weights(enc, 3)

    def run(crops):
        x = crops.astype(np.float32) / 255.0
        x -= np.array([0.406, 0.456, 0.485])
        x /= np.array([0.225, 0.224, 0.299])
        xt = torch.from_numpy(x).float()[..., [2, 1, 0]].permute(0, 3, 1, 2)
        del x
        with torch.no_grad():
            _, feats = enc(xt)
        return feats.numpy()

    out = {}
    out["feats_n%d_seed%d" % (REID_CFG4_N, REID_CFG4_SEED)] = run(smooth_crops(REID_CFG4_SEED, REID_CFG4_N))
    slots, distinct, seed = REID_CFG4_DUP
    idx = cfg4_dup_indices()
    feats = run(smooth_crops(seed, distinct)[idx])
    first = np.array([int(np.argmax(idx == k)) for k in range(distinct)])
    assert np.abs(feats - feats[first][idx]).max() == 0.0         # copies of a crop get identical features
    out["dupfeats_slots%d_distinct%d_seed%d" % (slots, distinct, seed)] = feats[first]
    np.savez_compressed(os.path.join(OUT, "reid_cfg4.npz"), **out)
    print("wrote reid_cfg4.npz", {k: v.shape for k, v in out.items()})


def make_reid_cfg4_f64(ref):
    """The 1 408-crop batch of reid_cfg4 once more in FLOAT64 (oracle/reid.py with dtype=float64, ~45 GB of RAM): the reference's
    float32 CPU kernels and the HIP extractor's exact-f32 flavour differ by 1.7e-3 on this batch (5e-5 at 200 crops) - this
    fixture says which of the two is nearer to the exact result."""
    from oracle import reid as oreid
    crops = smooth_crops(REID_CFG4_SEED, REID_CFG4_N)
    x = oreid.crops_to_reid_input(crops)
    feats = oreid.reid_forward(synth.reid_state_dict(3), x, dtype=torch.float64).numpy()
    np.savez_compressed(os.path.join(OUT, "reid_cfg4_f64.npz"), **{"feats64_n%d_seed%d" % (REID_CFG4_N, REID_CFG4_SEED): feats.astype(np.float32)})
    print("wrote reid_cfg4_f64.npz", feats.shape)


def make_reid(ref, cases=((3, 43), (5, 45)), fname="reid.npz"):
    ref_network = ref[0]
    enc = ref_network.ReID_Encoder(num_classes=299, device=torch.device("cpu"), pretrained_path="no",
                                   use_domain_adaptation=True, trainable=False, use_checkpointing=False)
    load_reid_weights(enc, 3)
    out = {}
    for n, seed in cases:
        crops = smooth_crops(seed, n)
        x = crops.astype(np.float32) / 255.0
        x -= np.array([0.406, 0.456, 0.485])
        x /= np.array([0.225, 0.224, 0.299])
        xt = torch.from_numpy(x).float()[..., [2, 1, 0]].permute(0, 3, 1, 2)
        _, feats = enc(xt)
        out["feats_n%d_seed%d" % (n, seed)] = feats.numpy()
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print("wrote", fname, {k: v.shape for k, v in out.items()})


ASSOC_CASES = [("a", [15, 3, 11], 3, True, 5), ("b", [12, 30], 8, True, 5), ("c", [11, 11, 20], 4, False, 5), ("d", [13], 0, True, 5)]


def make_assoc(ref):
    """associate_embeddings end to end (reference ReID + DT on CPU, float32) on fake tracker objects."""
    ref_network, ref_tracking, _ = ref
    d, ff, seed = 64, 128, 17
    model = build_ref_model(ref_network, d, ff)
    load_dt_weights(model, synth.dt_state_dict(seed, d=d, ff=ff))
    model.reid_encoder = ref_network.ReID_Encoder(num_classes=299, device=torch.device("cpu"), pretrained_path="no",
                                                  use_domain_adaptation=True, trainable=False, use_checkpointing=False)
    load_reid_weights(model.reid_encoder, seed)
    out = {}
    for ci, (name, hist, n_det, kal, P) in enumerate(ASSOC_CASES):
        tracks, dets, kals = assoc_scene(seed + ci, hist, n_det, kal)
        if n_det:
            dists = ref_tracking.center_distance(np.array([t.tlbr * t.scale for t in tracks]), np.array([x.tlbr * x.scale for x in dets]))
        else:
            dists = np.zeros((len(tracks), 0))
        for mode, f64 in (("f64", True), ("f32", False)):
            set_fake_dtype(model, f64)
            for sel in (True, False):
                with torch.no_grad():
                    pm, rel = model.associate_embeddings(tracks_embeddings=tracks, dets_embeddings=dets, dists_matrix=dists, seq_len=11,
                                                         num_candidates=P, use_broader_memory=True, select_highest_candidate=sel,
                                                         extra_kalman_candidates=kals, normalize_ims=True)
                out["%s_probs_%s_sel%d" % (name, mode, int(sel))] = pm
                out["%s_reliable" % name] = rel
        out["%s_dists" % name] = dists
        print("assoc", name, pm.shape, rel, flush=True)
    np.savez_compressed(os.path.join(OUT, "assoc.npz"), **out)
    print("wrote assoc")


# (threshold, keep_highest_value) of the one-hot selection (network.py:415-422; StrongSORT tracker.py:332-333, GHOST :757-758)
ASSOC_SELECT_CASES = [(None, True), (0.0, True), (0.35, False), (0.35, True), (0.6, True), (-1.0, False)]


def make_assoc_select(ref):
    """highest_candidate_minimum_thresh / keep_highest_value of associate_embeddings on scenes a and b (d=64)."""
    ref_network, ref_tracking, _ = ref
    d, ff, seed = 64, 128, 17
    model = build_ref_model(ref_network, d, ff)
    load_dt_weights(model, synth.dt_state_dict(seed, d=d, ff=ff))
    model.reid_encoder = ref_network.ReID_Encoder(num_classes=299, device=torch.device("cpu"), pretrained_path="no",
                                                  use_domain_adaptation=True, trainable=False, use_checkpointing=False)
    load_reid_weights(model.reid_encoder, seed)
    set_fake_dtype(model, True)
    out = {}
    for ci in (0, 1):
        name, hist, n_det, kal, P = ASSOC_CASES[ci]
        tracks, dets, kals = assoc_scene(seed + ci, hist, n_det, kal)
        dists = ref_tracking.center_distance(np.array([t.tlbr * t.scale for t in tracks]), np.array([x.tlbr * x.scale for x in dets]))
        for si, (th, keep) in enumerate(ASSOC_SELECT_CASES):
            with torch.no_grad():
                pm, _ = model.associate_embeddings(tracks_embeddings=tracks, dets_embeddings=dets, dists_matrix=dists, seq_len=11,
                                                   num_candidates=P, use_broader_memory=True, select_highest_candidate=True,
                                                   highest_candidate_minimum_thresh=th, keep_highest_value=keep,
                                                   extra_kalman_candidates=kals, normalize_ims=True)
            out["%s_sel%d" % (name, si)] = pm
        out["%s_dists" % name] = dists
    np.savez_compressed(os.path.join(OUT, "assoc_select.npz"), **out)
    print("wrote assoc_select", {k: v.shape for k, v in out.items()})


def normalise(images):
    """network.py:470-478 applied to a list of u8 crops -> list of float32 crops (what a normalize_ims=False caller holds)."""
    out = []
    for im in images:
        x = np.asarray(im).astype(np.float32) / 255.0
        x -= np.array([0.406, 0.456, 0.485])
        x /= np.array([0.225, 0.224, 0.299])
        out.append(x)
    return out


def denormalise(images):
    return [np.clip(np.rint((np.asarray(x).astype(np.float64) * np.array([0.225, 0.224, 0.299]) + np.array([0.406, 0.456, 0.485])) * 255.0), 0, 255).astype(np.uint8)
            for x in images]


def assoc_nonorm_scene():
    """Scene for normalize_ims=False: float32 pre-normalised crops, one incomplete memory (float zero crops), fewer
    detections than candidates (float zero padding)."""
    tracks, dets, kals = assoc_scene(31, [14, 4, 11], 2, True)
    f32 = lambda trk: FakeTrack(trk.tlwh_mem, normalise(trk.images_mem), trk.scale)
    return [f32(t) for t in tracks], [f32(t) for t in dets], [f32(t) for t in kals]


def make_assoc_nonorm(ref):
    ref_network, ref_tracking, _ = ref
    d, ff, seed = 64, 128, 17
    model = build_ref_model(ref_network, d, ff)
    load_dt_weights(model, synth.dt_state_dict(seed, d=d, ff=ff))
    model.reid_encoder = ref_network.ReID_Encoder(num_classes=299, device=torch.device("cpu"), pretrained_path="no",
                                                  use_domain_adaptation=True, trainable=False, use_checkpointing=False)
    load_reid_weights(model.reid_encoder, seed)
    set_fake_dtype(model, True)
    tracks, dets, kals = assoc_nonorm_scene()
    dists = ref_tracking.center_distance(np.array([t.tlbr * t.scale for t in tracks]), np.array([x.tlbr * x.scale for x in dets]))
    with torch.no_grad():
        pm, rel = model.associate_embeddings(tracks_embeddings=tracks, dets_embeddings=dets, dists_matrix=dists, seq_len=11, num_candidates=5,
                                             use_broader_memory=True, select_highest_candidate=False, extra_kalman_candidates=kals,
                                             normalize_ims=False)
    np.savez_compressed(os.path.join(OUT, "assoc_nonorm.npz"), probs=pm, reliable=rel, dists=dists)
    print("wrote assoc_nonorm", pm.shape, rel)


ASSOC512_CASE = ("r", [15, 3, 11, 12, 40, 11], 9, True, 5)


def make_assoc512(ref):
    """associate_embeddings end to end on the SHIPPED model shape (cfgR: d=512, ff=1024, L=11, P=5, Kalman candidates,
    use_broader_memory) - reference ReID + DT on CPU."""
    ref_network, ref_tracking, _ = ref
    d, ff, seed = 512, 1024, 23
    model = build_ref_model(ref_network, d, ff)
    load_dt_weights(model, synth.dt_state_dict(seed, d=d, ff=ff))
    model.reid_encoder = ref_network.ReID_Encoder(num_classes=299, device=torch.device("cpu"), pretrained_path="no",
                                                  use_domain_adaptation=True, trainable=False, use_checkpointing=False)
    load_reid_weights(model.reid_encoder, seed)
    name, hist, n_det, kal, P = ASSOC512_CASE
    tracks, dets, kals = assoc_scene(seed, hist, n_det, kal)
    dists = ref_tracking.center_distance(np.array([t.tlbr * t.scale for t in tracks]), np.array([x.tlbr * x.scale for x in dets]))
    out = {"%s_dists" % name: dists}
    for mode, f64 in (("f64", True), ("f32", False)):
        set_fake_dtype(model, f64)
        for sel in (True, False):
            with torch.no_grad():
                pm, rel = model.associate_embeddings(tracks_embeddings=tracks, dets_embeddings=dets, dists_matrix=dists, seq_len=11,
                                                     num_candidates=P, use_broader_memory=True, select_highest_candidate=sel,
                                                     extra_kalman_candidates=kals, normalize_ims=True)
            out["%s_probs_%s_sel%d" % (name, mode, int(sel))] = pm
            out["%s_reliable" % name] = rel
    np.savez_compressed(os.path.join(OUT, "assoc512.npz"), **out)
    print("wrote assoc512", pm.shape, rel)


def make_track(ref):
    """Kalman prediction of STrack.multi_predict (byte_tracker.py:50-61) through the reference's vendored KalmanFilter
    (adapters/CenterTrack/src/lib/utils/mot_online/kalman_filter.py, the adapter's fallback import)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_kalman_filter", os.path.join(REF, "adapters/CenterTrack/src/lib/utils/mot_online/kalman_filter.py"))
    kfm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kfm)
    kf = kfm.KalmanFilter()
    n = 37
    cx, cy = synth.uniform(11, "cx", (n,), 50, 1800), synth.uniform(11, "cy", (n,), 50, 1000)
    hh = synth.uniform(11, "h", (n,), 30, 500)
    ar = synth.uniform(11, "a", (n,), 0.25, 0.6)
    means, covs = [], []
    for i in range(n):
        m, c = kf.initiate(np.array([cx[i], cy[i], ar[i], hh[i]], dtype=np.float64))
        for step in range(1 + i % 3):                    # a few predict/update cycles: dense covariance, non-zero velocities
            m, c = kf.predict(m, c)
            z = np.array([cx[i] + 3.0 * (step + 1), cy[i] - 2.0 * (step + 1), ar[i] * 1.01, hh[i] * 1.02])
            m, c = kf.update(m, c, z)
        means.append(m); covs.append(c)
    mean, cov = np.asarray(means), np.asarray(covs)
    not_tracked = (synth.uniform(11, "st", (n,), 0, 1) < 0.4)
    mm = mean.copy()                                     # byte_tracker.py:52-57
    mm[not_tracked, 7] = 0
    pm, pc = kf.multi_predict(mm, cov)
    np.savez_compressed(os.path.join(OUT, "track.npz"), mean=mean, cov=cov, not_tracked=not_tracked, pred_mean=pm, pred_cov=pc)
    print("wrote track", pm.shape, pc.shape)


def main():
    which = set(sys.argv[1:]) or {"dt", "enc", "geom", "assoc", "reid", "track"}
    ref = import_reference()
    if "dt" in which:
        make_dt(ref)
    if "dt_geometry" in which:
        make_dt_geometry(ref)
    if "dt_flavours" in which:
        make_dt_flavours(ref)
    if "dt512" in which:
        make_dt512(ref)
    if "assoc512" in which:
        make_assoc512(ref)
        _MODELS.pop((512, 1024), None)
    for name in ("enc", "enc_big", "geom", "assoc", "assoc_select", "assoc_nonorm", "reid", "reid_big", "reid_cfg4", "reid_cfg4_f64", "track"):
        fn = globals().get("make_" + name)
        if name in which and fn is not None:
            fn(ref)


if __name__ == "__main__":
    main()
