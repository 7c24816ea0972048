#!/usr/bin/env python3
"""How often do the DEFAULT (fast) flavours decide differently from the EXACT ones?

    python tools/decision_agreement.py [steps=2000] [out.json]

Default flavours: fp16 ReID + f16-operand Decision Transformer (`reid_precision` / `precision` defaults of busca_amd.network.BUSCA).
Exact flavours:   float32 ReID convs + float32 Decision Transformer (the reference's arithmetic; <= 1e-3 against the reference's own
                  associate_embeddings at the shipped shape, tests/test_associate_gpu.py).

Every step is one `associate_embeddings` call at the SHIPPED model shape (d = 512, ff = 1024, L = 11, P = 5, broader memory, Kalman
candidates - config/*/*/*.yml) on a seeded scene: 1-8 lost tracks with 1-14 remembered crops each (shorter than L = incomplete
memories, longer = the strided memory selection of network.py:247-279), 2-12 detections (fewer than P = padded candidates; every
track takes its P nearest, so the candidate BatchNorm batch repeats detections), a Kalman candidate per track.  What the adapters
decide from the result (adapters/ByteTrack/yolox/tracker/byte_tracker.py:504-527, adapters/StrongSORT/deep_sort/tracker.py:332-371):
  * `probs[i, N + i] > busca_thresh` - the track's own Kalman prediction wins (thresholds in the shipped configs: 0.3 and 0.5);
  * the winning candidate (`select_highest_candidate`).
Reported: flips of both decisions between the flavours, the |delta prob| histogram, and - because random weights put few
probabilities near 0.3 / 0.5 - the flip rate at the WORST threshold (the one that splits the Kalman probabilities in half)."""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from busca_amd import synth  # noqa: E402


class Track:
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


def crop_pool(n, seed=5):
    """n smooth-ish u8 crops [384,128,3] from the portable PRNG (generated once; scenes draw from the pool)."""
    base = synth.randint_u8(seed, "pool", (n, 24, 8, 3)).astype(np.float32)
    up = np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)
    noise = synth.randint_u8(seed, "pooln", (n, 384, 128, 3)).astype(np.float32) - 128
    return np.clip(up + 0.25 * noise, 0, 255).astype(np.uint8)


def scene(rng, pool):
    nt, nd = int(rng.integers(1, 9)), int(rng.integers(2, 13))

    def boxes(n):
        return np.stack([rng.uniform(50, 1500, n), rng.uniform(50, 800, n), rng.uniform(30, 120, n), rng.uniform(80, 300, n)], 1)
    tracks = []
    for t in range(nt):
        hl = int(rng.integers(1, 15))
        base = boxes(1)[0]
        ident = int(rng.integers(0, len(pool) - 16))             # a track's crops are neighbours in the pool (same "person")
        hist = [base + np.array([2.0 * i, 1.0 * i, 0.3 * i, 0.5 * i]) for i in range(hl)]
        tracks.append(Track(hist, [pool[ident + (i % 8)] for i in range(hl)], scale=1.0 + 0.25 * (t % 2)))
    db = boxes(nd)
    for i in range(min(nd, nt)):                                   # some detections close to tracks
        if rng.random() < 0.7:
            db[i] = tracks[i].tlwh_mem[-1] + rng.normal(0, 6, 4)
    dets = [Track([db[i]], [pool[int(rng.integers(0, len(pool)))]], 1.0) for i in range(nd)]
    kal = [Track([tr.tlwh_mem[-1] + rng.normal(0, 2, 4)], [tr.images_mem[-1] if rng.random() < 0.5 else pool[int(rng.integers(0, len(pool)))]], tr.scale)
           for tr in tracks]
    tc = np.array([[t.tlwh[0] + t.tlwh[2] / 2, t.tlwh[1] + t.tlwh[3] / 2] for t in tracks])
    dc = np.array([[d.tlwh[0] + d.tlwh[2] / 2, d.tlwh[1] + d.tlwh[3] / 2] for d in dets])
    dists = np.sqrt(((tc[:, None, :] - dc[None, :, :]) ** 2).sum(-1))
    return tracks, dets, kal, dists


def build(precision, reid_precision, seed=23, decoder_gain=1.0):
    from busca_amd.network import BUSCA
    a = types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=512, ff_size=1024, activation="gelu", dropout_p=0.1,
                              input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                              encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), precision=precision,
                              reid_precision=reid_precision, pinned_numpy_semantics=True)
    m = BUSCA(a).to(torch.device("cuda:0")).eval()
    sd = dict(synth.dt_state_dict(seed, d=512, ff=1024))
    if decoder_gain != 1.0:          # sharper logits: a trained model decides near 0 / 1, random weights near 1 / (P + 2)
        sd["decoder.1.weight"] = sd["decoder.1.weight"] * np.float32(decoder_gain)
    sd.update({"reid_encoder.model." + k: v for k, v in synth.reid_state_dict(seed).items()})
    m.load_state_dict(sd)
    return m


def _compare(name, kal_e, kal_f, win_e, win_f, margin_e, dprob):
    nt = len(kal_e)
    out = {"flavour": name}
    for t in (0.3, 0.5):
        flip = (kal_e > t) != (kal_f > t)
        out["kalman_gt_%.1f" % t] = {"flips": int(flip.sum()), "flip_rate": float(flip.mean()),
                                      "tracks_within_0.05_of_threshold": int((np.abs(kal_e - t) < 0.05).sum()),
                                      "exact_positive_rate": float((kal_e > t).mean()),
                                      "largest_distance_to_threshold_among_flips": float(np.abs(kal_e[flip] - t).max()) if flip.any() else 0.0}
    tw = float(np.median(kal_e))
    flip = (kal_e > tw) != (kal_f > tw)
    out["kalman_gt_worst_threshold"] = {"threshold": tw, "flips": int(flip.sum()), "flip_rate": float(flip.mean())}
    wf = win_e != win_f
    out["winner"] = {"flips": int(wf.sum()), "flip_rate": float(wf.mean()), "largest_exact_margin_among_flips": float(margin_e[wf].max()) if wf.any() else 0.0,
                     "median_exact_margin": float(np.median(margin_e))}
    edges = [0, 1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 1.0]
    h, _ = np.histogram(dprob, bins=edges)
    out["abs_delta_prob"] = {"max": float(dprob.max()), "mean": float(dprob.mean()), "p99": float(np.quantile(dprob, 0.99)),
                             "histogram": {"<=%g" % edges[i + 1]: int(h[i]) for i in range(len(h))}}
    out["abs_delta_kalman_prob"] = {"max": float(np.abs(kal_e - kal_f).max()), "p99": float(np.quantile(np.abs(kal_e - kal_f), 0.99))}
    return out


SHARP_GAIN = 12.0       # decoder gain of the second run (probabilities straddle 0.5)
FLAVOURS = (("opt-in fast ReID (the default until round 3): float32 Decision Transformer + fp16 ReID", "f32", "f16"),
            ("fastest: f16-operand Decision Transformer + fp16 ReID", "f16", "f16"),
            ("default: float32 Decision Transformer + float32-equivalent ReID on split-fp16 MFMA (x3)", "f32", "x3"))


def run(steps=2000, seed=2026, verbose=False, decoder_gain=1.0):
    """decoder_gain > 1 scales the decoder's output layer: the probabilities then spread over (0, 1) and a share of the tracks lies
    next to the 0.5 threshold - with the plain random weights every Kalman probability sits between 0.13 and 0.30, so 'no flips at
    0.5' says nothing there."""
    pool = crop_pool(192)
    exact = build("f32", "f32", decoder_gain=decoder_gain)
    others = [build(p, r, decoder_gain=decoder_gain) for _, p, r in FLAVOURS]
    rng = np.random.default_rng(seed)
    kal_e, win_e, margin_e = [], [], []
    acc = [dict(kal=[], win=[], dprob=[]) for _ in FLAVOURS]
    incomplete = slots = 0
    for s in range(steps):
        tracks, dets, kal, dists = scene(rng, pool)
        n = len(dets)
        idx = np.arange(len(tracks))
        pe, _ = exact.associate_embeddings(tracks, dets, dists, 11, 5, True, False, extra_kalman_candidates=kal, normalize_ims=True)
        fe = exact._last["probs"].cpu().numpy()
        kal_e.append(pe[idx, n + idx]); win_e.append(fe.argmax(-1))
        srt = np.sort(fe, -1)
        margin_e.append(srt[:, -1] - srt[:, -2])
        for m, a in zip(others, acc):
            pf, _ = m.associate_embeddings(tracks, dets, dists, 11, 5, True, False, extra_kalman_candidates=kal, normalize_ims=True)
            ff = m._last["probs"].cpu().numpy()
            a["kal"].append(pf[idx, n + idx]); a["win"].append(ff.argmax(-1)); a["dprob"].append(np.abs(fe - ff).ravel())
        incomplete += sum(len(t.images_mem) < 11 for t in tracks)
        slots += len(tracks) * 5
        if verbose and (s + 1) % 500 == 0:
            print("step", s + 1, flush=True)
    kal_e, win_e, margin_e = np.concatenate(kal_e), np.concatenate(win_e), np.concatenate(margin_e)
    out = {"steps": steps, "decoder_gain": decoder_gain, "tracks": int(len(kal_e)), "tracks_with_incomplete_memory": int(incomplete), "candidate_slots": int(slots),
           "model": "d=512 ff=1024 L=11 P=5 (shipped shape), random weights seed 23, broader memory, Kalman candidates",
           "exact": "float32 ReID + float32 Decision Transformer (reference arithmetic)",
           "exact_kalman_prob_quantiles": {q: float(np.quantile(kal_e, float(q))) for q in ("0.05", "0.25", "0.5", "0.75", "0.95")},
           "comparisons": [_compare(name, kal_e, np.concatenate(a["kal"]), win_e, np.concatenate(a["win"]), margin_e, np.concatenate(a["dprob"]))
                           for (name, _, _), a in zip(FLAVOURS, acc)]}
    return out


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    res = run(steps, verbose=True)
    res["sharpened"] = run(max(200, steps // 3), verbose=True, decoder_gain=SHARP_GAIN)
    txt = json.dumps(res, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
