"""Oracle (CPU checker, test infrastructure only): host-side batching and scatter of
/root/reference/busca/network.py:247-429 (`_get_track_mem`, `associate_embeddings`), restated as plain
Python/numpy loops.  `step_fn(mem_u8, can_u8, mem_ltrb, can_ltrb) -> probs[B, P+2]` supplies the
network (oracle ReID+DT, or anything else under test).
"""
import numpy as np

from .geometry import missing_candidate_bbox

IMG_H, IMG_W = 384, 128   # ReID_Encoder.PRETRAINED_SIZE, network.py:512


def get_track_mem_indices(n_hist, seq_len, use_broader_memory):
    """network.py:247-275 as an index list into the track history (None -> incomplete memory)."""
    if use_broader_memory and not (seq_len == 1 and n_hist >= 1) and n_hist >= seq_len:
        sep = float(n_hist - 1) / float(seq_len - 1)
        return [int(i * sep) for i in range(seq_len)]
    return list(range(n_hist))[-seq_len:]


def build_batch(tracks, dets, dists, seq_len, num_candidates, use_broader_memory, kalman=(), pinned_numpy=True, normalize_ims=True):
    """network.py:293-398 up to the forward call.  Returns dict with u8 crops, float32 ltrb boxes,
    candidate global indices (None for padding), n_avail and the reliable flags.
    normalize_ims=False (network.py:285): the crops are float32, already normalised by the caller, and the zero crops of
    incomplete memories / padded candidates are float 0.0 in that normalised space; `mem_u8` / `can_u8` are then float32."""
    B = len(tracks)
    im_dtype = np.uint8 if normalize_ims else np.float32
    mem_u8 = np.zeros((B, seq_len, IMG_H, IMG_W, 3), im_dtype)
    mem_box = np.zeros((B, seq_len, 4), np.float64)
    reliable = np.zeros(B, bool)
    for t, trk in enumerate(tracks):
        idx = get_track_mem_indices(len(trk.images_mem), seq_len, use_broader_memory)
        if len(idx) == seq_len:
            for j, i in enumerate(idx):
                mem_u8[t, j] = trk.images_mem[i]
                mem_box[t, j] = np.asarray(trk.tlwh_mem[i]) * trk.scale
            reliable[t] = True
        else:                                                        # :303-308
            mem_box[t] = np.array([250.0, 250.0, 500.0, 500.0])
    P = num_candidates
    can_u8 = np.zeros((B, P, IMG_H, IMG_W, 3), im_dtype)
    can_box = np.zeros((B, P, 4), np.float64)
    inds = []
    n_avail = min(len(dets), P)
    miss = missing_candidate_bbox(flavour="ltwh", pinned_numpy=pinned_numpy)
    for t in range(B):
        order = np.argsort(dists[t], kind="stable")[:P].tolist() if len(dets) else []
        order += [None] * (P - len(order))
        for j, di in enumerate(order):
            if di is None:
                can_box[t, j] = miss
            else:
                can_u8[t, j] = dets[di].images_mem[-1]
                can_box[t, j] = np.asarray(dets[di].tlwh_mem[-1]) * dets[di].scale
        inds.append(order)
    if len(kalman) > 0:                                              # :363-380
        n_avail = min(len(dets) + 1, P)
        slot = min(len(dets), P - 1)
        for t in range(B):
            k = kalman[t]
            inds[t][slot] = len(dets) + t
            can_box[t, slot] = np.asarray(k.tlwh) * k.scale
            can_u8[t, slot] = k.images_mem[-1]
    mem_box = mem_box.astype(np.float32)                             # :319,389 .float()
    can_box = can_box.astype(np.float32)
    with np.errstate(over="ignore"):
        mem_box[..., 2:] += mem_box[..., :2]                         # ltwh_to_ltrb :483-489
        can_box[..., 2:] += can_box[..., :2]
    return dict(mem_u8=mem_u8, can_u8=can_u8, mem_ltrb=mem_box, can_ltrb=can_box, inds=inds,
                n_avail=n_avail, reliable=reliable)


def scatter_probs(probs, inds, n_avail, n_tracks, n_dets, n_kalman, select_highest_candidate,
                  highest_candidate_minimum_thresh=None, keep_highest_value=False):
    """network.py:407-425."""
    num_cols = n_dets if n_kalman == 0 else n_dets + n_kalman
    out = np.zeros((n_tracks, num_cols))
    for t in range(n_tracks):
        p = np.asarray(probs[t])
        if select_highest_candidate:
            new = np.zeros_like(p)
            th = highest_candidate_minimum_thresh
            if th is None or th == 0 or (th > 0.0 and np.max(p) >= th):
                new[np.argmax(p)] = np.max(p) if keep_highest_value else 1.0
            p = new
        cols = inds[t][:n_avail]
        out[t, cols] = p[:n_avail]
    return out


def associate_embeddings(step_fn, tracks, dets, dists, seq_len, num_candidates, use_broader_memory,
                         select_highest_candidate, highest_candidate_minimum_thresh=None,
                         keep_highest_value=False, extra_kalman_candidates=(), pinned_numpy=True, normalize_ims=True):
    """network.py:282-429."""
    if len(tracks) == 0:
        return None, None
    if len(dets) == 0 and len(extra_kalman_candidates) == 0:
        return None, None
    b = build_batch(tracks, dets, dists, seq_len, num_candidates, use_broader_memory,
                    extra_kalman_candidates, pinned_numpy, normalize_ims)
    probs = step_fn(b["mem_u8"], b["can_u8"], b["mem_ltrb"], b["can_ltrb"])
    out = scatter_probs(probs, b["inds"], b["n_avail"], len(tracks), len(dets), len(extra_kalman_candidates),
                        select_highest_candidate, highest_candidate_minimum_thresh, keep_highest_value)
    return out, b["reliable"]
