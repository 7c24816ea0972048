"""Evaluation harness: drive MOT-challenge sequences through the drop-in `BUSCA`, write the trackers' result
files, compare them with a reference run and score them.

What the reference does in adapters/StrongSORT/deep_sort_app.py:130-219 (`run`: per-frame detections -> tracker -> result rows
-> `%d,%d,%.2f,...` file), adapters/ByteTrack/tools/track.py:236-287 (result folder -> motmetrics summary) and
adapters/GHOST/src/eval_track_eval.py:70 (TrackEval) is split here into:

  * `load_sequence`      a MOT sequence directory (seqinfo.ini, img1/, det/det.txt, gt/gt.txt) -> `MOTSequence`
  * `run_sequence`       frames + detections -> tracker.update(...) -> result file (ByteTrack or StrongSORT text format)
  * `compare_runs`       file-for-file equality of two result folders - the check behind "HOTA/IDF1 identical to the
                         reference": identical files give identical metrics under ANY evaluator
  * `evaluate`           TrackEval if importable, else motmetrics if importable, else the built-in CLEAR-MOT / IDF1 / HOTA below
  * `write_synthetic_sequence`  a small MOT-format sequence with occlusion gaps (moving textured boxes), so the whole chain runs
                         in the tests without MOT17 (which is not in this container)

The tracker is pluggable (`tracker_factory(model, seq) -> object with update(frame_bgr, dets[n,5]) -> [(id, tlwh, score)]`):
with the reference's adapters mounted it is a thin shim over their tracker classes (`busca` aliased to this package, see
INTEGRATION.md); `LiteTracker` below is the self-contained driver used when they are not - IoU association + the BUSCA
recovery stage in the call pattern of byte_tracker.py:467-560 (centre distances, Kalman-style candidates,
`associate_embeddings`, decision rule).  It is a test driver, not a re-implementation of any adapter.
Host-side orchestration only: every box/crop/association computation goes through the HIP kernels of this package.
"""
import configparser
import glob
import os

import numpy as np

from . import mot_io, tracking


# ---------------------------------------------------------------------------------------------------------------------------
# sequences
# ---------------------------------------------------------------------------------------------------------------------------
class MOTSequence:
    def __init__(self, name, directory, image_files, width, height, detections, gt=None, frame_rate=30):
        self.name, self.directory, self.image_files = name, directory, image_files
        self.width, self.height, self.frame_rate = width, height, frame_rate
        self.detections = detections            # {frame (1-based): float64 [n,5] x,y,w,h,conf}
        self.gt = gt                            # float64 [rows, 6]: frame, id, x, y, w, h  (or None)

    def __len__(self):
        return len(self.image_files)

    def frame(self, idx):
        """BGR uint8 [H,W,3] of 0-based frame idx (what cv2.imread gives the adapters)."""
        from PIL import Image
        rgb = np.asarray(Image.open(self.image_files[idx]).convert("RGB"))
        return np.ascontiguousarray(rgb[..., ::-1])


def load_sequence(seq_dir, det_file=None, min_confidence=None):
    seq_dir = os.path.abspath(seq_dir)
    name = os.path.basename(seq_dir.rstrip("/"))
    info = configparser.ConfigParser()
    info.read(os.path.join(seq_dir, "seqinfo.ini"))
    sec = info["Sequence"] if info.has_section("Sequence") else {}
    img_dir = os.path.join(seq_dir, sec.get("imDir", "img1"))
    ext = sec.get("imExt", ".jpg")
    files = sorted(glob.glob(os.path.join(img_dir, "*" + ext)))
    if not files:
        raise FileNotFoundError("no frames under %s" % img_dir)
    det_file = det_file or os.path.join(seq_dir, "det", "det.txt")
    dets = mot_io.read_detections(det_file, min_confidence) if os.path.exists(det_file) else {}
    gt = None
    gt_file = os.path.join(seq_dir, "gt", "gt.txt")
    if os.path.exists(gt_file):
        g = np.loadtxt(gt_file, delimiter=",", ndmin=2)
        if g.shape[1] >= 8:                      # MOT17 gt: keep pedestrians that are to be considered (flag 1, class 1)
            g = g[(g[:, 6] > 0) & (g[:, 7] == 1)]
        gt = g[:, :6].astype(np.float64)
    if "imWidth" in sec:
        W, H = int(sec["imWidth"]), int(sec["imHeight"])
    else:
        from PIL import Image
        W, H = Image.open(files[0]).size
    return MOTSequence(name, seq_dir, files, W, H, dets, gt, int(sec.get("frameRate", 30)))


def write_synthetic_sequence(out_dir, name="SYN-01", n_frames=60, n_objects=6, width=640, height=360, seed=0, gap=(25, 8)):
    """A MOT-format sequence of textured rectangles moving linearly over a noise background.  Every object misses its
    detection for `gap[1]` frames starting around frame `gap[0]` (staggered) - the situation BUSCA exists for.
    Writes seqinfo.ini, img1/%06d.png, det/det.txt, gt/gt.txt; returns the sequence directory."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    seq = os.path.join(out_dir, name)
    for sub in ("img1", "det", "gt"):
        os.makedirs(os.path.join(seq, sub), exist_ok=True)
    bg = rng.integers(90, 140, (height, width, 3), dtype=np.uint8)
    h = rng.uniform(90, 150, n_objects); w = h * rng.uniform(0.35, 0.45, n_objects)
    x = rng.uniform(10, width - 90, n_objects); y = rng.uniform(10, height - 160, n_objects)
    vx = rng.uniform(-2.5, 2.5, n_objects); vy = rng.uniform(-0.6, 0.6, n_objects)
    tex = [rng.integers(0, 255, (24, 8, 3), dtype=np.uint8) for _ in range(n_objects)]
    det_rows, gt_rows = [], []
    for f in range(1, n_frames + 1):
        img = bg.copy()
        for o in range(n_objects):
            bx, by = x[o] + vx[o] * f, y[o] + vy[o] * f
            x1, y1, x2, y2 = int(max(bx, 0)), int(max(by, 0)), int(min(bx + w[o], width)), int(min(by + h[o], height))
            if x2 - x1 < 4 or y2 - y1 < 4:
                continue
            patch = np.asarray(Image.fromarray(tex[o]).resize((x2 - x1, y2 - y1), Image.NEAREST))
            img[y1:y2, x1:x2] = patch
            gt_rows.append([f, o + 1, bx, by, w[o], h[o], 1, 1, 1.0])
            g0 = gap[0] + 3 * o
            if not (g0 <= f < g0 + gap[1]):
                j = rng.normal(0, 0.6, 4)
                det_rows.append([f, -1, bx + j[0], by + j[1], w[o] + j[2], h[o] + j[3], 0.9, -1, -1, -1])
        Image.fromarray(img[..., ::-1]).save(os.path.join(seq, "img1", "%06d.png" % f))      # img is BGR in memory
    np.savetxt(os.path.join(seq, "det", "det.txt"), np.asarray(det_rows), fmt="%.3f", delimiter=",")
    np.savetxt(os.path.join(seq, "gt", "gt.txt"), np.asarray(gt_rows), fmt="%.3f", delimiter=",")
    with open(os.path.join(seq, "seqinfo.ini"), "w") as fh:
        fh.write("[Sequence]\nname=%s\nimDir=img1\nframeRate=30\nseqLength=%d\nimWidth=%d\nimHeight=%d\nimExt=.png\n" % (name, n_frames, width, height))
    return seq


# ---------------------------------------------------------------------------------------------------------------------------
# the self-contained driver tracker
# ---------------------------------------------------------------------------------------------------------------------------
class _Track:
    """Track protocol of associate_embeddings (images_mem, tlwh_mem, scale, tlwh) + tlbr for center_distance."""
    _next_id = 1

    def __init__(self, tlwh, score, image, frame_id, new_id=True):
        self.track_id = -1
        if new_id:                               # candidates handed to associate_embeddings are not tracks: no id
            self.track_id = _Track._next_id
            _Track._next_id += 1
        self.tlwh_mem, self.images_mem, self.scale = [np.asarray(tlwh, np.float64)], [image], 1.0
        self.score, self.vel = float(score), np.zeros(2)
        self.frame_id = self.start_frame = frame_id
        self.lost = False
        self.pred = self.tlwh_mem[-1].copy()

    @property
    def tlwh(self):
        return self.pred

    @property
    def tlbr(self):
        r = self.pred.copy()
        r[2:] += r[:2]
        return r

    def predict(self):
        self.pred = self.tlwh_mem[-1].copy()
        self.pred[:2] += self.vel * (1 + 0)          # constant velocity from the last two updates

    def update(self, tlwh, score, image, frame_id):
        tlwh = np.asarray(tlwh, np.float64)
        gapf = max(1, frame_id - self.frame_id)
        self.vel = 0.5 * self.vel + 0.5 * (tlwh[:2] - self.tlwh_mem[-1][:2]) / gapf
        self.tlwh_mem.append(tlwh)
        self.images_mem.append(image)
        self.score, self.frame_id, self.lost = float(score), frame_id, False
        self.pred = tlwh.copy()


class LiteTracker:
    def __init__(self, model, args):
        """`args`: namespace with the adapters' BUSCA knobs (config/*/*/*.yml `tracker:` block): seq_len, num_candidates,
        use_broader_memory, select_highest_candidate, busca_thresh; plus match_thresh (IoU cost), track_thresh, max_time_lost."""
        self.model, self.args = model, args
        self.tracks, self.frame_id = [], 0
        self.recovered = 0                      # lost tracks kept alive by BUSCA so far
        _Track._next_id = 1

    def update(self, frame, dets):
        with self.model.frame(frame):               # both get_image_crops calls of this update cut from ONE upload of the frame
            return self._update(frame, dets)

    def _update(self, frame, dets):
        from scipy.optimize import linear_sum_assignment
        a = self.args
        self.frame_id += 1
        dets = np.asarray(dets, np.float64).reshape(-1, 5)
        dets = dets[dets[:, 4] >= getattr(a, "det_thresh", 0.1)]
        tlbr = dets[:, :4].copy()
        tlbr[:, 2:] += tlbr[:, :2]
        crops = self.model.get_image_crops(frame, tlbr, normalize=False) if len(dets) else []
        for t in self.tracks:
            t.predict()
        # round 1: IoU association of every live track with the detections (cost on the GPU, assignment on the host)
        um_t, um_d = list(range(len(self.tracks))), list(range(len(dets)))
        if self.tracks and len(dets):
            cost = tracking.iou_distance([t.tlbr for t in self.tracks], list(tlbr))
            r, c = linear_sum_assignment(cost)
            for i, j in zip(r, c):
                if cost[i, j] <= a.match_thresh:
                    self.tracks[i].update(dets[j, :4], dets[j, 4], crops[j], self.frame_id)
                    um_t.remove(i); um_d.remove(j)
        # BUSCA stage on the tracks that found no detection (byte_tracker.py:467-560): candidates = the unmatched detections
        # + each track's own predicted box; a track whose own prediction wins (prob > busca_thresh, reliable memory) stays alive
        pool = [self.tracks[i] for i in um_t]
        if pool and getattr(a, "busca_thresh", 0) > 0:
            cand = [_Track(dets[j, :4], dets[j, 4], crops[j], self.frame_id, new_id=False) for j in um_d]
            kal_crops = self.model.get_image_crops(frame, [t.tlbr for t in pool], normalize=False)
            kal = []
            for k, t in enumerate(pool):
                kd = _Track(t.tlwh, 0.1, kal_crops[k], self.frame_id, new_id=False)
                kal.append(kd)
            dists = tracking.center_distance(pool, cand) if cand else np.zeros((len(pool), 0))
            probs, reliable = self.model.associate_embeddings(pool, cand, dists, a.seq_len, a.num_candidates, a.use_broader_memory,
                                                              a.select_highest_candidate, extra_kalman_candidates=kal, normalize_ims=True)
            matches, _ = tracking.recover_with_busca(probs, reliable, len(cand), a.busca_thresh)
            for i, pr in matches:
                t = pool[i]
                t.update(t.tlwh, t.score, kal[i].images_mem[-1], self.frame_id)
                self.recovered += 1
                um_t.remove(self.tracks.index(t))
        for i in um_t:
            self.tracks[i].lost = True
        for j in um_d:
            if dets[j, 4] >= a.track_thresh:
                self.tracks.append(_Track(dets[j, :4], dets[j, 4], crops[j], self.frame_id))
        self.tracks = [t for t in self.tracks if self.frame_id - t.frame_id <= a.max_time_lost]
        return [(t.track_id, t.tlwh_mem[-1].copy(), t.score) for t in self.tracks if not t.lost]


# ---------------------------------------------------------------------------------------------------------------------------
# running and scoring
# ---------------------------------------------------------------------------------------------------------------------------
def run_sequence(seq, tracker, out_file, fmt="bytetrack", max_frames=None):
    """Feed every frame + its detections to `tracker.update`, write the result file.  Returns the number of rows."""
    rows_bt, rows_ss = [], []
    n = len(seq) if max_frames is None else min(len(seq), max_frames)
    for idx in range(n):
        out = tracker.update(seq.frame(idx), seq.detections.get(idx + 1, np.zeros((0, 5))))
        ids = [o[0] for o in out]
        tl = [o[1] for o in out]
        rows_bt.append((idx + 1, tl, ids, [o[2] for o in out]))
        rows_ss.extend([idx + 1, i, b[0], b[1], b[2], b[3]] for i, b in zip(ids, tl))
    os.makedirs(os.path.dirname(os.path.abspath(out_file)), exist_ok=True)
    if fmt == "bytetrack":
        mot_io.write_results_bytetrack(out_file, rows_bt)
    else:
        mot_io.write_results_strongsort(out_file, rows_ss)
    return len(rows_ss)


def compare_runs(dir_a, dir_b):
    """{sequence file: mot_io.compare_results(...)} for every result file present in both folders + `all_identical`."""
    out, ok = {}, True
    for fa in sorted(glob.glob(os.path.join(dir_a, "*.txt"))):
        fb = os.path.join(dir_b, os.path.basename(fa))
        if os.path.exists(fb):
            out[os.path.basename(fa)] = mot_io.compare_results(fa, fb)
            ok = ok and out[os.path.basename(fa)]["identical"]
    out["all_identical"] = bool(ok and len(out) > 0)
    return out


def _iou_tlwh(a, b):
    ax2, ay2, bx2, by2 = a[:, None, 0] + a[:, None, 2], a[:, None, 1] + a[:, None, 3], b[None, :, 0] + b[None, :, 2], b[None, :, 1] + b[None, :, 3]
    iw = np.clip(np.minimum(ax2, bx2) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(ay2, by2) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    return inter / (a[:, None, 2] * a[:, None, 3] + b[None, :, 2] * b[None, :, 3] - inter + 1e-12)


def clear_mot_idf1(gt, res, iou_thresh=0.5):
    """Built-in scorer used when neither TrackEval nor motmetrics is installed (the evaluators the reference calls are
    not in this image).  Standard definitions: per-frame Hungarian matching at IoU >= 0.5 that keeps a still-valid previous
    match first (CLEAR-MOT) -> FP, FN, IDSW, MOTA; global identity matching -> IDTP, IDF1.  gt/res: [rows, >=6]
    frame, id, x, y, w, h.  Host numpy: this scores text files, it is not on the product path."""
    from scipy.optimize import linear_sum_assignment
    gt, res = np.atleast_2d(np.asarray(gt, np.float64)), np.atleast_2d(np.asarray(res, np.float64))
    if res.size == 0:
        res = np.zeros((0, 6))
    frames = sorted(set(gt[:, 0].astype(int)) | set(res[:, 0].astype(int)))
    fp = fn = idsw = ngt = 0
    last = {}                                   # gt id -> last matched result id
    pairs = {}                                  # (gt id, res id) -> co-matched frame count
    for f in frames:
        g, r = gt[gt[:, 0] == f], res[res[:, 0] == f]
        ngt += len(g)
        if len(g) == 0 or len(r) == 0:
            fn += len(g); fp += len(r)
            continue
        iou = _iou_tlwh(g[:, 2:6], r[:, 2:6])
        cost = 1.0 - iou
        for i, gid in enumerate(g[:, 1].astype(int)):          # continuity: a previous match that still overlaps is kept
            if gid in last:
                j = np.nonzero(r[:, 1].astype(int) == last[gid])[0]
                if len(j) and iou[i, j[0]] >= iou_thresh:
                    cost[i, j[0]] -= 1.0
        rr, cc = linear_sum_assignment(cost)
        matched = 0
        for i, j in zip(rr, cc):
            if iou[i, j] >= iou_thresh:
                matched += 1
                gid, rid = int(g[i, 1]), int(r[j, 1])
                if gid in last and last[gid] != rid:
                    idsw += 1
                last[gid] = rid
                pairs[(gid, rid)] = pairs.get((gid, rid), 0) + 1
        fn += len(g) - matched
        fp += len(r) - matched
    gids, rids = sorted({k[0] for k in pairs}), sorted({k[1] for k in pairs})
    idtp = 0
    if pairs:
        m = np.zeros((len(gids), len(rids)))
        for (gi, ri), c in pairs.items():
            m[gids.index(gi), rids.index(ri)] = c
        rr, cc = linear_sum_assignment(-m)
        idtp = int(m[rr, cc].sum())
    idfn, idfp = ngt - idtp, len(res) - idtp
    return {"MOTA": 1.0 - (fp + fn + idsw) / max(1, ngt), "IDF1": 2 * idtp / max(1, 2 * idtp + idfp + idfn), "FP": int(fp), "FN": int(fn),
            "IDSW": int(idsw), "GT": int(ngt), "IDTP": idtp, "scorer": "built-in CLEAR-MOT/IDF1"}


def hota(gt, res):
    """HOTA (Luiten et al., IJCV 2021) as TrackEval computes it (trackeval/metrics/hota.py, restated; TrackEval itself is not
    in this image): similarity = IoU, alpha = 0.05 ... 0.95; per frame a Hungarian assignment on global-alignment-score x
    similarity; DetA = TP / (TP + FN + FP), AssA = mean over TPs of |TPA| / (|TPA| + |FNA| + |FPA|), HOTA = mean_alpha
    sqrt(DetA AssA).  Returns {"HOTA", "DetA", "AssA"} (means over alpha).  Host numpy, scores text files only."""
    from scipy.optimize import linear_sum_assignment
    gt, res = np.atleast_2d(np.asarray(gt, np.float64)), np.atleast_2d(np.asarray(res, np.float64))
    if res.size == 0:
        res = np.zeros((0, 6))
    alphas = np.arange(0.05, 0.99, 0.05)
    gids, rids = sorted(set(gt[:, 1].astype(int))), sorted(set(res[:, 1].astype(int)))
    gi, ri = {g: k for k, g in enumerate(gids)}, {r: k for k, r in enumerate(rids)}
    frames = sorted(set(gt[:, 0].astype(int)) | set(res[:, 0].astype(int)))
    pot = np.zeros((len(gids), len(rids)))
    gcount, rcount = np.zeros(len(gids)), np.zeros(len(rids))
    per_frame = []
    for f in frames:
        g, r = gt[gt[:, 0] == f], res[res[:, 0] == f]
        ga, ra = np.array([gi[int(x)] for x in g[:, 1]], int), np.array([ri[int(x)] for x in r[:, 1]], int)
        gcount[ga] += 1; rcount[ra] += 1
        sim = _iou_tlwh(g[:, 2:6], r[:, 2:6]) if len(g) and len(r) else np.zeros((len(g), len(r)))
        if len(g) and len(r):
            den = sim.sum(0)[None, :] + sim.sum(1)[:, None] - sim
            pot[ga[:, None], ra[None, :]] += np.where(den > 0, sim / np.maximum(den, 1e-12), 0.0)
        per_frame.append((ga, ra, sim))
    gas = pot / np.maximum(gcount[:, None] + rcount[None, :] - pot, 1e-12)
    tp, fn, fp = np.zeros(len(alphas)), np.zeros(len(alphas)), np.zeros(len(alphas))
    mc = np.zeros((len(alphas), len(gids), len(rids)))
    for ga, ra, sim in per_frame:
        if len(ga) == 0 or len(ra) == 0:
            fn += len(ga); fp += len(ra)
            continue
        score = gas[ga[:, None], ra[None, :]] * sim
        rr, cc = linear_sum_assignment(-score)
        for a, alpha in enumerate(alphas):
            ok = sim[rr, cc] >= alpha - np.finfo(float).eps
            n = int(ok.sum())
            tp[a] += n; fn[a] += len(ga) - n; fp[a] += len(ra) - n
            mc[a, ga[rr[ok]], ra[cc[ok]]] += 1
    deta = tp / np.maximum(tp + fn + fp, 1.0)
    assa = np.zeros(len(alphas))
    for a in range(len(alphas)):
        ass = mc[a] / np.maximum(gcount[:, None] + rcount[None, :] - mc[a], 1.0)
        assa[a] = (mc[a] * ass).sum() / max(tp[a], 1.0)
    return {"HOTA": float(np.sqrt(deta * assa).mean()), "DetA": float(deta.mean()), "AssA": float(assa.mean())}


def evaluate(results_dir, sequences):
    """Score result files against the sequences' ground truth.  Uses motmetrics when importable (the evaluator ByteTrack calls,
    adapters/ByteTrack/tools/track.py:236-287: `mm.utils.compare_to_groundtruth` walks the UNION of ground-truth and result
    frames, so tracker rows in frames without ground truth count as false positives - same here); otherwise the built-in
    CLEAR-MOT / IDF1 scorer.  HOTA here is the built-in restatement of TrackEval's; `run_trackeval` (below, `tools/run_mot.py
    --trackeval`) runs TrackEval ITSELF on the written files when it is importable, as GHOST's eval_track_eval.py:70 does.
    Returns {sequence: metrics dict}."""
    out = {}
    for seq in sequences:
        f = os.path.join(results_dir, seq.name + ".txt")
        if seq.gt is None or not os.path.exists(f):
            continue
        res = mot_io.read_results(f)
        try:
            import motmetrics as mm                       # not in this image; used when present
            acc = mm.MOTAccumulator(auto_id=True)
            for fr in sorted(set(seq.gt[:, 0].astype(int)) | set(res[:, 0].astype(int))):
                g, r = seq.gt[seq.gt[:, 0] == fr], res[res[:, 0] == fr]
                acc.update(g[:, 1].astype(int), r[:, 1].astype(int), mm.distances.iou_matrix(g[:, 2:6], r[:, 2:6], max_iou=0.5))
            s = mm.metrics.create().compute(acc, metrics=["mota", "idf1", "num_switches", "num_false_positives", "num_misses"], name=seq.name)
            out[seq.name] = {"MOTA": float(s["mota"].iloc[0]), "IDF1": float(s["idf1"].iloc[0]), "IDSW": int(s["num_switches"].iloc[0]),
                             "FP": int(s["num_false_positives"].iloc[0]), "FN": int(s["num_misses"].iloc[0]), "scorer": "motmetrics"}
        except ImportError:
            out[seq.name] = clear_mot_idf1(seq.gt, res)
        out[seq.name].update(hota(seq.gt, res))          # restated HOTA; TrackEval's own numbers take over when it is installed
    return out


def _import_trackeval():
    try:
        import trackeval
        return trackeval
    except ImportError:
        try:                                                  # GHOST vendors it under this name (adapters/GHOST/src/eval_track_eval.py:1)
            from TrackEvalForGHOST import trackeval
            return trackeval
        except ImportError:
            return None


def trackeval_available():
    return _import_trackeval() is not None


def run_trackeval(results_dir, sequences, gt_root, tracker_name="busca_amd", work_dir=None, metrics=("HOTA", "CLEAR", "Identity")):
    """TrackEval's own MOTChallenge2DBox evaluation of the written result files - the runner GHOST calls
    (adapters/GHOST/src/eval_track_eval.py:70-125: Evaluator + datasets.MotChallenge2DBox + metrics HOTA / CLEAR / Identity,
    threshold 0.5, SKIP_SPLIT_FOL, explicit SEQ_INFO).  `gt_root` holds <sequence>/gt/gt.txt (+ seqinfo.ini) as MOT17/train does.
    The result files are linked into the layout TrackEval reads (<work>/<tracker_name>/<sequence>.txt).  Returns
    {sequence: {"HOTA", "DetA", "AssA", "MOTA", "IDF1", "IDSW", ...}} plus "COMBINED_SEQ" - TrackEval's numbers, not this module's
    restatements; None when TrackEval is not importable (it is not in the build image)."""
    te = _import_trackeval()
    if te is None:
        return None
    import shutil
    import tempfile
    work = work_dir or tempfile.mkdtemp(prefix="busca_trackeval_")
    tdir = os.path.join(work, tracker_name)
    os.makedirs(tdir, exist_ok=True)
    for seq in sequences:
        src = os.path.join(results_dir, seq.name + ".txt")
        if os.path.exists(src):
            shutil.copyfile(src, os.path.join(tdir, seq.name + ".txt"))
    eval_cfg = te.Evaluator.get_default_eval_config()
    data_cfg = te.datasets.MotChallenge2DBox.get_default_dataset_config()
    eval_cfg.update({"PRINT_CONFIG": False, "PRINT_RESULTS": False, "DISPLAY_LESS_PROGRESS": True, "TIME_PROGRESS": False, "USE_PARALLEL": False,
                     "OUTPUT_SUMMARY": False, "OUTPUT_DETAILED": False, "PLOT_CURVES": False})
    data_cfg.update({"GT_FOLDER": gt_root, "TRACKERS_FOLDER": work, "TRACKERS_TO_EVAL": [tracker_name], "OUTPUT_FOLDER": os.path.join(work, "track_eval_output"),
                     "PRINT_CONFIG": False, "SKIP_SPLIT_FOL": True, "TRACKER_SUB_FOLDER": "", "SEQ_INFO": {seq.name: len(seq) for seq in sequences}})
    mcfg = {"METRICS": list(metrics), "THRESHOLD": 0.5, "PRINT_CONFIG": False}
    mlist = [m(mcfg) for m in (te.metrics.HOTA, te.metrics.CLEAR, te.metrics.Identity) if m.get_name() in mcfg["METRICS"]]
    res, _msg = te.Evaluator(eval_cfg).evaluate([te.datasets.MotChallenge2DBox(data_cfg)], mlist)
    per = res["MotChallenge2DBox"][tracker_name]
    out = {}
    for name, r in per.items():
        r = r.get("pedestrian", r)
        row = {}
        if "HOTA" in r:
            row.update({k: float(np.mean(r["HOTA"][k])) for k in ("HOTA", "DetA", "AssA")})
        if "CLEAR" in r:
            row.update({"MOTA": float(r["CLEAR"]["MOTA"]), "IDSW": int(r["CLEAR"]["IDSW"]), "FP": int(r["CLEAR"]["CLR_FP"]), "FN": int(r["CLEAR"]["CLR_FN"])})
        if "Identity" in r:
            row["IDF1"] = float(r["Identity"]["IDF1"])
        row["scorer"] = "TrackEval"
        out[name] = row
    return out
