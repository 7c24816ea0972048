"""MOT-challenge text formats the reference's trackers read and write, so a run of an adapter on
busca_amd can be compared file-for-file with a run on the reference ("HOTA/IDF1 identical" needs identical files).

* `write_results_strongsort`  - adapters/StrongSORT/deep_sort_app.py:216-219 (`%d,%d,%.2f,%.2f,%.2f,%.2f,1,-1,-1,-1`)
* `write_results_bytetrack`   - adapters/ByteTrack/yolox/evaluators/mot_evaluator.py:30-40 (rounded to 1 decimal, score to 2)
* `read_results`, `compare_results` - parse either flavour; exact comparison of two result files
* `read_detections`           - MOT `det/det.txt` (frame, -1, x, y, w, h, conf, ...) grouped per frame
Host-side text I/O only; nothing here touches the GPU.
"""
import numpy as np


def write_results_strongsort(filename, results):
    """results: iterable of [frame_idx, track_id, x, y, w, h]."""
    with open(filename, "w") as f:
        for row in results:
            print("%d,%d,%.2f,%.2f,%.2f,%.2f,1,-1,-1,-1" % (row[0], row[1], row[2], row[3], row[4], row[5]), file=f)


def write_results_bytetrack(filename, results):
    """results: iterable of (frame_id, tlwhs, track_ids, scores); negative ids are skipped."""
    save_format = "{frame},{id},{x1},{y1},{w},{h},{s},-1,-1,-1\n"
    with open(filename, "w") as f:
        for frame_id, tlwhs, track_ids, scores in results:
            for tlwh, track_id, score in zip(tlwhs, track_ids, scores):
                if track_id < 0:
                    continue
                x1, y1, w, h = tlwh
                f.write(save_format.format(frame=frame_id, id=track_id, x1=round(x1, 1), y1=round(y1, 1), w=round(w, 1),
                                           h=round(h, 1), s=round(score, 2)))


def read_results(filename):
    """-> float64 array [rows, 7]: frame, id, x, y, w, h, score (the remaining columns are constants)."""
    rows = []
    with open(filename) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([float(v) for v in line.split(",")[:7]])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 7)


def compare_results(file_a, file_b):
    """Exact comparison of two result files (order-insensitive within a frame).  Returns a dict with `identical`, the
    number of rows of each file, and the first differing (frame, id) if any."""
    a, b = read_results(file_a), read_results(file_b)
    ka = a[np.lexsort((a[:, 1], a[:, 0]))] if len(a) else a
    kb = b[np.lexsort((b[:, 1], b[:, 0]))] if len(b) else b
    out = {"rows_a": int(len(a)), "rows_b": int(len(b)), "identical": False, "first_difference": None}
    if ka.shape == kb.shape and np.array_equal(ka, kb):
        out["identical"] = True
        return out
    for i in range(min(len(ka), len(kb))):
        if not np.array_equal(ka[i], kb[i]):
            out["first_difference"] = (int(ka[i, 0]), int(ka[i, 1]), ka[i].tolist(), kb[i].tolist())
            break
    return out


def read_detections(filename, min_confidence=None):
    """MOT det.txt -> {frame: float64 [n, 5] (x, y, w, h, conf)}."""
    det = np.loadtxt(filename, delimiter=",", ndmin=2)
    out = {}
    for fr in np.unique(det[:, 0]).astype(int):
        d = det[det[:, 0] == fr][:, 2:7]
        if min_confidence is not None:
            d = d[d[:, 4] >= min_confidence]
        out[int(fr)] = d
    return out
