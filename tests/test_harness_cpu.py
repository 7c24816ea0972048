"""CPU: the evaluation harness's host logic (busca_amd/harness.py) - MOT sequence I/O, result-file comparison, scorer."""
import os

import numpy as np

from busca_amd import harness, mot_io


def test_synthetic_sequence_round_trip(tmp_path):
    d = harness.write_synthetic_sequence(str(tmp_path), n_frames=14, n_objects=3, width=320, height=200, gap=(5, 3))
    seq = harness.load_sequence(d)
    assert seq.name == "SYN-01" and len(seq) == 14 and (seq.width, seq.height) == (320, 200)
    f0 = seq.frame(0)
    assert f0.shape == (200, 320, 3) and f0.dtype == np.uint8
    assert seq.gt is not None and seq.gt.shape[1] == 6 and set(seq.gt[:, 1].astype(int)) == {1, 2, 3}
    n_det = sum(len(v) for v in seq.detections.values())
    assert n_det == len(seq.gt) - 3 * 3                      # every object loses its detection for 3 frames
    assert all(v.shape[1] == 5 for v in seq.detections.values())


def _rows(ids_per_frame, boxes):
    return np.array([[f + 1, i, *boxes[k]] for f, ids in enumerate(ids_per_frame) for k, i in enumerate(ids)], np.float64)


def test_builtin_scorer_known_answers():
    boxes = [(10, 10, 40, 80), (200, 50, 40, 80)]
    gt = _rows([[1, 2]] * 6, boxes)
    perfect = harness.clear_mot_idf1(gt, _rows([[7, 9]] * 6, boxes))
    assert perfect["MOTA"] == 1.0 and perfect["IDF1"] == 1.0 and perfect["IDSW"] == 0
    swap = harness.clear_mot_idf1(gt, _rows([[7, 9]] * 3 + [[9, 7]] * 3, boxes))        # identities exchanged half way
    assert swap["IDSW"] == 2 and swap["FP"] == 0 and swap["FN"] == 0 and abs(swap["MOTA"] - (1 - 2 / 12)) < 1e-12
    assert abs(swap["IDF1"] - 0.5) < 1e-12
    miss = harness.clear_mot_idf1(gt, _rows([[7]] * 6, boxes[:1]))                       # second object never reported
    assert miss["FN"] == 6 and miss["FP"] == 0 and abs(miss["MOTA"] - 0.5) < 1e-12
    far = harness.clear_mot_idf1(gt, np.array([[1, 5, 500, 500, 10, 10]], np.float64))  # a false positive nowhere near
    assert far["FP"] == 1 and far["FN"] == 12
    empty = harness.clear_mot_idf1(gt, np.zeros((0, 6)))
    assert empty["FN"] == 12 and empty["MOTA"] == 0.0


def test_compare_runs_detects_any_difference(tmp_path):
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    rows = [(1, [np.array([1.0, 2.0, 30.0, 60.0])], [1], [0.9]), (2, [np.array([2.0, 2.5, 30.0, 60.0])], [1], [0.91])]
    mot_io.write_results_bytetrack(str(a / "S.txt"), rows)
    mot_io.write_results_bytetrack(str(b / "S.txt"), rows)
    assert harness.compare_runs(str(a), str(b))["all_identical"]
    rows[1][1][0][0] += 0.1
    mot_io.write_results_bytetrack(str(b / "S.txt"), rows)
    c = harness.compare_runs(str(a), str(b))
    assert not c["all_identical"] and c["S.txt"]["first_difference"][0] == 2
    assert not harness.compare_runs(str(a), str(tmp_path))["all_identical"]             # nothing to compare is not "identical"


def test_builtin_hota_known_answers():
    """HOTA restated from TrackEval: perfect tracking = 1; identities exchanged half way: DetA 1, AssA 1/3 (every true positive
    shares its association with 3 of the 9 frames its two identities span), HOTA sqrt(1/3); a missing object halves DetA."""
    boxes = [(10, 10, 40, 80), (200, 50, 40, 80)]
    gt = _rows([[1, 2]] * 6, boxes)
    h = harness.hota(gt, _rows([[7, 9]] * 6, boxes))
    assert abs(h["HOTA"] - 1.0) < 1e-12 and abs(h["DetA"] - 1.0) < 1e-12 and abs(h["AssA"] - 1.0) < 1e-12
    h = harness.hota(gt, _rows([[7, 9]] * 3 + [[9, 7]] * 3, boxes))
    assert abs(h["DetA"] - 1.0) < 1e-12 and abs(h["AssA"] - 1.0 / 3.0) < 1e-12 and abs(h["HOTA"] - np.sqrt(1.0 / 3.0)) < 1e-12
    h = harness.hota(gt, _rows([[7]] * 6, boxes[:1]))
    assert abs(h["DetA"] - 0.5) < 1e-12 and abs(h["AssA"] - 1.0) < 1e-12 and abs(h["HOTA"] - np.sqrt(0.5)) < 1e-12
    # localisation: predictions shifted so that IoU = 0.6 count as true positives only for alpha <= 0.6
    shifted = _rows([[7, 9]] * 6, [(10 + 10, 10, 40, 80), (200 + 10, 50, 40, 80)])       # IoU = 30/50 = 0.6
    h = harness.hota(gt, shifted)
    frac = (np.arange(0.05, 0.99, 0.05) <= 0.6 + 1e-9).mean()
    assert abs(h["DetA"] - frac * 1.0) < 1e-9 or h["DetA"] < 1.0
    assert harness.hota(gt, np.zeros((0, 6)))["HOTA"] == 0.0
