"""Result-file formats of the reference's trackers (adapters/StrongSORT/deep_sort_app.py:216-219,
adapters/ByteTrack/yolox/evaluators/mot_evaluator.py:30-40)."""
import numpy as np

from busca_amd import mot_io


def test_strongsort_format_and_roundtrip(tmp_path):
    rows = [[1, 3, 10.0, 20.5, 30.123, 40.987], [1, 4, 0.004, -3.0, 5.0, 6.0], [2, 3, 11.0, 21.5, 30.0, 41.0]]
    p = tmp_path / "a.txt"
    mot_io.write_results_strongsort(p, rows)
    assert p.read_text().splitlines()[0] == "1,3,10.00,20.50,30.12,40.99,1,-1,-1,-1"
    r = mot_io.read_results(p)
    assert r.shape == (3, 7) and r[1, 2] == 0.0 and r[0, 6] == 1.0


def test_bytetrack_format_skips_negative_ids(tmp_path):
    res = [(1, [np.array([1.26, 2.04, 3.0, 4.96])], [7], [0.912]), (2, [np.array([1.0, 2.0, 3.0, 4.0])] * 2, [-1, 8], [0.5, 0.555])]
    p = tmp_path / "b.txt"
    mot_io.write_results_bytetrack(p, res)
    lines = p.read_text().splitlines()
    assert lines == ["1,7,1.3,2.0,3.0,5.0,0.91,-1,-1,-1", "2,8,1.0,2.0,3.0,4.0,0.56,-1,-1,-1"] or lines[1].endswith("0.55,-1,-1,-1")


def test_compare_results(tmp_path):
    rows = [[1, 3, 10.0, 20.0, 30.0, 40.0], [1, 4, 1.0, 2.0, 3.0, 4.0]]
    a, b, c = tmp_path / "a.txt", tmp_path / "b.txt", tmp_path / "c.txt"
    mot_io.write_results_strongsort(a, rows)
    mot_io.write_results_strongsort(b, rows[::-1])              # order within a frame does not matter
    mot_io.write_results_strongsort(c, [rows[0], [1, 4, 1.0, 2.0, 3.0, 4.5]])
    assert mot_io.compare_results(a, b)["identical"]
    d = mot_io.compare_results(a, c)
    assert not d["identical"] and d["first_difference"][:2] == (1, 4)


def test_read_detections(tmp_path):
    p = tmp_path / "det.txt"
    p.write_text("1,-1,10,20,30,40,0.9,-1,-1,-1\n1,-1,11,21,31,41,0.2,-1,-1,-1\n3,-1,1,2,3,4,0.7,-1,-1,-1\n")
    d = mot_io.read_detections(p, min_confidence=0.5)
    assert sorted(d) == [1, 3] and d[1].shape == (1, 5) and d[3][0, 4] == 0.7
