"""GPU: a MOT-format sequence end to end through the drop-in BUSCA (busca_amd/harness.py + tools/run_mot.py)."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    from busca_amd.network import BUSCA
    a = types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=512, ff_size=1024, activation="gelu", dropout_p=0.1,
                              input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                              encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), precision="f16", seed=7)
    return BUSCA(a).to(torch.device("cuda:0")).eval()


def _targs(busca_thresh):
    # raw probabilities (not the one-hot winner): with random weights the argmax is arbitrary, a tiny threshold on the raw
    # probability of the track's own prediction still exercises the whole recovery path
    return types.SimpleNamespace(seq_len=11, num_candidates=5, use_broader_memory=True, select_highest_candidate=False,
                                 busca_thresh=busca_thresh, match_thresh=0.8, track_thresh=0.5, det_thresh=0.1, max_time_lost=30)


def test_sequence_runs_end_to_end_and_is_reproducible(tmp_path):
    from busca_amd import harness
    seq = harness.load_sequence(harness.write_synthetic_sequence(str(tmp_path / "data"), n_frames=50, n_objects=5))
    model = _model()
    outs = []
    for run in ("a", "b"):
        trk = harness.LiteTracker(model, _targs(1e-6))          # tiny threshold: random weights still recover something
        n = harness.run_sequence(seq, trk, str(tmp_path / run / (seq.name + ".txt")))
        assert n > 0
        outs.append(trk.recovered)
    assert outs[0] == outs[1] and outs[0] > 0                   # the BUSCA stage ran and kept lost tracks alive
    cmp = harness.compare_runs(str(tmp_path / "a"), str(tmp_path / "b"))
    assert cmp["all_identical"], cmp                            # deterministic: two runs give byte-identical result files
    sc = harness.evaluate(str(tmp_path / "a"), [seq])[seq.name]
    assert sc["MOTA"] > 0.5 and 0 < sc["IDF1"] <= 1.0, sc
    # without BUSCA the occlusion gaps stay misses
    off = harness.LiteTracker(model, _targs(0.0))
    harness.run_sequence(seq, off, str(tmp_path / "off" / (seq.name + ".txt")))
    sc_off = harness.evaluate(str(tmp_path / "off"), [seq])[seq.name]
    assert off.recovered == 0 and sc_off["FN"] > sc["FN"]


def test_run_mot_cli_synthetic(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_mot.py"), "--synthetic", str(tmp_path / "syn"), "--out", str(tmp_path / "out"),
                        "--busca-thresh", "1e-6", "--max-frames", "30"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout)
    assert "SYN-01" in rep["scores"] and rep["scores"]["SYN-01"]["GT"] > 0
    assert os.path.exists(tmp_path / "out" / "SYN-01.txt")
