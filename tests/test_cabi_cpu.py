"""CPU: the C-ABI library loads and exports every symbol include/busca_hip.h declares; host-side entry
points that need no GPU behave (blob sizes, error codes).  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from busca_amd.build import build
    build()
    from busca_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "busca_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(busca_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 15
    from busca_amd import _lib
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)
    for n in names:
        assert getattr(lib, n) is not None


def test_blob_sizes_and_version(lib):
    from busca_amd import _lib, synth, weights
    assert lib.busca_version() >= 1000
    for d in (64, 256, 512):
        cfg = _lib.DTCfg(d, 2 * d, 4, 4, 512, 0, 1, 0)
        blob = weights.dt_blob(synth.dt_state_dict(1, d=d, ff=2 * d), 4)
        assert lib.busca_dt_blob_floats(ctypes.byref(cfg)) == blob.size
    bad = _lib.DTCfg(100, 200, 4, 4, 512, 0, 1, 0)
    assert lib.busca_dt_blob_floats(ctypes.byref(bad)) == 0
    assert lib.busca_reid_blob_floats() == weights.reid_blob(synth.reid_state_dict(1)).size
    assert lib.busca_reid_workspace_bytes(8) > 8 * 5_000_000


def test_no_gpu_fails_loudly(lib):
    """Without a visible GPU, creating a context reports an error instead of falling back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from busca_amd import _lib
    with pytest.raises(_lib.BuscaError):
        _lib.Context(0)


def test_option_and_alias_package():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "busca_amd", "compat"))
    for m in [k for k in sys.modules if k == "busca" or k.startswith("busca.")]:
        del sys.modules[m]
    from busca.option import load_args_from_config, merge_args
    import types
    trk, trn = load_args_from_config(os.path.join(ROOT, "busca_amd", "configs", "strongsort_mot17.yml"))
    assert trk.transformer.trans_dim == 512 and trk.seq_len == 11 and trk.num_candidates == 5
    assert trn.transformer is trk.transformer and trn.dataset.neg_threshold == 0.5
    merged = merge_args(trk, types.SimpleNamespace(busca_thresh=0.3, seq_len=None, new_flag=True), verbose=False)
    assert merged.busca_thresh == 0.3 and merged.seq_len == 11 and merged.new_flag is True and trk.busca_thresh == 0.5
    from busca.network import BUSCA  # noqa: F401
    from busca.tracking import center_distance, missing_candidate_bbox  # noqa: F401
    from busca.visualization import plot_box  # noqa: F401
    import numpy as np
    assert center_distance([], []).shape == (0, 0)
    assert missing_candidate_bbox(flavour="ltwh").dtype == np.float64
