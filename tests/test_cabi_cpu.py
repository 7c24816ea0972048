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
    assert lib.busca_version() // 1000 == 2         # 2000: busca_dt_cfg.layout is part of the ABI, BUSCA_PREC_F16X3
    for d in (64, 256, 512):
        cfg = _lib.DTCfg(d, 2 * d, 4, 4, 512, 0, 1, 0)
        blob = weights.dt_blob(synth.dt_state_dict(1, d=d, ff=2 * d), 4)
        assert lib.busca_dt_blob_floats(ctypes.byref(cfg)) == blob.size
    bad = _lib.DTCfg(100, 200, 4, 4, 512, 0, 1, 0)
    assert lib.busca_dt_blob_floats(ctypes.byref(bad)) == 0
    # model options beyond the shipped geometry: accepted / refused exactly as include/busca_hip.h says
    def floats(d, ff, nhead, layout=0):
        return lib.busca_dt_blob_floats(ctypes.byref(_lib.DTCfg(d, ff, nhead, 4, 512, 0, 1, 0, layout)))
    assert floats(256, 1024, 8) == weights.dt_blob(synth.dt_state_dict(1, d=256, ff=1024), 4).size      # head width 32, ff = 4 d
    assert floats(64, 64, 2) > 0 and floats(512, 512, 4, _lib.LAYOUT_CAN_FIRST | _lib.LAYOUT_NO_BAD | _lib.LAYOUT_SEP_AS_CAN) > 0
    assert floats(64, 128, 8) == 0            # head width 8
    assert floats(256, 384, 4) == 0           # ff not a multiple of d
    assert floats(256, 9 * 256, 4) == 0       # ff > 8 d
    assert floats(256, 512, 4, 8) == 0        # unknown layout bit
    # a state_dict without the BAD token (input flavours without -BAD) packs into the same blob size
    assert weights.dt_blob(synth.dt_state_dict(1, d=64, ff=128, flavour="MEM-SEP-CAN"), 4).size == floats(64, 128, 4, _lib.LAYOUT_NO_BAD)
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


def _args(d=64):
    import types, torch
    return types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=d, ff_size=2 * d, activation="gelu", dropout_p=0.1,
                                 input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                                 encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"))


def test_load_pretrained_checkpoint_formats(tmp_path):
    """BUSCA.load_pretrained (network.py:432-467): raw state_dict or {'model_state_dict', 'optimizer_state_dict'};
    ReID classifier / whole-ReID dropping; foreign keys (cls_token, BN buffers, fc) are ignored.  No GPU needed."""
    import numpy as np, torch
    from busca_amd import synth
    from busca_amd.network import BUSCA
    m = BUSCA(_args())
    sd = {k: torch.from_numpy(v) for k, v in synth.dt_state_dict(99, d=64, ff=128).items()}
    sd.update({"reid_encoder.model." + k: torch.from_numpy(v) for k, v in synth.reid_state_dict(99, with_fc=True).items()})
    sd["cls_token"] = torch.zeros(64)
    sd["reid_encoder.model.bn1.running_mean"] = torch.zeros(64)
    sd["reid_encoder.model.bn1.num_batches_tracked"] = torch.tensor(5)
    before_reid = m._sd["reid_encoder.model.conv1.weight"].copy()
    p1 = tmp_path / "raw.pth"
    torch.save(sd, p1)
    m.load_pretrained(str(p1), ignore_reid_fc=True)
    assert np.array_equal(m._sd["encoder.weight"], sd["encoder.weight"].numpy()) and m._dirty
    assert np.array_equal(m._sd["reid_encoder.model.conv1.weight"], sd["reid_encoder.model.conv1.weight"].numpy())
    assert "reid_encoder.model.fc.weight" not in m._sd and "cls_token" not in m._sd
    m2 = BUSCA(_args())
    p2 = tmp_path / "wrapped.pth"
    torch.save({"model_state_dict": sd, "optimizer_state_dict": {}}, p2)
    m2.load_pretrained(str(p2), ignore_reid=True)
    assert np.array_equal(m2._sd["sep_token"], sd["sep_token"].numpy())
    assert np.array_equal(m2._sd["reid_encoder.model.conv1.weight"], before_reid) or not np.array_equal(
        m2._sd["reid_encoder.model.conv1.weight"], sd["reid_encoder.model.conv1.weight"].numpy())
    # state_dict round trip and size check
    out = m.state_dict()
    assert set(out) == set(m._sd) and out["encoder.weight"].shape == (64, 512)
    import pytest
    bad = dict(sd)
    bad["encoder.weight"] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)


def test_unsupported_configs_raise_like_the_reference():
    import pytest
    from busca_amd.network import BUSCA
    a = _args(); a.input_flavour = "CLS-MEM-SEP-CAN"
    with pytest.raises(NotImplementedError):
        BUSCA(a)
    a = _args(); a.activation = "swish"
    with pytest.raises(RuntimeError):
        BUSCA(a)
    import torch
    a = _args(); a.device = torch.device("cpu")
    with pytest.raises(RuntimeError):
        BUSCA(a)


def test_blob_round_trip_and_decision_rule():
    import numpy as np
    from busca_amd import synth, weights
    from busca_amd.tracking import recover_with_busca
    sd = synth.dt_state_dict(5, d=64, ff=128)
    back = weights.dt_unblob(weights.dt_blob(sd, 4), 64, 128, 4)
    assert list(back) == weights.dt_blob_keys(4) and all(np.array_equal(back[k], sd[k]) for k in back)
    probs = np.zeros((3, 5 + 3)); probs[0, 5] = 0.9; probs[1, 6] = 0.4; probs[2, 7] = 0.95
    matches, unmatched = recover_with_busca(probs, np.array([True, True, False]), 5, 0.5)
    assert matches == [[0, 0.9]] and unmatched == [1, 2]
    assert recover_with_busca(None, None, 5, 0.5) == ([], [])


def test_checkpoint_loading_is_strict_like_the_reference(tmp_path):
    """`model_dict.update(ckpt); load_state_dict(model_dict)` (network.py:465-467, load_trained_net.py:64-66) raises on keys the
    model does not have; here too - a wrong checkpoint must not leave the tracker on its seeded synthetic weights silently."""
    import warnings
    import numpy as np, pytest, torch
    from busca_amd import synth
    from busca_amd.network import BUSCA
    sd = {k: torch.from_numpy(v) for k, v in synth.dt_state_dict(99, d=64, ff=128).items()}
    sd.update({"reid_encoder.model." + k: torch.from_numpy(v) for k, v in synth.reid_state_dict(99).items()})
    # DDP-style prefix: every key is foreign -> raise
    p = tmp_path / "ddp.pth"
    torch.save({"module." + k: v for k, v in sd.items()}, p)
    with pytest.raises(RuntimeError, match="unexpected key"):
        BUSCA(_args()).load_pretrained(str(p))
    # a different flavour (an extra learned token) -> raise
    p = tmp_path / "flavour.pth"
    torch.save(dict(sd, extra_token=torch.zeros(64)), p)
    with pytest.raises(RuntimeError, match="extra_token"):
        BUSCA(_args()).load_pretrained(str(p))
    # a checkpoint that lacks parameters keeps the current values but says so
    p = tmp_path / "partial.pth"
    torch.save({k: v for k, v in sd.items() if not k.startswith("decoder.")}, p)
    m = BUSCA(_args())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.load_pretrained(str(p))
    assert any("decoder" in str(x.message) for x in w)
    assert np.array_equal(m._sd["encoder.weight"], sd["encoder.weight"].numpy())
    # ReID weights file (load_net): the same strictness, fc heads dropped as the reference does
    p = tmp_path / "reid.pth"
    rsd = {k: torch.from_numpy(v) for k, v in synth.reid_state_dict(5, with_fc=True).items()}
    torch.save(rsd, p)
    a = _args(); a.reid_weights_file = str(p)
    m = BUSCA(a)
    assert np.array_equal(m._sd["reid_encoder.model.layer1.0.conv1.weight"], rsd["layer1.0.conv1.weight"].numpy())
    torch.save(dict(rsd, **{"backbone.conv9.weight": torch.zeros(3)}), p)
    with pytest.raises(RuntimeError, match="unexpected key"):
        BUSCA(a)
    # strict load_state_dict: missing keys raise
    with pytest.raises(RuntimeError, match="missing"):
        BUSCA(_args()).load_state_dict({"encoder.weight": sd["encoder.weight"]})


def test_x3_activation_range_check_of_a_checkpoint():
    """The split-fp16 ReID flavour clamps staged activations at |x| <= 1023.5 (reid_x3.hip.inc).  The bound derived from a checkpoint's BatchNorm affines
    (weights.x3_activation_bound) is far below that for ordinary weights and trips for a BatchNorm with a huge gamma - the case ReIDEncoderHIP warns about
    instead of clipping silently (busca/reid/resnet.py:108-128: relu(bn(.)) and the residual sum are what a conv reads)."""
    from busca_amd import synth, weights
    sd = dict(synth.reid_state_dict(3))
    bound, where = weights.x3_activation_bound(sd)
    assert 0 < bound < weights.X3_OPERAND_LIMIT, (bound, where)
    g = sd["layer3.2.bn2.weight"].copy()
    g[7] = 40.0
    sd["layer3.2.bn2.weight"] = g
    bound, where = weights.x3_activation_bound(sd)
    assert bound > weights.X3_OPERAND_LIMIT and where == "layer3.2.bn2", (bound, where)
