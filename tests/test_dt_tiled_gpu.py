"""GPU parity of the tiled (layer-wise) Decision-Transformer path: forced on the golden shapes, and on the shapes
only it can run (BASELINE configs 3 and 4: 128 x 32 -> T = 79 and 64 proposals x d512 -> T = 143)."""
import glob
import os

import numpy as np
import pytest
import torch

from busca_amd import synth

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dt_*.npz")))
TOL = dict(logit=6e-2, prob=5e-3, att=5e-3, hidden=8e-2, margin=2e-2)     # f16-operand tolerances (as test_dt_gpu.py)
TOL32 = dict(logit=2e-4, prob=2e-5, att=2e-5, hidden=5e-4, margin=1e-4)   # float32 tolerances of test_dt_gpu.py


@pytest.fixture(scope="module")
def ctx():
    from busca_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture()
def force_tiled(ctx):
    ctx.set_option("dt_tiled", 1)          # per-context option, read at every forward (include/busca_hip.h: busca_set_option)
    yield
    ctx.set_option("dt_tiled", 0)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
@pytest.mark.parametrize("mode", ["f64", "f32"])
@pytest.mark.parametrize("prec", ["f16", "f32", "x3"])
def test_tiled_vs_reference_golden(ctx, force_tiled, path, mode, prec):
    """(x3: the split-fp16 layer kernels at d >= 256 - float32-equivalent, held to the f32 bars; d = 64 runs the exact f32 kernels.)"""
    from busca_amd.dt import DecisionTransformerHIP
    TOL = TOL32 if prec in ("f32", "x3") else globals()["TOL"]
    g = np.load(path)
    d, ff, B, L, P, seed = (int(g[k]) for k in ("d", "ff", "B", "L", "P", "seed"))
    sd = synth.dt_state_dict(seed, d=d, ff=ff)
    inp = synth.dt_inputs(seed, B, L, P, sentinel_every=4 if B <= 8 else 16)
    has_att = ("att_" + mode) in g
    m = DecisionTransformerHIP(ctx, sd, fake_bbox_f64=(mode == "f64"), precision=prec)
    out = m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"], want_hidden=True, want_att=has_att)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    assert np.abs(out["logits"] - g["logits_" + mode]).max() <= TOL["logit"]
    assert np.abs(out["probs"] - g["probs_" + mode]).max() <= TOL["prob"]
    pos = [L + 2 * j + 1 for j in range(P + 2)]
    assert np.abs(out["hidden"][:, pos] - g["can_hidden_" + mode]).max() <= TOL["hidden"]
    if has_att:
        assert np.abs(out["att"] - g["att_" + mode]).max() <= TOL["att"]
    ref_p = g["probs_" + mode]
    srt = np.sort(ref_p, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > TOL["margin"]
    assert (out["argmax"][clear] == g["argmax_" + mode][clear]).all()
    assert (out["argmax"] == out["probs"].argmax(-1)).all()


@pytest.mark.parametrize("shape", [(128, 11, 32, 512), (24, 11, 64, 512), (40, 11, 40, 256), (9, 11, 62, 64)],
                         ids=["cfg4_128x32_d512", "cfg5_shape_x64_d512", "T95_d256", "T139_d64"])
@pytest.mark.parametrize("prec", ["f16", "f32", "x3"])
def test_tiled_large_shapes_vs_oracle(ctx, shape, prec):
    """Shapes the fused kernel cannot hold (dispatch picks the tiled path by itself), in both arithmetic types: the f32
    flavour (v_mfma_f32_16x16x4_f32, the reference's own precision, busca/custom_layers.py:30-41) to the float32
    tolerances of test_dt_gpu.py - BASELINE configs[3] (128 x 32, T = 79) at reference precision."""
    from busca_amd.dt import DecisionTransformerHIP
    from oracle import dt as odt
    TOL = TOL32 if prec in ("f32", "x3") else globals()["TOL"]
    B, L, P, d = shape
    seed = 300 + P + d
    sd = synth.dt_state_dict(seed, d=d, ff=2 * d)
    inp = synth.dt_inputs(seed, B, L, P, sentinel_every=8)
    m = DecisionTransformerHIP(ctx, sd, precision=prec)
    m.reserve(B, L, P)                                   # workspace sized ahead of time: the forward does not allocate
    out = m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    ref = odt.dt_forward(sd, odt.DTConfig(d=d, ff=2 * d), **inp, return_all=True)
    assert np.abs(out["logits"] - ref["logits"].numpy()).max() <= TOL["logit"]
    assert np.abs(out["probs"] - ref["probs"].numpy()).max() <= TOL["prob"]
    rp = ref["probs"].numpy()
    srt = np.sort(rp, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > TOL["margin"]
    assert (out["argmax"][clear] == ref["argmax"].numpy()[clear]).all()


@pytest.mark.parametrize("shape", [(16, 11, 32, 512), (6, 11, 30, 256), (5, 11, 64, 512), (4, 11, 40, 64)], ids=lambda s: "B%d_L%d_P%d_d%d" % s)
def test_x3_beyond_the_one_kernel_shapes(ctx, shape):
    """The split-fp16 flavour (the default) beyond the one-kernel path (round 6): the two fused layer kernels run their GEMMs as three fp16 MFMAs per product
    block on hi / lo operands (d >= 256; T <= 80 for the QKV + attention kernel - at T = 143 that half of a layer runs the exact f32 kernels), d = 64 runs the
    exact f32 layer-wise path (bit-identical to the f32 flavour).  Float32-equivalent: logits within 5e-5 of the exact f32 flavour's, same argmax outside a
    1e-4 margin, `dt_status` 0; and `dt_exact_f32` gives the f32 flavour's bits on the same context (the re-run route of a clipped step)."""
    from busca_amd.dt import DecisionTransformerHIP
    B, L, P, d = shape
    sd = synth.dt_state_dict(70 + d, d=d, ff=2 * d)
    inp = synth.dt_inputs(70 + d, B, L, P, sentinel_every=4)
    outs = {}
    for prec in ("f32", "x3"):
        m = DecisionTransformerHIP(ctx, sd, activation="relu", precision=prec)
        o = m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"], want_hidden=True)
        torch.cuda.synchronize()
        outs[prec] = {k: v.cpu().numpy() for k, v in o.items()}
        assert ctx.get_option("dt_status") == 0
        if prec == "x3":
            ctx.set_option("dt_exact_f32", 1)
            try:
                e = m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"], want_hidden=True)
                torch.cuda.synchronize()
            finally:
                ctx.set_option("dt_exact_f32", 0)
            for k in ("logits", "probs", "argmax", "hidden"):
                assert np.array_equal(e[k].cpu().numpy(), outs["f32"][k]), k
    if d == 64:
        for k in ("logits", "probs", "argmax", "hidden"):
            assert np.array_equal(outs["f32"][k], outs["x3"][k]), k
    else:
        assert not np.array_equal(outs["f32"]["logits"], outs["x3"]["logits"])            # it IS another kernel
        assert np.abs(outs["f32"]["logits"] - outs["x3"]["logits"]).max() <= 5e-5
        assert np.abs(outs["f32"]["probs"] - outs["x3"]["probs"]).max() <= 5e-6
        assert np.abs(outs["f32"]["hidden"] - outs["x3"]["hidden"]).max() <= 1e-4
        srt = np.sort(outs["f32"]["probs"], axis=-1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-4
        assert (outs["f32"]["argmax"][clear] == outs["x3"]["argmax"][clear]).all()
    assert np.abs(outs["x3"]["logits"]).max() > 0 and np.ptp(outs["x3"]["logits"]) > 1e-3


@pytest.mark.parametrize("flavour,sep_ref", [("MEM-CAN-SEP", False), ("MEM-SEP-CAN", True), ("MEM-CAN-SEP-BAD", True), ("MEM-SEP-CAN-BAD", False)])
@pytest.mark.parametrize("prec", ["f32", "x3", "f16"])
@pytest.mark.parametrize("d", [256, 512])
def test_tiled_token_layouts_at_the_fused_layer_kernels_widths(ctx, force_tiled, flavour, sep_ref, prec, d):
    """The layer-wise path at d >= 256 has kernels of its own for the embed pass (compacted feature rows, special rows filled by the workgroups behind them) and for the
    decoder (per-row logits from the last layer kernel + dtl_decoder_rows_kernel): every token layout of network.py:103-165 - candidate before / after its separator,
    with / without the BAD token, separators encoded as the reference box or as their candidate's - against the oracle, hidden states of every token included (the
    golden fixtures of these layouts are d = 64, which runs the generic kernels)."""
    from busca_amd.dt import DecisionTransformerHIP
    from oracle import dt as odt
    tol = TOL32 if prec in ("f32", "x3") else TOL
    B, L, P, seed = 7, 11, 9, 400 + d
    sd = synth.dt_state_dict(seed, d=d, ff=2 * d, flavour=flavour)
    inp = synth.dt_inputs(seed, B, L, P, sentinel_every=3)
    m = DecisionTransformerHIP(ctx, sd, activation="relu", precision=prec, input_flavour=flavour, encode_separator_as_reference=sep_ref)
    out = m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"], want_hidden=True)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in out.items()}
    ref = odt.dt_forward(sd, odt.DTConfig(d=d, ff=2 * d, flavour=flavour, encode_sep_as_ref=sep_ref), **inp, return_all=True)
    n = P + (2 if "BAD" in flavour else 1)
    assert out["logits"].shape == (B, n) and out["hidden"].shape == (B, L + 2 * n, d)
    assert np.abs(out["logits"] - ref["logits"].numpy()).max() <= tol["logit"]
    assert np.abs(out["probs"] - ref["probs"].numpy()).max() <= tol["prob"]
    assert np.abs(out["hidden"] - ref["hidden"].numpy()).max() <= tol["hidden"]
    rp = ref["probs"].numpy()
    srt = np.sort(rp, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > tol["margin"]
    assert (out["argmax"][clear] == ref["argmax"].numpy()[clear]).all() and (out["argmax"] == out["probs"].argmax(-1)).all()
    assert ctx.get_option("dt_status") == 0


def test_unsupported_shape_is_refused_loudly(ctx):
    """More than 144 tokens per track is beyond every path: a BuscaError, never a silent fallback."""
    from busca_amd import _lib
    from busca_amd.dt import DecisionTransformerHIP
    sd = synth.dt_state_dict(1, d=256, ff=512)
    inp = synth.dt_inputs(1, 2, 11, 70)
    for prec in ("f32", "f16"):
        m = DecisionTransformerHIP(ctx, sd, precision=prec)
        with pytest.raises(_lib.BuscaError):
            m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
