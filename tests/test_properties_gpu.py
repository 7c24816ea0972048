"""GPU: size-independent properties at BASELINE.json's full sizes (no oracle needed) and C-ABI error paths."""
import ctypes as C

import numpy as np
import pytest
import torch

from busca_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from busca_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def _fwd(m, inp, **kw):
    out = m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"], **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("cfg", [(256, 16, 256, "f32"), (256, 16, 256, "f16"), (128, 32, 512, "f16"), (512, 64, 512, "f16")],
                         ids=["cfgN_x8_f32", "cfgN_x8_f16", "cfg4_128x32_d512", "cfg5_512x64_d512"])
def test_full_size_properties(ctx, cfg):
    """Rows are probability vectors, argmax is the first maximum, tracks are independent of their batch, and the
    network is equivariant to a permutation of a track's candidate slots (SEP/CAN tokens carry no slot index:
    busca/encodings.py:150-180 gives every candidate the same temporal bucket)."""
    from busca_amd.dt import DecisionTransformerHIP
    B, P, d, prec = cfg
    L = 11
    sd = synth.dt_state_dict(50 + P, d=d, ff=2 * d)
    inp = synth.dt_inputs(50 + P, B, L, P)
    m = DecisionTransformerHIP(ctx, sd, precision=prec)
    out = _fwd(m, inp)
    pr = out["probs"]
    assert np.isfinite(out["logits"]).all() and np.isfinite(pr).all()
    assert np.abs(pr.sum(-1) - 1).max() < 1e-5 and (pr >= 0).all()
    assert (out["argmax"] == pr.argmax(-1)).all()
    # batch independence (bit-exact: one workgroup / one row block per track never mixes tracks)
    sel = [3, B // 2, B - 1]
    sub = {k: v[sel] for k, v in inp.items()}
    part = _fwd(m, sub)
    # fused path: one workgroup per track; tiled path: 64-/128-row GEMM tiles span tracks but the summation order inside a row
    # does not depend on the tile -> exact on both
    assert np.array_equal(part["logits"], out["logits"][sel])
    # candidate permutation equivariance
    perm = np.arange(P)[::-1].copy()
    pin = dict(inp)
    pin["can_feat"] = inp["can_feat"][:, perm]
    pin["can_boxes"] = inp["can_boxes"][:, perm]
    pout = _fwd(m, pin)
    tol = 2e-4 if prec == "f32" else 3e-2
    assert np.abs(pout["logits"][:, :P] - out["logits"][:, perm]).max() <= tol
    assert np.abs(pout["logits"][:, P:] - out["logits"][:, P:]).max() <= tol


def test_empty_and_error_paths(ctx):
    from busca_amd import _lib
    from busca_amd.dt import DecisionTransformerHIP
    lib, h = ctx.lib, ctx.h
    sd = synth.dt_state_dict(3, d=64, ff=128)
    m = DecisionTransformerHIP(ctx, sd, precision="f32")
    inp = synth.dt_inputs(3, 4, 11, 5)
    t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    logits = torch.empty(4, 7, device="cuda")
    # B = 0 is a no-op
    assert lib.busca_dt_forward(h, t["mem_feat"].data_ptr(), t["can_feat"].data_ptr(), t["mem_boxes"].data_ptr(), t["can_boxes"].data_ptr(),
                                0, 11, 5, logits.data_ptr(), None, None, None, None, None) == 0
    # null logits / bad shapes are refused with a message
    assert lib.busca_dt_forward(h, t["mem_feat"].data_ptr(), t["can_feat"].data_ptr(), t["mem_boxes"].data_ptr(), t["can_boxes"].data_ptr(),
                                4, 11, 5, None, None, None, None, None, None) == -1
    assert b"logits" in lib.busca_last_error(h)
    assert lib.busca_dt_forward(h, None, t["can_feat"].data_ptr(), t["mem_boxes"].data_ptr(), t["can_boxes"].data_ptr(),
                                4, 11, 5, logits.data_ptr(), None, None, None, None, None) == -1
    # wrong blob size
    cfg = _lib.DTCfg(64, 128, 4, 4, 512, 0, 1, 0)
    blob = np.zeros(10, np.float32)
    lut = np.zeros((211, 22), np.uint16)
    assert lib.busca_dt_load_weights(h, C.byref(cfg), blob.ctypes.data, blob.size, lut.ctypes.data, lut.ctypes.data, lut.ctypes.data, 22) == -1
    # a fresh context refuses forward before load
    c2 = _lib.Context(0)
    assert c2.lib.busca_dt_forward(c2.h, t["mem_feat"].data_ptr(), t["can_feat"].data_ptr(), t["mem_boxes"].data_ptr(), t["can_boxes"].data_ptr(),
                                   4, 11, 5, logits.data_ptr(), None, None, None, None, None) == -2
    assert c2.lib.busca_reid_forward(c2.h, t["mem_feat"].data_ptr(), 1, logits.data_ptr(), None) == -2
    assert c2.lib.busca_pairwise(c2.h, None, 3, None, 3, 0, None, None, None) == -1
    assert c2.lib.busca_pairwise(c2.h, None, 0, None, 3, 0, None, None, None) == 0       # empty -> no-op
    assert c2.lib.busca_topk_rows(c2.h, None, 0, 0, 5, None, None) == 0
    c2.close()


def test_kernel_timing_facility(ctx):
    from busca_amd.dt import DecisionTransformerHIP
    lib, h = ctx.lib, ctx.h
    m = DecisionTransformerHIP(ctx, synth.dt_state_dict(3, d=64, ff=128), precision="f16")
    inp = synth.dt_inputs(3, 8, 11, 5)
    lib.busca_timing_read(h, None, None, 1)
    lib.busca_timing_enable(h, 1)
    for _ in range(5):
        m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])
    torch.cuda.synchronize()
    avg, n = C.c_double(0), C.c_int64(0)
    lib.busca_timing_read(h, C.byref(avg), C.byref(n), 1)
    lib.busca_timing_enable(h, 0)
    assert n.value == 5 and 0 < avg.value < 5.0


def test_cfg4_full_step_on_gpu_crops():
    """BASELINE configs[3] (MOT20 dense crowd: 128 lost x 32 proposals, crops cut and gathered on the GPU, BatchNorm batches of
    1 408 and 4 096 crops, T = 79 tokens): the whole step runs; the Decision-Transformer leg is checked against the oracle on
    the features the HIP extractor produced (the extractor itself is oracle-checked at 352 / 512 crops in test_reid_gpu.py -
    4 096 crops are 33 TFLOP on the CPU), features are unit vectors, and the step is deterministic."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cfg4_step
    from oracle import dt as odt
    a = cfg4_step.run(1, "f16", check=True)
    b = cfg4_step.run(1, "f16", check=True)
    assert np.array_equal(a["_out"]["probs"], b["_out"]["probs"]) and np.array_equal(a["_feat"][1], b["_feat"][1])
    mf, cf = a["_feat"]
    assert mf.shape == (128, 11, 512) and cf.shape == (128, 32, 512)
    assert np.allclose(np.linalg.norm(cf, axis=-1), 1.0, atol=1e-4) and np.isfinite(mf).all()
    inp = a["_inp"]
    ref = odt.dt_forward(a["_sd"], odt.DTConfig(d=512, ff=1024), mf, cf, inp["mem_boxes"], inp["can_boxes"], return_all=True)
    assert np.abs(a["_out"]["probs"] - ref["probs"].numpy()).max() <= 5e-3
    assert np.abs(a["_out"]["logits"] - ref["logits"].numpy()).max() <= 6e-2
    assert a["crops_per_step"] == 128 * 43 and a["crops_computed"] <= 128 * 11 + 160
    # the expanded candidate batch (4 096 crops, as the reference builds it) gives the same step up to summation order
    e = cfg4_step.run(1, "f16", check=True, dedup=False)
    assert (e["_feat"][1] * cf).sum(-1).min() >= 0.9998 and np.abs(e["_feat"][1] - cf).max() <= 5e-3
    assert np.abs(e["_out"]["probs"] - a["_out"]["probs"]).max() <= 5e-3


def test_new_entry_points_empty_and_error_paths(ctx):
    """Round-2 C-ABI entries: empty inputs are no-ops, bad arguments are refused with a message (never a crash or a silent pass)."""
    import ctypes as C
    from busca_amd import _lib
    lib, h = ctx.lib, ctx.h
    out = torch.zeros(2, 384, 128, 3, dtype=torch.uint8, device="cuda")
    assert lib.busca_gather_crops(h, None, 0, None, None) == 0
    assert lib.busca_gather_crops(h, None, 2, out.data_ptr(), None) == -1 and b"null" in lib.busca_last_error(h)
    assert lib.busca_gather_crops(h, out.data_ptr(), -1, out.data_ptr(), None) == -1
    zero_src = torch.zeros(2, dtype=torch.int64, device="cuda")                      # address 0 = zero crop
    out.fill_(7)
    assert lib.busca_gather_crops(h, zero_src.data_ptr(), 2, out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert int(out.max()) == 0
    frame = torch.zeros(64, 64, 3, dtype=torch.uint8, device="cuda")
    rects = torch.zeros(1, 4, dtype=torch.int32, device="cuda")
    assert lib.busca_crop_gather_ex(h, frame.data_ptr(), 64, 64, 192, rects.data_ptr(), 0, None, None, None, None) == 0
    assert lib.busca_crop_gather_ex(h, frame.data_ptr(), 64, 64, 192, rects.data_ptr(), 1, None, None, None, None) == -1
    assert lib.busca_crop_gather_ex(h, frame.data_ptr(), 64, 64, 10, rects.data_ptr(), 1, None, out.data_ptr(), None, None) == -1
    c2 = _lib.Context(0)
    feats = torch.zeros(2, 512, device="cuda")
    assert c2.lib.busca_reid_forward_w(c2.h, out.data_ptr(), 2, None, None, 0.0, feats.data_ptr(), None) == -2           # no weights loaded
    assert c2.lib.busca_dt_reserve(c2.h, 4, 11, 5, None) == -2
    warp = np.eye(2, 3, dtype=np.float32)
    cc = C.c_double(0)
    assert c2.lib.busca_ecc_align(c2.h, None, frame.data_ptr(), 64, 64, 192, 192, 0, 10, 1e-5, warp.ctypes.data, C.byref(cc), None, None) == -1
    assert c2.lib.busca_ecc_align(c2.h, frame.data_ptr(), frame.data_ptr(), 64, 64, 192, 192, 2, 10, 1e-5, warp.ctypes.data, C.byref(cc), None, None) == -1
    # a constant image pair has no gradient: OpenCV raises, the kernel path returns an error code with a message
    assert c2.lib.busca_ecc_align(c2.h, frame.data_ptr(), frame.data_ptr(), 64, 64, 192, 192, 0, 10, 1e-5, warp.ctypes.data, C.byref(cc), None, None) == -1
    assert len(c2.lib.busca_last_error(c2.h)) > 0
    c2.close()
    # weighted forward: weight_sum must be given with the weights
    from busca_amd.reid import ReIDEncoderHIP
    m = ReIDEncoderHIP(ctx, synth.reid_state_dict(3))
    w = torch.ones(2, device="cuda")
    assert lib.busca_reid_forward_w(h, out.data_ptr(), 2, None, w.data_ptr(), 0.0, feats.data_ptr(), None) == -1
    assert lib.busca_reid_forward_w(h, out.data_ptr(), 0, None, None, 0.0, feats.data_ptr(), None) == 0
    del m
