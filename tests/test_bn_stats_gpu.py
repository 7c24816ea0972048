"""GPU unit test of busca_bn_stats_1x1 (reid_gram.hip.inc): BatchNorm statistics of a 1x1 conv from the Gram matrix
of its input, against the direct float64 computation on the same fp16 operands."""
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


def _reference(x16, in_ss, stride, w16, gamma, beta):
    x = x16.astype(np.float32)
    if in_ss is not None:
        fma = (x.astype(np.float64) * in_ss[:, 0].astype(np.float64) + in_ss[:, 1].astype(np.float64)).astype(np.float32)   # == fmaf up to double rounding
        x = np.maximum(fma, 0.0).astype(np.float16).astype(np.float32)
    xs = x[:, ::stride, ::stride, :].reshape(-1, x.shape[-1]).astype(np.float64)
    y = xs @ w16.astype(np.float64).T
    mean, var = y.mean(0), y.var(0)
    sc = gamma.astype(np.float64) / np.sqrt(var + 1e-5)
    return np.stack([sc, beta.astype(np.float64) - mean * sc], 1)


@pytest.mark.parametrize("n,H,W,Cin,Cout,stride,transform", [
    (3, 12, 8, 64, 256, 1, True),       # layer1 conv3: 4-channel lanes, single group
    (2, 9, 7, 64, 256, 1, False),       # ragged pixel count (not a multiple of 32)
    (5, 8, 6, 128, 512, 1, True),       # one 128-channel group
    (2, 10, 16, 256, 512, 2, False),    # downsample: strided sampling (OW = 8), 3 group pairs
    (3, 6, 4, 256, 1024, 1, True),
    (2, 6, 32, 512, 1024, 2, False),    # 10 group pairs, OW = 16
    (40, 24, 8, 128, 512, 1, True),     # several chunks per pair
])
def test_bn_stats_1x1(ctx, n, H, W, Cin, Cout, stride, transform):
    seed = 1000 + Cin + Cout + n
    x16 = (synth.normal(seed, "x", (n, H, W, Cin)) * 1.5).astype(np.float16)
    w16 = (synth.normal(seed, "w", (Cout, Cin)) * (1.0 / np.sqrt(Cin))).astype(np.float16)
    gamma = (1.0 + 0.1 * synth.normal(seed, "g", (Cout,))).astype(np.float32)
    beta = (0.1 * synth.normal(seed, "b", (Cout,))).astype(np.float32)
    in_ss = None
    if transform:
        in_ss = np.stack([1.0 + 0.2 * synth.normal(seed, "s", (Cin,)), 0.3 * synth.normal(seed, "t", (Cin,))], 1).astype(np.float32)
    dev = torch.device("cuda", 0)
    tx = torch.from_numpy(x16).to(dev); tw = torch.from_numpy(w16).to(dev)
    tg = torch.from_numpy(gamma).to(dev); tb = torch.from_numpy(beta).to(dev)
    tss = torch.from_numpy(in_ss).to(dev) if transform else None
    out = torch.zeros(Cout, 2, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    ctx.check(ctx.lib.busca_bn_stats_1x1(ctx.h, tx.data_ptr(), tss.data_ptr() if transform else None, n, H, W, Cin, stride,
                                         tw.data_ptr(), Cout, tg.data_ptr(), tb.data_ptr(), out.data_ptr(), s))
    got = out.cpu().numpy().astype(np.float64)
    ref = _reference(x16, in_ss, stride, w16, gamma, beta)
    # identical fp16 operands on both sides -> float32-roundoff agreement (the transform's emulated fma can differ from
    # fmaf by a double rounding on a handful of inputs)
    tol = 1e-4 if transform else 2e-5
    assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), (np.abs(got - ref).max(), np.abs(ref).max())


def test_bn_stats_rejects_bad_shapes(ctx):
    dev = torch.device("cuda", 0)
    z = torch.zeros(16, device=dev)
    rc = ctx.lib.busca_bn_stats_1x1(ctx.h, z.data_ptr(), None, 1, 2, 2, 96, 1, z.data_ptr(), 8, z.data_ptr(), z.data_ptr(), z.data_ptr(), None)
    assert rc == -1 and b"unsupported shape" in ctx.lib.busca_last_error(ctx.h)
