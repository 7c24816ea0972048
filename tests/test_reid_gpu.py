"""GPU parity: the HIP ReID extractor (fp16 MFMA convs, f32 accumulation and statistics) vs the float32
oracle restatement of the reference's ResNet-50 with train-mode BatchNorm."""
import numpy as np
import pytest
import torch

from busca_amd import synth

pytestmark = pytest.mark.gpu

# Stated tolerance of the fp16-operand path on L2-normalised 512-d features.  Emulating fp16 rounding of
# weights + stored activations inside the float32 oracle (random weights, n=3) gives cosine 0.99927-0.99934
# and max |delta| 6.5e-3 against the unrounded oracle; the kernel lands on the same figures, i.e. the
# deviation is the rounding of the fp16 design, not a defect.
FEAT_ATOL = 1e-2
COS_MIN = 0.9990
# The two float32-class flavours: "f32" = float32 operands on v_mfma_f32_16x16x4_f32 (an fmaf chain), "x3" = float32 activations with
# error-corrected split-fp16 products (three fp16 MFMAs per block, BUSCA_PREC_F16X3).  Both are held to the SAME bars.
EXACT_FLAVOURS = ("f32", "x3")


@pytest.fixture(scope="module")
def ctx():
    from busca_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def model(ctx):
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    return ReIDEncoderHIP(ctx, sd), sd


def _crops(seed, n):
    """Smooth-ish random u8 crops (pure noise would make every crop statistically identical)."""
    base = synth.randint_u8(seed, "crops", (n, 24, 8, 3)).astype(np.float32)
    up = np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)
    noise = synth.randint_u8(seed, "noise", (n, 384, 128, 3)).astype(np.float32) - 128
    return np.clip(up + 0.25 * noise, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("n", [3, 8])
def test_reid_vs_oracle(model, n):
    from oracle import reid as oreid
    m, sd = model
    crops = _crops(40 + n, n)
    got = m.forward(crops).cpu().numpy()
    ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(crops)).numpy()
    assert got.shape == (n, 512)
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-4)
    cos = (got * ref).sum(1)
    assert cos.min() >= COS_MIN, cos
    assert np.abs(got - ref).max() <= FEAT_ATOL, np.abs(got - ref).max()


@pytest.mark.parametrize("n", [3, 40])
def test_stem_pool_with_negative_batchnorm_scales(ctx, n):
    """The stem kernel pools the RAW conv output and leaves BatchNorm + ReLU to the consumers (stem_pool_kernel): exact because
    relu(bn(.)) is monotone per channel - NON-INCREASING where gamma < 0, where the kernel must pool with min instead of max.
    Random weights have gamma ~ 1, so this case flips the sign of every third stem gamma (and zeroes one) and checks the
    features against the float32 oracle, in both precisions of the extractor (the f32 flavour keeps the separate pooling pass)."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = dict(synth.reid_state_dict(3))
    g = sd["bn1.weight"].copy()
    g[::3] *= -1.0
    g[5] = 0.0
    sd["bn1.weight"] = g
    crops = _crops(140 + n, n)
    ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(crops)).numpy()
    got = ReIDEncoderHIP(ctx, sd, precision="f16").forward(crops).cpu().numpy()
    cos = (got * ref).sum(1)
    assert cos.min() >= COS_MIN, cos.min()
    assert np.abs(got - ref).max() <= FEAT_ATOL, np.abs(got - ref).max()
    for prec in EXACT_FLAVOURS:
        got32 = ReIDEncoderHIP(ctx, sd, precision=prec).forward(crops).cpu().numpy()
        assert np.abs(got32 - ref).max() <= 5e-5, (prec, np.abs(got32 - ref).max())


def test_reid_batch_composition_matters(model):
    """Train-mode BN: features depend on the batch they are computed in - and the kernel is
    deterministic for a fixed batch."""
    m, _ = model
    crops = _crops(77, 6)
    full = m.forward(crops).cpu().numpy()
    again = m.forward(crops).cpu().numpy()
    assert np.array_equal(full, again)
    part = m.forward(crops[:3]).cpu().numpy()
    assert np.abs(part - full[:3]).max() > 1e-4


@pytest.mark.parametrize("prec", EXACT_FLAVOURS)
def test_reid_f32_mode_matches_oracle(ctx, prec):
    """Float32-class flavours (f32 activations; v_mfma_f32_16x16x4_f32 or split-fp16 products): float32-roundoff parity with the
    oracle, and they bound the fp16 flavour's deviation on the same batch."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = synth.reid_state_dict(3)
    crops = _crops(43, 3)
    ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(crops)).numpy()
    m32 = ReIDEncoderHIP(ctx, sd, precision=prec)
    got = m32.forward(crops).cpu().numpy()
    print("%s vs oracle at 3 crops: %.2e" % (prec, np.abs(got - ref).max()))
    assert np.abs(got - ref).max() <= 5e-5, np.abs(got - ref).max()
    big = _crops(99, 24)
    f32 = m32.forward(big).cpu().numpy()
    m16 = ReIDEncoderHIP(ctx, sd, precision="f16")
    f16 = m16.forward(big).cpu().numpy()
    assert ((f32 * f16).sum(1) >= COS_MIN).all() and np.abs(f32 - f16).max() <= FEAT_ATOL


def test_reid_golden_reference_features(ctx, golden_dir):
    """Both flavours against features computed by the reference's own ReID_Encoder (tests/golden/reid.npz)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import smooth_crops
    from busca_amd.reid import ReIDEncoderHIP
    g = np.load(os.path.join(golden_dir, "reid.npz"))
    sd = synth.reid_state_dict(3)
    for n, seed in ((3, 43), (5, 45)):
        ref = g["feats_n%d_seed%d" % (n, seed)]
        crops = smooth_crops(seed, n)
        for prec in EXACT_FLAVOURS:
            got32 = ReIDEncoderHIP(ctx, sd, precision=prec).forward(crops).cpu().numpy()
            print("%s vs the reference's features at %d crops: %.2e" % (prec, n, np.abs(got32 - ref).max()))
            assert np.abs(got32 - ref).max() <= 5e-5, (prec, np.abs(got32 - ref).max())
        got16 = ReIDEncoderHIP(ctx, sd, precision="f16").forward(crops).cpu().numpy()
        assert ((got16 * ref).sum(1) >= COS_MIN).all()


@pytest.mark.parametrize("n,seed", [(96, 1096), (200, 1200)])
def test_reid_default_large_batch_schedule_vs_reference(ctx, golden_dir, monkeypatch, n, seed):
    """The DEFAULT schedule at batch sizes where all of it is active (Gram-matrix statistics + fused downsample, halo-resident
    3x3 convs, one-kernel stem, block tails fused with the next conv1) against features computed by the reference's own
    ReID_Encoder (tests/golden/reid_big.npz): fp16 flavour within the stated fp16 tolerance, exact-f32 flavour <= 1e-4."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import smooth_crops
    from busca_amd.reid import ReIDEncoderHIP
    for k in ("BUSCA_REID_GRAM", "BUSCA_REID_HALO", "BUSCA_REID_FUSE_C1", "BUSCA_REID_DIRECT_ROWS",
              "BUSCA_REID_KWAVE_BLOCKS", "BUSCA_REID_KWAVE_HALO", "BUSCA_REID_KWAVE_NW", "BUSCA_REID_KWAVE_PT",
              ):
        monkeypatch.delenv(k, raising=False)
    ref = np.load(os.path.join(golden_dir, "reid_big.npz"))["feats_n%d_seed%d" % (n, seed)]
    sd = synth.reid_state_dict(3)
    crops = smooth_crops(seed, n)
    got16 = ReIDEncoderHIP(ctx, sd, precision="f16").forward(crops).cpu().numpy()
    cos = (got16 * ref).sum(1)
    assert cos.min() >= COS_MIN, cos.min()
    assert np.abs(got16 - ref).max() <= FEAT_ATOL, np.abs(got16 - ref).max()
    # float32 round-off through 53 conv + batch-statistics layers grows with the batch: the reference's own two CPU layouts
    # (channels_last vs contiguous input, see oracle/reid.py) already differ by 2.0e-5 at 96 crops; measured here
    # 5.2e-5 at 200 crops -> stated tolerance 1e-4 for these batch sizes (5e-5 stays the bar for the small batches above)
    for prec in EXACT_FLAVOURS:
        got32 = ReIDEncoderHIP(ctx, sd, precision=prec).forward(crops).cpu().numpy()
        print("%s vs the reference's features at %d crops: %.2e" % (prec, n, np.abs(got32 - ref).max()))
        assert np.abs(got32 - ref).max() <= 1e-4, (prec, np.abs(got32 - ref).max())


def test_reid_cfg4_sized_batches_vs_reference(ctx, golden_dir):
    """BASELINE configs[3]-sized BatchNorm batches against features computed by the reference's own ReID_Encoder in the build
    container (tests/golden/reid_cfg4.npz, make_golden.py reid_cfg4): (a) 1 408 distinct crops through the default large-batch
    schedule, fp16 and exact-f32 flavours; (b) a 1 408-slot candidate batch drawn from 55 distinct crops (the MOT20 duplication
    ratio 4 096 / 160): the reference computed the EXPANDED batch, the extractor computes each distinct crop once with weighted
    statistics (busca_reid_forward_w) - and also the expanded batch itself."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_golden as mg
    from busca_amd.reid import ReIDEncoderHIP
    gold = np.load(os.path.join(golden_dir, "reid_cfg4.npz"))
    sd = synth.reid_state_dict(3)
    n, seed = mg.REID_CFG4_N, mg.REID_CFG4_SEED
    ref = gold["feats_n%d_seed%d" % (n, seed)]
    crops = torch.from_numpy(mg.smooth_crops(seed, n)).cuda()
    slots, distinct, dseed = mg.REID_CFG4_DUP
    dref = gold["dupfeats_slots%d_distinct%d_seed%d" % (slots, distinct, dseed)]
    idx = mg.cfg4_dup_indices()
    dcrops = torch.from_numpy(mg.smooth_crops(dseed, distinct)).cuda()
    counts = np.bincount(idx, minlength=distinct).astype(np.float32)
    # The reference's float32 CPU kernels are themselves 1.7e-3 away from a float64 evaluation of the same network on this batch
    # (tests/golden/reid_cfg4_f64.npz, oracle/reid.py with dtype=float64; 5e-5 at 200 crops): at 4.3 M values per channel its
    # float32 batch statistics carry that much round-off.  The extractor accumulates statistics in float64, so its exact-f32
    # flavour is held to 2e-4 against the float64 result and to the reference's own error band (3e-3) against the reference.
    ref64 = np.load(os.path.join(golden_dir, "reid_cfg4_f64.npz"))["feats64_n%d_seed%d" % (n, seed)]
    # (the committed fixture is ONE run of the reference: regenerating it moves the features by ~1e-3, its float32 batch statistics
    # depend on the run - so only the upper bound of the reference's distance from float64 is a property worth asserting)
    assert np.abs(ref - ref64).max() < 3e-3
    for prec, atol in (("f16", FEAT_ATOL), ("f32", 3e-3), ("x3", 3e-3)):
        m = ReIDEncoderHIP(ctx, sd, precision=prec)
        got = m.forward(crops).cpu().numpy()
        d, d64 = np.abs(got - ref).max(), np.abs(got - ref64).max()
        print("cfg4 batch %s: max |delta| vs reference %.2e, vs float64 %.2e, min cos %.6f" % (prec, d, d64, (got * ref).sum(1).min()))
        assert d <= atol, d
        assert d64 <= (FEAT_ATOL if prec == "f16" else 2e-4), d64
        assert (got * ref).sum(1).min() >= (COS_MIN if prec == "f16" else 0.9999)
        gw = m.forward(dcrops, weights=counts).cpu().numpy()           # 55 crops computed once, statistics weighted by multiplicity
        dw = np.abs(gw - dref).max()
        ge = m.forward(dcrops[torch.from_numpy(idx).cuda()]).cpu().numpy()[: distinct]     # the expanded 1 408-slot batch (first occurrences = first 55 rows)
        de = np.abs(ge - dref).max()
        print("cfg4 duplicated batch %s: weighted %.2e, expanded %.2e" % (prec, dw, de))
        assert dw <= atol and de <= atol, (dw, de)


@pytest.mark.parametrize("n", [352, 512])
def test_reid_benchmarked_batches_vs_oracle(ctx, n):
    """The batch sizes bench.py's full_step times (32 x 11 memory crops, 32 x 16 candidate crops): fp16 default schedule vs
    the float32 oracle run on this host."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = synth.reid_state_dict(3)
    crops = _crops(2000 + n, n)
    got = ReIDEncoderHIP(ctx, sd, precision="f16").forward(crops).cpu().numpy()
    ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(crops)).numpy()
    cos = (got * ref).sum(1)
    assert cos.min() >= COS_MIN, cos.min()
    assert np.abs(got - ref).max() <= FEAT_ATOL, np.abs(got - ref).max()


@pytest.mark.parametrize("n", [5, 24])
def test_reid_gram_statistics_path(ctx, monkeypatch, n):
    """Large batches take BN3 / downsample-BN statistics from the Gram matrix of the conv's input and fuse the downsample
    conv into the block tail (reid_gram.hip.inc).  Forced on at a small batch it must reproduce the direct-statistics
    schedule (same roundings of every stored tensor; statistics differ at the 1e-7 level) and stay on the oracle."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = synth.reid_state_dict(3)
    crops = _crops(300 + n, n)
    monkeypatch.setenv("BUSCA_REID_GRAM", "0")
    direct = ReIDEncoderHIP(ctx, sd).forward(crops).cpu().numpy()
    monkeypatch.setenv("BUSCA_REID_GRAM", "1")
    m = ReIDEncoderHIP(ctx, sd)
    gram = m.forward(crops).cpu().numpy()
    assert np.array_equal(gram, m.forward(crops).cpu().numpy())            # deterministic
    # the two schedules differ only in how the statistics are summed (~1e-7), but one flipped fp16 rounding early in the
    # network moves the features by ~2e-3; both sit at the same distance (~6e-3) from the exact-f32 flavour
    assert np.abs(gram - direct).max() <= 5e-3, np.abs(gram - direct).max()
    assert ((gram * direct).sum(1)).min() >= 0.9998
    if n <= 8:
        ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(crops)).numpy()
        assert (gram * ref).sum(1).min() >= COS_MIN
        assert np.abs(gram - ref).max() <= FEAT_ATOL
    monkeypatch.delenv("BUSCA_REID_GRAM")
    ReIDEncoderHIP(ctx, sd)                                                   # leave the shared ctx in auto mode


def test_reid_halo_conv_path(ctx, monkeypatch):
    """Large batches run the stride-1 3x3 convs of layers 1-3 through the halo-resident kernel (reid_halo.hip.inc; all
    three variants are active from 96 crops).  Same stored roundings as the generic kernel; only the f32 summation order
    inside a conv differs."""
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    n = 198
    crops = _crops(900, n)
    monkeypatch.setenv("BUSCA_REID_HALO", "0")
    generic = ReIDEncoderHIP(ctx, sd).forward(crops).cpu().numpy()
    monkeypatch.setenv("BUSCA_REID_HALO", "1")
    m = ReIDEncoderHIP(ctx, sd)
    halo = m.forward(crops).cpu().numpy()
    assert np.array_equal(halo, m.forward(crops).cpu().numpy())
    assert np.abs(halo - generic).max() <= 5e-3, np.abs(halo - generic).max()
    assert (halo * generic).sum(1).min() >= 0.9998
    monkeypatch.delenv("BUSCA_REID_HALO")
    ReIDEncoderHIP(ctx, sd)


def test_reid_halo_two_by_two_waves_path(ctx, monkeypatch):
    """Layer 1's 3x3 at very large batches: 2 x 2 waves on 256-pixel tiles, two partial-sum rows per tile (conv3x3_halo_kernel
    <2, 32, 8, 2>; automatic from 5120 tiles on, forced here at a batch the suite can afford).  A wave's half tile is exactly one
    128-pixel tile of the one-tile-wide kernel and both accumulate K in the same order, so plain and weighted features must be
    BIT-IDENTICAL to that kernel's."""
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    n = 112
    crops = _crops(901, n)
    wts = (1 + (np.arange(n) % 5)).astype(np.float32)
    monkeypatch.setenv("BUSCA_REID_HALO_WPX", "0")
    m0 = ReIDEncoderHIP(ctx, sd)
    narrow, narrow_w = m0.forward(crops).cpu().numpy(), m0.forward(crops, weights=wts).cpu().numpy()
    monkeypatch.setenv("BUSCA_REID_HALO_WPX", "1")
    monkeypatch.setenv("BUSCA_REID_HALO_WPX_MIN", "1")
    m = ReIDEncoderHIP(ctx, sd)
    wide, wide_w = m.forward(crops).cpu().numpy(), m.forward(crops, weights=wts).cpu().numpy()
    assert np.isfinite(wide).all() and np.array_equal(wide, narrow) and np.array_equal(wide_w, narrow_w)
    monkeypatch.delenv("BUSCA_REID_HALO_WPX")
    monkeypatch.delenv("BUSCA_REID_HALO_WPX_MIN")
    ReIDEncoderHIP(ctx, sd)


@pytest.mark.parametrize("n,nw,pt", [(5, 0, 0), (8, 16, 2), (8, 8, 4), (24, 4, 4), (24, 8, 2), (40, 4, 2)])
def test_reid_kwave_conv_path(ctx, monkeypatch, n, nw, pt):
    """Small / mid batches run most convs through conv_kwave_kernel (K split across the waves of a workgroup,
    reid_kwave.hip.inc).  Every (waves, tile) variant against the LDS-tiled schedule (same stored roundings; f32 summation
    order differs) and, at oracle-sized batches, against the float32 oracle."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = synth.reid_state_dict(3)
    crops = _crops(1500 + n, n)
    monkeypatch.setenv("BUSCA_REID_KWAVE_BLOCKS", "0")
    tiled = ReIDEncoderHIP(ctx, sd).forward(crops).cpu().numpy()
    monkeypatch.setenv("BUSCA_REID_KWAVE_BLOCKS", "100000")    # every eligible conv
    monkeypatch.setenv("BUSCA_REID_KWAVE_HALO", "100000")      # including the 3x3 convs the halo kernel would take
    if nw:
        monkeypatch.setenv("BUSCA_REID_KWAVE_NW", str(nw))
        monkeypatch.setenv("BUSCA_REID_KWAVE_PT", str(pt))
    m = ReIDEncoderHIP(ctx, sd)
    kw = m.forward(crops).cpu().numpy()
    assert np.array_equal(kw, m.forward(crops).cpu().numpy())              # fixed summation order: reproducible
    assert np.abs(kw - tiled).max() <= 5e-3, np.abs(kw - tiled).max()
    assert (kw * tiled).sum(1).min() >= 0.9998
    if n <= 8:
        ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(crops)).numpy()
        assert (kw * ref).sum(1).min() >= COS_MIN
        assert np.abs(kw - ref).max() <= FEAT_ATOL
    for k in ("BUSCA_REID_KWAVE_BLOCKS", "BUSCA_REID_KWAVE_HALO", "BUSCA_REID_KWAVE_NW", "BUSCA_REID_KWAVE_PT"):
        monkeypatch.delenv(k, raising=False)
    ReIDEncoderHIP(ctx, sd)


@pytest.mark.parametrize("n", [7, 24])
def test_reid_pipelined_conv_path(ctx, monkeypatch, n):
    """conv_pipe_kernel (reid_pipe.hip.inc: weights direct into a three-deep register ring, two LDS activation tiles, one barrier per K
    step; automatic for the large launches of layers 3-4) forced onto EVERY eligible raw-output conv at small batches - 1x1 and 3x3,
    stride 1 and 2, 128- and 256-channel tiles, with and without the producer's BatchNorm, ragged last pixel tile: same stored
    roundings as the tiled kernel, different f32 summation order; weighted statistics too."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = synth.reid_state_dict(3)
    crops = _crops(1900 + n, n)
    knobs = {"BUSCA_REID_KWAVE_BLOCKS": "0", "BUSCA_REID_HALO_MIN": "100000"}
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("BUSCA_REID_PIPE_MIN", "0")
    tiled = ReIDEncoderHIP(ctx, sd).forward(crops).cpu().numpy()
    monkeypatch.setenv("BUSCA_REID_PIPE_MIN", "1")
    monkeypatch.setenv("BUSCA_REID_PIPE_ALL", "1")
    m = ReIDEncoderHIP(ctx, sd)
    got = m.forward(crops).cpu().numpy()
    assert np.array_equal(got, m.forward(crops).cpu().numpy())
    assert not np.array_equal(got, tiled)                               # another kernel really ran
    assert np.abs(got - tiled).max() <= 5e-3, np.abs(got - tiled).max()
    assert (got * tiled).sum(1).min() >= 0.9998
    if n <= 8:
        ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(crops)).numpy()
        assert (got * ref).sum(1).min() >= COS_MIN
        assert np.abs(got - ref).max() <= FEAT_ATOL
    w = np.ones(n, np.float32); w[::3] = 4
    idx = np.repeat(np.arange(n), w.astype(int))
    gw = m.forward(crops, weights=w).cpu().numpy()
    ge = m.forward(crops[idx]).cpu().numpy()[np.searchsorted(idx, np.arange(n))]
    assert np.abs(gw - ge).max() <= 5e-3, np.abs(gw - ge).max()
    for k in list(knobs) + ["BUSCA_REID_PIPE_MIN", "BUSCA_REID_PIPE_ALL"]:
        monkeypatch.delenv(k, raising=False)
    ReIDEncoderHIP(ctx, sd)


@pytest.mark.parametrize("n", [1, 13, 33, 65, 129, 193])
def test_reid_default_schedule_vs_plain_tiled_schedule(ctx, monkeypatch, n):
    """Batch sizes on both sides of the launch-time switch-overs (K-split kernel below 288 tiles / 600 MB, half-image halo tiles
    below 192 crops, Gram statistics from 65 536 pixels, fused tails ...): whatever mix of kernels the default picks must agree
    with the plain tiled schedule (tools/reid_schedule_soak.py runs the long list)."""
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    crops = _crops(2100 + n, n)
    plain_env = {"BUSCA_REID_KWAVE_BLOCKS": "0", "BUSCA_REID_HALO_HALF": "0", "BUSCA_REID_GRAM": "0", "BUSCA_REID_FUSE_C1": "0"}
    for k in plain_env:
        monkeypatch.delenv(k, raising=False)
    a = ReIDEncoderHIP(ctx, sd).forward(crops).cpu().numpy()
    for k, v in plain_env.items():
        monkeypatch.setenv(k, v)
    b = ReIDEncoderHIP(ctx, sd).forward(crops).cpu().numpy()
    assert np.isfinite(a).all()
    assert np.abs(a - b).max() <= 5e-3, np.abs(a - b).max()
    assert (a * b).sum(1).min() >= 0.9998
    for k in plain_env:
        monkeypatch.delenv(k, raising=False)
    ReIDEncoderHIP(ctx, sd)


def test_reid_fused_tail_conv1_path(ctx, monkeypatch):
    """Large batches: the block tails of layers 1-3 also run the next bottleneck's conv1 on the tile they hold
    (tail_conv1_kernel; with the Gram schedule forced on, all seven instantiations run at 96 crops).  The stored tensors
    are rounded exactly as in the two-kernel schedule; only statistics summation order differs."""
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    n = 96
    crops = _crops(1200, n)
    monkeypatch.setenv("BUSCA_REID_GRAM", "1")
    monkeypatch.setenv("BUSCA_REID_FUSE_C1", "0")
    plain = ReIDEncoderHIP(ctx, sd).forward(crops).cpu().numpy()
    monkeypatch.setenv("BUSCA_REID_FUSE_C1", "3")             # through layer 3 (the default stops at layer 2)
    m = ReIDEncoderHIP(ctx, sd)
    fused = m.forward(crops).cpu().numpy()
    assert np.array_equal(fused, m.forward(crops).cpu().numpy())
    assert np.abs(fused - plain).max() <= 5e-3, np.abs(fused - plain).max()
    assert (fused * plain).sum(1).min() >= 0.9998
    monkeypatch.delenv("BUSCA_REID_FUSE_C1")
    monkeypatch.delenv("BUSCA_REID_GRAM")
    ReIDEncoderHIP(ctx, sd)


@pytest.mark.parametrize("prec", ["f16", "f32", "x3"])
def test_reid_weighted_statistics_equal_the_expanded_batch(ctx, prec):
    """busca_reid_forward_w: a BatchNorm batch in which crops repeat (the same detection among the candidates of several tracks,
    zero padding) given as distinct crops + multiplicities equals the forward over the expanded batch - up to floating-point
    summation order of the statistics - and stays on the oracle evaluated on the expanded batch."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = synth.reid_state_dict(3)
    uniq = _crops(777, 7)
    uniq[3] = 0                                                   # a zero (padding) crop among them
    counts = np.array([5, 1, 2, 9, 1, 3, 1])
    inverse = np.repeat(np.arange(7), counts)
    rng = np.random.default_rng(0)
    rng.shuffle(inverse)
    expanded = uniq[inverse]                                      # 22 crops, 7 distinct
    m = ReIDEncoderHIP(ctx, sd, precision=prec)
    full = m.forward(expanded).cpu().numpy()
    w = m.forward(uniq, weights=counts).cpu().numpy()[inverse]
    assert np.array_equal(m.forward(uniq, weights=counts).cpu().numpy()[inverse], w)            # deterministic
    if prec in EXACT_FLAVOURS:
        assert np.abs(w - full).max() <= 2e-5, np.abs(w - full).max()
    else:     # one flipped fp16 rounding early in the network moves the features by ~2e-3 (as between the other schedules)
        assert np.abs(w - full).max() <= 5e-3 and (w * full).sum(1).min() >= 0.9998
    ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(expanded)).numpy()
    if prec in EXACT_FLAVOURS:
        assert np.abs(w - ref).max() <= 5e-5
    else:
        assert (w * ref).sum(1).min() >= COS_MIN and np.abs(w - ref).max() <= FEAT_ATOL
    # weights of all ones are the plain forward, bit for bit
    assert np.array_equal(m.forward(uniq, weights=np.ones(7)).cpu().numpy(), m.forward(uniq).cpu().numpy())


def test_reid_weighted_statistics_large_duplication(ctx):
    """MOT20-like candidate batch: 1 024 slots drawn from 48 detections (every layer's tile / crop alignment is exercised:
    48 crops put layer 3 at 9 216 pixels, layer 4 at 2 304) against the expanded batch."""
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    uniq = _crops(4242, 48)
    inverse = np.random.default_rng(1).integers(0, 48, 1024)
    inverse[:48] = np.arange(48)
    counts = np.bincount(inverse, minlength=48)
    m = ReIDEncoderHIP(ctx, sd)
    full = m.forward(uniq[inverse]).cpu().numpy()
    w = m.forward(uniq, weights=counts).cpu().numpy()[inverse]
    assert np.abs(w - full).max() <= 5e-3 and (w * full).sum(1).min() >= 0.9998


def test_reid_schedule_options_change_between_forwards(ctx):
    """The ReID schedule knobs are per-context options of a loaded extractor (busca_set_option "reid_*"), not only BUSCA_REID_* variables
    latched at load time: flipping the halo-resident 3x3 kernel / the fused tails off between two forwards of the SAME handle gives the
    plain-schedule result (within the schedules' summation-order band), flipping them back restores the first result bit for bit."""
    from busca_amd import _lib
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    m = ReIDEncoderHIP(ctx, sd, precision="f16")
    crops = _crops(611, 24)
    a = m.forward(crops).cpu().numpy()
    assert ctx.get_option("reid_halo") == 1 and ctx.get_option("reid_fuse_c1") == 1
    ctx.set_option("reid_halo", 0); ctx.set_option("reid_fuse_c1", 0); ctx.set_option("reid_gram", 0)
    try:
        b = m.forward(crops).cpu().numpy()
    finally:
        ctx.set_option("reid_halo", 1); ctx.set_option("reid_fuse_c1", 1); ctx.set_option("reid_gram", -1)
    c = m.forward(crops).cpu().numpy()
    assert np.array_equal(a, c)
    assert np.abs(a - b).max() <= 5e-3 and (a * b).sum(1).min() >= 0.9998
    with pytest.raises(_lib.BuscaError):
        ctx.set_option("reid_no_such_knob", 1)


def test_reid_weighted_batch_with_many_distinct_crops(ctx):
    """A weighted BatchNorm batch with more distinct crops than the Gram-statistics scratch holds chunk partials for (a weighted pass cuts its
    chunks at crop boundaries: ~one chunk per crop; from ~615 distinct crops the 256-channel downsample input of layer 2 no longer fits): the
    forward must take the direct-statistics schedule instead of failing with BUSCA_EINVAL (round-3 advisor finding), and agree with the
    expanded batch."""
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    uniq = _crops(7001, 700)
    counts = np.where(np.arange(700) % 2 == 0, 2, 1)
    inverse = np.repeat(np.arange(700), counts)
    m = ReIDEncoderHIP(ctx, sd, precision="f16")
    w = m.forward(uniq, weights=counts).cpu().numpy()
    full = m.forward(torch.from_numpy(uniq).cuda()[torch.from_numpy(inverse).cuda()]).cpu().numpy()
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    assert np.abs(w - full[first]).max() <= 5e-3 and (w * full[first]).sum(1).min() >= 0.9998


def test_reid_x3_large_batch_schedules_at_oracle_size(ctx):
    """The two schedules the split-fp16 flavour switches to on large batches - BN3 statistics of layers 1-2 from the Gram matrix of conv3's
    input (x3_gram_kernel) and the block tails of layers 3-4 formed by the NEXT conv1 while it stages (X3_MRG) - forced on at a batch the
    oracle finishes in seconds (their thresholds put to zero through busca_set_option): same bar against the oracle as the default
    schedule (5e-5), plain and with multiplicities, deterministic, and the default schedule is back bit for bit afterwards."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = synth.reid_state_dict(3)
    uniq = _crops(909, 5)
    counts = np.array([3, 1, 2, 1, 4])
    expanded = uniq[np.repeat(np.arange(5), counts)]
    m = ReIDEncoderHIP(ctx, sd, precision="x3")
    base = m.forward(uniq).cpu().numpy()
    gmin, mmin = ctx.get_option("reid_x3_gram_min"), ctx.get_option("reid_x3_merge_in_min")
    assert gmin > 5 * 3072 * 64 and mmin > 5 * 192 * 1024          # i.e. the default schedule above did NOT take them
    ctx.set_option("reid_x3_gram_min", 0); ctx.set_option("reid_x3_merge_in_min", 0)
    try:
        a = m.forward(uniq).cpu().numpy()
        assert np.array_equal(a, m.forward(uniq).cpu().numpy())
        w = m.forward(uniq, weights=counts).cpu().numpy()
    finally:
        ctx.set_option("reid_x3_gram_min", gmin); ctx.set_option("reid_x3_merge_in_min", mmin)
    assert np.array_equal(base, m.forward(uniq).cpu().numpy())
    # ... and layer 1's fused tails with the next bottleneck's conv1 inside (X3_MERGE_C1, on by default) against the two-kernel schedule: the conv's raw
    # output is the same to the bit, its BatchNorm statistics are summed in another order
    assert ctx.get_option("reid_x3_fuse_c1") == 1
    ctx.set_option("reid_x3_fuse_c1", 0)
    try:
        plain = m.forward(uniq).cpu().numpy()
        plain_w = m.forward(uniq, weights=counts).cpu().numpy()
    finally:
        ctx.set_option("reid_x3_fuse_c1", 1)
    assert np.abs(plain - base).max() <= 2e-5 and np.abs(plain_w - m.forward(uniq, weights=counts).cpu().numpy()).max() <= 2e-5
    # ... and the stride-1 3x3 convs of layers 1-2 staged once per kernel row (ROW3, on by default; 2 = layers 3-4 too on 64-pixel tiles) against the
    # tap-by-tap schedule: the same products, summed chunk-major instead of tap-major where a conv has more than 64 input channels
    assert ctx.get_option("reid_x3_row3") == 1
    for mode in (0, 2):
        ctx.set_option("reid_x3_row3", mode)
        try:
            alt = m.forward(uniq).cpu().numpy()
            alt_w = m.forward(uniq, weights=counts).cpu().numpy()
        finally:
            ctx.set_option("reid_x3_row3", 1)
        assert np.abs(alt - base).max() <= 2e-5 and np.abs(alt_w - plain_w).max() <= 2e-5, (mode, np.abs(alt - base).max())
    # ... and the stem with its input rows staged once per tile as an LDS halo (STEMH, default) against the tap-by-tap stem: the same products in the same order
    assert ctx.get_option("reid_x3_stem_halo") == 1
    ctx.set_option("reid_x3_stem_halo", 0)
    try:
        alt = m.forward(uniq).cpu().numpy()
    finally:
        ctx.set_option("reid_x3_stem_halo", 1)
    assert np.abs(alt - base).max() <= 2e-5, np.abs(alt - base).max()
    # ... and that stem filled straight from the u8 crops through the byte table (default) against the same stem fed from the normalised float copy:
    # the table holds exactly the preprocess kernel's values, so the features are bit-identical - with a padding crop (zero_norm) in the batch too
    assert ctx.get_option("reid_x3_stem_u8") == 1
    zn = torch.tensor([0, 0, 1, 0, 0], dtype=torch.uint8, device="cuda")
    with_bytes = m.forward(uniq, zero_norm=zn).cpu().numpy()
    ctx.set_option("reid_x3_stem_u8", 0)
    try:
        assert np.array_equal(with_bytes, m.forward(uniq, zero_norm=zn).cpu().numpy())
        assert np.array_equal(base, m.forward(uniq).cpu().numpy())
    finally:
        ctx.set_option("reid_x3_stem_u8", 1)
    # ... and the stem that writes the 3x3 / stride-2 max pool of its raw output (times sign(gamma)) in two parts, with layer 1's first conv1 and downsample
    # conv staging relu(bn(max(P, Q above))) (X3_POOL / X3_POOLIN, default), against raw map + pooling pass: the pool commutes with the monotone
    # BatchNorm + ReLU, so the features are the same to the bit
    assert ctx.get_option("reid_x3_stem_pool") == 1
    ctx.set_option("reid_x3_stem_pool", 0)
    try:
        assert np.array_equal(base, m.forward(uniq).cpu().numpy())
        assert np.array_equal(with_bytes, m.forward(uniq, zero_norm=zn).cpu().numpy())
    finally:
        ctx.set_option("reid_x3_stem_pool", 1)
    ref = oreid.reid_forward(sd, oreid.crops_to_reid_input(uniq)).numpy()
    assert np.abs(plain - ref).max() <= 5e-5
    assert np.abs(a - ref).max() <= 5e-5 and np.abs(a - base).max() <= 2e-5, (np.abs(a - ref).max(), np.abs(a - base).max())
    refw = oreid.reid_forward(sd, oreid.crops_to_reid_input(expanded)).numpy()
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    assert np.abs(w - refw[first]).max() <= 5e-5, np.abs(w - refw[first]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5, 44])
def test_reid_x3_persistent_tails_are_bit_identical(ctx, n):
    """Round 5: the fused block tails of layers 1-2 as PERSISTENT workgroups (x3_ptail_kernel, reid_x3p.hip.inc: weights in registers, the next
    half tile's rows prefetched under the current one's products) against the one-shot kernels (option reid_x3_ptail = 0): the same arithmetic
    in the same order, so the features are equal to the bit - plain and with multiplicities (resnet.py:116-128, network.py:553-556)."""
    from busca_amd.reid import ReIDEncoderHIP
    sd = synth.reid_state_dict(3)
    crops = _crops(4242, n)
    counts = 1 + (np.arange(n) * 7) % 4
    m = ReIDEncoderHIP(ctx, sd, precision="x3")
    dflt = ctx.get_option("reid_x3_ptail")
    assert dflt > 0
    try:
        ctx.set_option("reid_x3_ptail", 0)
        one_shot = m.forward(crops).cpu().numpy()
        one_shot_w = m.forward(crops, weights=counts).cpu().numpy()
        ctx.set_option("reid_x3_ptail", 1)               # every tail of layers 1-2 on persistent workgroups, whatever the batch size
        pers = m.forward(crops).cpu().numpy()
        assert np.array_equal(pers, m.forward(crops).cpu().numpy())
        pers_w = m.forward(crops, weights=counts).cpu().numpy()
    finally:
        ctx.set_option("reid_x3_ptail", dflt)
    assert np.array_equal(one_shot, pers)
    assert np.array_equal(one_shot_w, pers_w)
    assert np.array_equal(m.forward(crops).cpu().numpy(), pers)          # the default schedule (persistent from 512 work items up) too


@pytest.mark.parametrize("n,where", [(6, "layer1.0.bn1"), (44, "layer2.1.bn2"), (6, "layer3.2.bn3"), (130, "layer1.1.bn2")])
def test_x3_reid_reports_operands_beyond_its_range(ctx, n, where):
    """The split-fp16 flavour stages activations as fp16 hi + lo of 64 x and does not clamp: an activation beyond |x| = 1023.5 turns its conv's output and BatchNorm
    statistics non-finite, and the pass raises `reid_status` 2 in host-mapped memory (the end-of-pass scan of the (scale, shift) table) instead of returning
    features computed from clipped values (round-5 finding).  BatchNorm affines x 4000 in a first / middle / last conv of a bottleneck, small and large batch
    schedules: the x3 pass reports, a healthy checkpoint never does, and the exact-f32 extractor takes the same checkpoint (<= 1e-4 from the oracle)."""
    from busca_amd.reid import ReIDEncoderHIP
    from oracle import reid as oreid
    sd = synth.reid_state_dict(3)
    crops = _crops(900 + n, n)
    ok = ReIDEncoderHIP(ctx, sd, precision="x3")
    f_ok = ok.forward(crops)
    torch.cuda.synchronize()
    assert ok.take_status() is False and np.isfinite(f_ok.cpu().numpy()).all()
    hot = dict(sd)
    hot[where + ".weight"] = sd[where + ".weight"] * 4000.0           # relu(bn(.)) of ~ 4000 x a few sigma
    hot[where + ".bias"] = sd[where + ".bias"] * 4000.0
    m3 = ReIDEncoderHIP(ctx, hot, precision="x3")
    m3.forward(crops)
    torch.cuda.synchronize()
    assert ctx.get_option("reid_status") == 2
    assert m3.take_status() is True and ctx.get_option("reid_status") == 0 and m3.take_status() is False
    f32 = ReIDEncoderHIP(ctx, hot, precision="f32").forward(crops).cpu().numpy()
    assert np.isfinite(f32).all()
    if n <= 8:
        want = oreid.reid_forward(hot, oreid.crops_to_reid_input(crops)).numpy()
        assert np.abs(f32 - want).max() <= 1e-4
    again = ReIDEncoderHIP(ctx, sd, precision="x3").forward(crops)      # a healthy model on the same context afterwards: clean status, same bits as before
    torch.cuda.synchronize()
    assert ctx.get_option("reid_status") == 0 and np.array_equal(again.cpu().numpy(), f_ok.cpu().numpy())
