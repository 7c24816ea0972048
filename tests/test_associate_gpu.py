"""GPU: the drop-in `BUSCA` class (busca_amd.network, same surface as the reference's busca.network) on fake
tracker objects, against the committed outputs of the reference's own associate_embeddings."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from busca_amd import synth

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


def _args(d=64, ff=128, precision="f32", pinned=True):
    return types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=d, ff_size=ff, activation="gelu",
                                 dropout_p=0.1, input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN",
                                 encode_separator_as_reference=True, encode_special_tokens=False, reid_weights_file="no",
                                 device=torch.device("cuda:0"), precision=precision, pinned_numpy_semantics=pinned)


@pytest.fixture(scope="module")
def model():
    from busca_amd.network import BUSCA
    m = BUSCA(_args()).to(torch.device("cuda:0")).eval()
    sd = dict(synth.dt_state_dict(17, d=64, ff=128))
    sd.update({"reid_encoder.model." + k: v for k, v in synth.reid_state_dict(17).items()})
    m.load_state_dict(sd)
    return m


def _case(ci):
    import make_golden as mg
    name, hist, n_det, kal, P = mg.ASSOC_CASES[ci]
    tracks, dets, kals = mg.assoc_scene(17 + ci, hist, n_det, kal)
    return name, tracks, dets, kals, P


# ReID runs with fp16 operands: the probabilities inherit its ~1e-2 feature tolerance
PROB_ATOL = 3e-2


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
@pytest.mark.parametrize("mode", ["f64", "f32"])
def test_associate_vs_reference(model, golden_dir, ci, mode):
    g = np.load(os.path.join(golden_dir, "assoc.npz"))
    name, tracks, dets, kals, P = _case(ci)
    model.pinned_numpy = (mode == "f64")
    model._dirty = True
    dists = g[name + "_dists"]
    pm, rel = model.associate_embeddings(tracks_embeddings=tracks, dets_embeddings=dets, dists_matrix=dists, seq_len=11,
                                         num_candidates=P, use_broader_memory=True, select_highest_candidate=False,
                                         extra_kalman_candidates=kals, normalize_ims=True)
    ref = g["%s_probs_%s_sel0" % (name, mode)]
    assert pm.shape == ref.shape and pm.dtype == np.float64
    assert np.array_equal(rel, g[name + "_reliable"])
    assert np.array_equal(pm == 0, ref == 0)                      # same scatter pattern (columns, padding, Kalman slot)
    assert np.abs(pm - ref).max() <= PROB_ATOL, np.abs(pm - ref).max()
    # one-hot mode: identical decision wherever the reference's winner is clear of the runner-up
    pm1, _ = model.associate_embeddings(tracks, dets, dists, 11, P, True, True, extra_kalman_candidates=kals, normalize_ims=True)
    ref1 = g["%s_probs_%s_sel1" % (name, mode)]
    assert set(np.unique(pm1)) <= {0.0, 1.0}
    full = model._last["probs"].cpu().numpy()
    srt = np.sort(full, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 2 * PROB_ATOL
    assert np.array_equal(pm1[clear], ref1[clear])


def test_early_returns_and_protocol(model):
    assert model.associate_embeddings([], [1], None, 11, 5, True, True) == (None, None)
    assert model.associate_embeddings([1], [], None, 11, 5, True, True, extra_kalman_candidates=[]) == (None, None)
    assert model.expected_image_size == (384, 128)
    assert model.num_params > 24_000_000
    assert model.get_image_crops(np.zeros((64, 64, 3), np.uint8), []).shape == (0, 128, 384, 3)
    crops = model.get_image_crops(synth.randint_u8(1, "f", (200, 300, 3)), [[10, 10, 60, 150], [-5, -5, 40, 90]], normalize=False)
    assert crops.shape == (2, 384, 128, 3) and crops.dtype == np.uint8
    n = model.get_image_crops(synth.randint_u8(1, "f", (200, 300, 3)), [[10, 10, 60, 150]], normalize=True)
    assert n.dtype == np.float32


def test_forward_accepts_reference_float_layout(model):
    """BUSCA.forward with the reference's normalised float RGB CHW input == the u8 BGR HWC input."""
    import make_golden as mg
    from busca_amd import tracking
    model.pinned_numpy = True
    model._dirty = True
    B, L, P = 2, 11, 5
    mem = mg.smooth_crops(5, B * L).reshape(B, L, 384, 128, 3)
    can = mg.smooth_crops(6, B * P).reshape(B, P, 384, 128, 3)
    inp = synth.dt_inputs(5, B, L, P, sentinel_every=0)
    a = model.forward(mem, can, inp["mem_boxes"], inp["can_boxes"]).cpu().numpy()
    memf = torch.from_numpy(tracking.normalize_crops(mem)).float()[..., [2, 1, 0]].permute(0, 1, 4, 2, 3)
    canf = torch.from_numpy(tracking.normalize_crops(can)).float()[..., [2, 1, 0]].permute(0, 1, 4, 2, 3)
    b = model.forward(memf, canf, inp["mem_boxes"], inp["can_boxes"], return_logits=True, return_att=True).cpu().numpy()
    assert np.array_equal(a, b)
    assert model.logits.shape == (B, P + 2, 64) and model.mem_logits.shape == (B, 64) and len(model.attentions) == 4


@pytest.mark.parametrize("host_copy", ["lazy", "eager"])
def test_device_resident_track_memory(model, host_copy):
    """Crops returned by get_image_crops keep a device twin; association then gathers them on the GPU
    (no H2D) and gives bit-identical results to the host path.  "lazy" (the default): the host bytes are fetched by the first host
    read; "eager": a real ndarray, as in rounds 2-3."""
    from busca_amd.tracking import DeviceBackedCrops, DeviceCrops
    import make_golden as mg
    model.pinned_numpy = True
    model._dirty = True
    assert model.crop_host_copy == "lazy"                              # the default
    frame = synth.randint_u8(4, "frame", (540, 960, 3))
    boxes = np.array([[50 + 30 * i, 40 + 5 * i, 110 + 30 * i, 260 + 5 * i] for i in range(24)], np.float32)
    model.crop_host_copy = host_copy
    try:
        crops = model.get_image_crops(frame, boxes, normalize=False)
        again = model.get_image_crops(frame, boxes[:3], normalize=False)
    finally:
        model.crop_host_copy = "lazy"
    assert np.array_equal(np.asarray(again[2]), np.asarray(crops[2]))
    assert isinstance(crops, DeviceCrops if host_copy == "lazy" else DeviceBackedCrops) and crops[3].slot is not None and crops[3].dev is not None
    if host_copy == "lazy":
        assert crops[4].slot.host is None and crops[4].slot.host_src is not None     # nobody has read (or waited for) the host bytes yet
    assert np.array_equal(crops[3].dev.cpu().numpy(), np.asarray(crops[3]))
    assert type(np.array(crops[3])) is np.ndarray and getattr(np.array(crops[3]), "slot", None) is None
    hist = [mg.FakeTrack([[50, 40, 60, 220]] * 12, [crops[i] for i in range(12)]), mg.FakeTrack([[300, 80, 60, 220]] * 11, [crops[i] for i in range(12, 23)])]
    dets = [mg.FakeTrack([[55, 45, 60, 220]], [crops[23]]), mg.FakeTrack([[310, 85, 60, 220]], [crops[5]])]
    kal = [mg.FakeTrack([[52, 42, 60, 220]], [crops[1]]), mg.FakeTrack([[305, 82, 60, 220]], [crops[14]])]
    dists = np.array([[7.0, 250.0], [250.0, 11.0]])
    a, ra = model.associate_embeddings(hist, dets, dists, 11, 5, True, False, extra_kalman_candidates=kal, normalize_ims=True)
    assert model.last_gather[1] == 0 and model.last_gather[0] > 0           # everything came from the device pool
    # same scene with plain host copies of the crops -> host path
    plain = lambda trk: mg.FakeTrack(trk.tlwh_mem, [np.array(c) for c in trk.images_mem], trk.scale)
    b, rb = model.associate_embeddings([plain(t) for t in hist], [plain(t) for t in dets], dists, 11, 5, True, False,
                                       extra_kalman_candidates=[plain(t) for t in kal], normalize_ims=True)
    assert model.last_gather[0] == 0 and model.last_gather[1] > 0
    assert np.array_equal(a, b) and np.array_equal(ra, rb)


@pytest.mark.parametrize("reid_prec", ["f32", "x3", "x3+x3"])
@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_associate_exact_flavours_vs_reference(golden_dir, ci, reid_prec):
    """f32 Decision Transformer + a float32-class ReID ("f32": exact f32 MFMA; "x3": split-fp16 products, BUSCA_PREC_F16X3; "x3+x3": the library's
    default, split-fp16 products in the Decision Transformer too): the whole associate_embeddings output agrees with the reference's to float32
    round-off, and the one-hot decisions are identical."""
    from busca_amd.network import BUSCA
    a = _args(precision="x3" if reid_prec == "x3+x3" else "f32")
    reid_prec = reid_prec.split("+")[0]
    a.reid_precision = reid_prec
    m = BUSCA(a).to(torch.device("cuda:0")).eval()
    sd = dict(synth.dt_state_dict(17, d=64, ff=128))
    sd.update({"reid_encoder.model." + k: v for k, v in synth.reid_state_dict(17).items()})
    m.load_state_dict(sd)
    g = np.load(os.path.join(golden_dir, "assoc.npz"))
    name, tracks, dets, kals, P = _case(ci)
    dists = g[name + "_dists"]
    pm, rel = m.associate_embeddings(tracks, dets, dists, 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
    ref = g["%s_probs_f64_sel0" % name]
    assert np.array_equal(rel, g[name + "_reliable"])
    assert np.abs(pm - ref).max() <= 2e-4, np.abs(pm - ref).max()
    pm1, _ = m.associate_embeddings(tracks, dets, dists, 11, P, True, True, extra_kalman_candidates=kals, normalize_ims=True)
    assert np.array_equal(pm1, g["%s_probs_f64_sel1" % name])


@pytest.mark.parametrize("flavour,sep_ref", [("MEM-CAN-SEP", False), ("MEM-SEP-CAN", True), ("MEM-CAN-SEP-BAD", True)])
def test_associate_other_input_flavours_vs_oracle(golden_dir, flavour, sep_ref):
    """associate_embeddings with a non-shipped token layout (P+1 probability columns without the BAD token): exact flavours against
    the oracle chain (oracle ReID + oracle DT of that flavour - pinned to the reference by tests/golden/flavours_dt.npz - + oracle
    host logic) on the first golden scene."""
    from busca_amd.network import BUSCA
    from oracle import associate as oa, dt as odt, reid as oreid
    a = _args(precision="f32")
    a.reid_precision = "f32"
    a.input_flavour, a.encode_separator_as_reference = flavour, sep_ref
    m = BUSCA(a).to(torch.device("cuda:0")).eval()
    sd_dt, sd_reid = synth.dt_state_dict(23, d=64, ff=128, flavour=flavour), synth.reid_state_dict(17)
    sd = dict(sd_dt)
    sd.update({"reid_encoder.model." + k: v for k, v in sd_reid.items()})
    m.load_state_dict(sd)
    g = np.load(os.path.join(golden_dir, "assoc.npz"))
    name, tracks, dets, kals, P = _case(0)
    dists = g[name + "_dists"]
    cfg = odt.DTConfig(d=64, ff=128, fake_f64=True, flavour=flavour, encode_sep_as_ref=sep_ref)

    def step(mem_u8, can_u8, mem_ltrb, can_ltrb):
        B, L = mem_u8.shape[:2]
        P_ = can_u8.shape[1]
        mf = oreid.reid_forward(sd_reid, oreid.crops_to_reid_input(mem_u8.reshape(B * L, 384, 128, 3))).view(B, L, -1)
        cf = oreid.reid_forward(sd_reid, oreid.crops_to_reid_input(can_u8.reshape(B * P_, 384, 128, 3))).view(B, P_, -1)
        return torch.softmax(odt.dt_forward(sd_dt, cfg, mf, cf, mem_ltrb, can_ltrb), -1).numpy()

    for sel in (False, True):
        want, wrel = oa.associate_embeddings(step, tracks, dets, dists, 11, P, True, sel, extra_kalman_candidates=kals)
        got, rel = m.associate_embeddings(tracks, dets, dists, 11, P, True, sel, extra_kalman_candidates=kals, normalize_ims=True)
        assert got.shape == want.shape and np.array_equal(rel, wrel)
        assert m._last["probs"].shape[1] == P + (2 if "BAD" in flavour else 1)
        if sel:
            assert np.array_equal(got, want)
        else:
            assert np.abs(got - want).max() <= 2e-4, np.abs(got - want).max()


def test_device_only_crops(model):
    """device_only_crops: get_image_crops skips the device->host copy; the crops associate through their pool slots
    (bit-identical), and ANY host read of such a crop (np.array, arithmetic, astype, pickle) returns the real pixels -
    there are no placeholder bytes that could reach a BatchNorm batch."""
    import pickle
    import make_golden as mg
    from busca_amd.tracking import DeviceCrops
    model.pinned_numpy = True
    model._dirty = True
    frame = synth.randint_u8(4, "frame", (540, 960, 3))
    boxes = np.array([[50 + 30 * i, 40 + 5 * i, 110 + 30 * i, 260 + 5 * i] for i in range(24)], np.float32)

    def scene(crops):
        hist = [mg.FakeTrack([[50, 40, 60, 220]] * 12, [crops[i] for i in range(12)]), mg.FakeTrack([[300, 80, 60, 220]] * 11, [crops[i] for i in range(12, 23)])]
        dets = [mg.FakeTrack([[55, 45, 60, 220]], [crops[23]]), mg.FakeTrack([[310, 85, 60, 220]], [crops[5]])]
        kal = [mg.FakeTrack([[52, 42, 60, 220]], [crops[1]]), mg.FakeTrack([[305, 82, 60, 220]], [crops[14]])]
        return hist, dets, kal
    dists = np.array([[7.0, 250.0], [250.0, 11.0]])
    ref_crops = model.get_image_crops(frame, boxes, normalize=False)
    h, d, k = scene(ref_crops)
    a, ra = model.associate_embeddings(h, d, dists, 11, 5, True, False, extra_kalman_candidates=k, normalize_ims=True)
    model.device_only_crops = True
    try:
        crops = model.get_image_crops(frame, boxes, normalize=False)
        assert isinstance(crops, DeviceCrops) and crops.shape == (24, 384, 128, 3) and len(crops) == 24 and crops[2].dev is not None
        assert crops[2].slot.host is None                                     # nothing was copied back
        h, d, k = scene(crops)
        b, rb = model.associate_embeddings(h, d, dists, 11, 5, True, False, extra_kalman_candidates=k, normalize_ims=True)
        assert model.last_gather[1] == 0
        assert np.array_equal(a, b) and np.array_equal(ra, rb)
        # host reads give the real pixels, whatever the route
        truth = np.asarray(ref_crops[7])
        assert np.array_equal(np.array(crops[7]), truth) and type(np.array(crops[7])) is np.ndarray
        assert np.array_equal(np.ascontiguousarray(crops[8]), np.asarray(ref_crops[8]))
        assert np.array_equal(np.stack([crops[9], crops[10]]), np.asarray(ref_crops[9:11]))
        assert np.array_equal(crops[11].astype(np.float32), np.asarray(ref_crops[11]).astype(np.float32))
        assert np.allclose(crops[12] / 255.0, np.asarray(ref_crops[12]) / 255.0)
        assert np.array_equal(pickle.loads(pickle.dumps(crops[13])), np.asarray(ref_crops[13]))
        assert np.array_equal(np.asarray(crops)[3], np.asarray(ref_crops[3]))
        # a copy has no slot: it takes the host path with the same pixels -> same result
        d[0].images_mem = [np.array(d[0].images_mem[0])]
        c2, _ = model.associate_embeddings(h, d, dists, 11, 5, True, False, extra_kalman_candidates=k, normalize_ims=True)
        assert model.last_gather[1] == 2 and np.array_equal(c2, a)       # that detection is a candidate of both tracks
    finally:
        model.device_only_crops = False


def test_lazy_host_bytes_are_fetched_on_demand_once_per_call(model):
    """Lazy mode (round 5): get_image_crops copies NOTHING to the host; the first host read of any crop of a call fetches that call's whole batch
    with one gather + one device->host copy (tracking.FrameHostCopy), later reads are views of it; crops of calls nobody reads never cost a
    transfer, however long the sequence; a crop read long after its call still hands out its own pixels (busca/network.py:492-507 returns host
    arrays: same bytes, later)."""
    from busca_amd import tracking
    model.pinned_numpy = True
    frame = synth.randint_u8(9, "frame", (540, 960, 3))
    boxes = np.array([[40 + 25 * i, 30 + 4 * i, 100 + 25 * i, 250 + 4 * i] for i in range(12)], np.float32)
    eager = np.asarray(tracking.get_image_crops(frame, boxes, normalize=False, ctx=model._ctx, host_copy="eager"))
    first = model.get_image_crops(frame, boxes, normalize=False)
    src = first[5].slot.host_src[0]
    assert first[5].slot.host is None and src._np is None            # nothing fetched yet
    keep = [first]
    for k in range(11):                                               # later calls: their crops are never read
        keep.append(model.get_image_crops(np.roll(frame, 3 * (k + 1), axis=1), boxes, normalize=False))
    assert all(c[0].slot.host_src[0]._np is None for c in keep)
    early_view = np.asarray(first[3])                                 # first host read: the whole batch of that call arrives
    assert src._np is not None and src._np.shape == (12, 384, 128, 3) and keep[4][0].slot.host_src[0]._np is None
    assert np.array_equal(early_view, eager[3])
    assert np.array_equal(np.asarray(first[5]), eager[5]) and np.asarray(first[5]).base is src._np          # a view of the same fetch
    assert np.array_equal(np.asarray(keep[-1][2]), np.asarray(tracking.get_image_crops(np.roll(frame, 3 * len(keep[1:]), axis=1), boxes, normalize=False,
                                                                                      ctx=model._ctx, host_copy="eager"))[2])


def test_two_models_on_one_gpu_do_not_share_weights():
    """Two BUSCA objects (different shapes and weights) used alternately give what each gives alone; the same for two
    DecisionTransformerHIP / ReIDEncoderHIP handles that share one context."""
    import make_golden as mg
    from busca_amd import _lib
    from busca_amd.dt import DecisionTransformerHIP
    from busca_amd.reid import ReIDEncoderHIP
    m1, m2 = _model(64, 128, 17), _model(256, 512, 5)
    name, tracks, dets, kals, P = _case(0)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "assoc.npz"))
    dists = g[name + "_dists"]
    run = lambda m: m.associate_embeddings(tracks, dets, dists, 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)[0]
    a1 = run(m1); a2 = run(m2); b1 = run(m1); b2 = run(m2)
    assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and not np.array_equal(a1, a2)
    assert m1._ctx is not m2._ctx
    ctx = _lib.Context(0)
    inp = synth.dt_inputs(3, 4, 11, 5)
    f = lambda m: m.forward(inp["mem_feat"], inp["can_feat"], inp["mem_boxes"], inp["can_boxes"])["logits"].cpu().numpy()
    d1 = DecisionTransformerHIP(ctx, synth.dt_state_dict(1, d=64, ff=128), precision="f32")
    x1 = f(d1)
    d2 = DecisionTransformerHIP(ctx, synth.dt_state_dict(2, d=256, ff=512), precision="f16")
    x2 = f(d2)
    assert np.array_equal(f(d1), x1) and np.array_equal(f(d2), x2) and np.array_equal(f(d1), x1)
    crops = mg.smooth_crops(9, 4)
    r1 = ReIDEncoderHIP(ctx, synth.reid_state_dict(1)); y1 = r1.forward(crops).cpu().numpy()
    r2 = ReIDEncoderHIP(ctx, synth.reid_state_dict(2), precision="f32"); y2 = r2.forward(crops).cpu().numpy()
    assert np.array_equal(r1.forward(crops).cpu().numpy(), y1) and np.array_equal(r2.forward(crops).cpu().numpy(), y2)
    ctx.close()


def test_crop_pool_is_bounded_over_a_long_sequence(model):
    """2 000 simulated frames: tracks are born, grow their `images_mem`, die.  HBM held by the crop pool stays flat
    (slots return when the last reference to a crop dies; beyond the budget the oldest crops spill to the host), and an
    association that mixes resident and spilled crops equals the all-host path bit for bit."""
    import make_golden as mg
    from busca_amd import geometry
    from busca_amd.crop_pool import CropPool, CROP_BYTES
    model.pinned_numpy = True
    model._dirty = True
    model._sync()
    ctx = model._ctx
    ctx._crop_pool = CropPool(ctx.device, budget_bytes=256 * CROP_BYTES, slab_crops=64)         # small budget: 256 crops
    pool = geometry.crop_pool(ctx)
    frame = synth.randint_u8(4, "frame", (540, 960, 3))
    rng = np.random.default_rng(0)
    tracks, sizes = [], []
    torch.cuda.synchronize()
    for f in range(2000):
        n = 12
        x = rng.uniform(0, 800, n); y = rng.uniform(0, 300, n)
        boxes = np.stack([x, y, x + rng.uniform(30, 120, n), y + rng.uniform(80, 220, n)], 1)
        crops = model.get_image_crops(frame, boxes, normalize=False)
        if f % 7 == 0 or not tracks:
            tracks.append(mg.FakeTrack([boxes[0][:2].tolist() + [60, 200]], [crops[0]]))
        for t_i, trk in enumerate(tracks[:n - 1]):
            trk.images_mem.append(crops[1 + t_i])
            trk.tlwh_mem.append(trk.tlwh_mem[-1])
        if len(tracks) > 8:                                  # the oldest track dies: its crops leave the pool
            tracks.pop(0)
        sizes.append(pool.device_bytes)
    assert max(sizes) <= 256 * CROP_BYTES and sizes[-1] == sizes[len(sizes) // 2]        # flat
    assert pool.n_live <= 256
    assert pool.spilled > 0                                  # the budget was exceeded and handled
    long_tracks = [t for t in tracks if len(t.images_mem) >= 11]
    assert long_tracks
    dets = [mg.FakeTrack([[55, 45, 60, 220]], [long_tracks[0].images_mem[-1]])]
    dists = np.zeros((len(long_tracks), 1))
    a, ra = model.associate_embeddings(long_tracks, dets, dists, 11, 5, True, False, extra_kalman_candidates=long_tracks, normalize_ims=True)
    mixed = model.last_gather
    memo = {}                                 # one host copy per distinct crop object: repeated crops stay repeated objects,
    def host(c):                              # which is what lets both routes give the extractor the same distinct-crop batch
        if id(c) not in memo:
            memo[id(c)] = np.array(c)
        return memo[id(c)]
    plain = lambda trk: mg.FakeTrack(trk.tlwh_mem, [host(c) for c in trk.images_mem], trk.scale)
    pl = [plain(t) for t in long_tracks]
    b, rb = model.associate_embeddings(pl, [plain(dets[0])], dists, 11, 5, True, False, extra_kalman_candidates=pl, normalize_ims=True)
    assert model.last_gather[0] == 0 and mixed[0] > 0
    assert np.array_equal(a, b) and np.array_equal(ra, rb)
    ctx._crop_pool = None


def test_associate_prenormalised_inputs_vs_reference(golden_dir):
    """normalize_ims=False (the function's default; no shipped adapter uses it): the caller's crops are already normalised
    float32 and the reference's zero crops are 0.0 AFTER normalisation (network.py:285,306,354)."""
    import make_golden as mg
    m = _model(64, 128, 17, "f32", "f32")
    g = np.load(os.path.join(golden_dir, "assoc_nonorm.npz"))
    tracks, dets, kals = mg.assoc_nonorm_scene()
    pm, rel = m.associate_embeddings(tracks, dets, g["dists"], 11, 5, True, False, extra_kalman_candidates=kals, normalize_ims=False)
    assert np.array_equal(rel, g["reliable"])
    assert np.array_equal(pm == 0, g["probs"] == 0)
    assert np.abs(pm - g["probs"]).max() <= 2e-4, np.abs(pm - g["probs"]).max()
    # and the u8 route with the reference's u8-zero padding differs (the zero crops sit elsewhere in the BN batch)
    u8 = lambda trk: mg.FakeTrack(trk.tlwh_mem, mg.denormalise(trk.images_mem), trk.scale)
    pm2, _ = m.associate_embeddings([u8(t) for t in tracks], [u8(t) for t in dets], g["dists"], 11, 5, True, False,
                                    extra_kalman_candidates=[u8(t) for t in kals], normalize_ims=True)
    assert np.abs(pm2 - g["probs"]).max() > 1e-3


def _model(d, ff, seed, precision="f32", reid_precision="f16"):
    from busca_amd.network import BUSCA
    a = _args(d=d, ff=ff, precision=precision)
    a.reid_precision = reid_precision
    m = BUSCA(a).to(torch.device("cuda:0")).eval()
    sd = dict(synth.dt_state_dict(seed, d=d, ff=ff))
    sd.update({"reid_encoder.model." + k: v for k, v in synth.reid_state_dict(seed).items()})
    m.load_state_dict(sd)
    return m


def test_associate_selection_thresholds_vs_reference(golden_dir):
    """highest_candidate_minimum_thresh / keep_highest_value (network.py:415-422; passed by StrongSORT tracker.py:332-333 and
    GHOST tracker.py:757-758) against the reference's outputs (tests/golden/assoc_select.npz), exact flavours."""
    import make_golden as mg
    m = _model(64, 128, 17, "f32", "f32")
    g = np.load(os.path.join(golden_dir, "assoc_select.npz"))
    for ci in (0, 1):
        name, tracks, dets, kals, P = _case(ci)
        for si, (th, keep) in enumerate(mg.ASSOC_SELECT_CASES):
            pm, _ = m.associate_embeddings(tracks, dets, g[name + "_dists"], 11, P, True, True, highest_candidate_minimum_thresh=th,
                                           keep_highest_value=keep, extra_kalman_candidates=kals, normalize_ims=True)
            ref = g["%s_sel%d" % (name, si)]
            assert np.array_equal(pm == 0, ref == 0), (name, si)          # same winners pass / fail the threshold
            assert np.abs(pm - ref).max() <= 2e-4, (name, si, np.abs(pm - ref).max())


@pytest.mark.parametrize("flavour", ["exact", "x3", "default", "fast"])
def test_associate_shipped_shape_vs_reference(golden_dir, flavour):
    """cfgR - the shape every shipped config runs (d=512, ff=1024, L=11, P=5, Kalman candidates, broader memory) - end to end
    against the reference's associate_embeddings (tests/golden/assoc512.npz): exact flavours to float32 round-off with
    identical decisions; default fast flavours (f16 DT operands, fp16 ReID) within the fp16 tolerance."""
    import make_golden as mg
    g = np.load(os.path.join(golden_dir, "assoc512.npz"))
    name, hist, n_det, kal, P = mg.ASSOC512_CASE
    tracks, dets, kals = mg.assoc_scene(23, hist, n_det, kal)
    # "x3": float32 DT + the float32-equivalent split-fp16 ReID; "default" (round 5): split-fp16 in the Decision Transformer too - both held to the exact flavour's bar
    m = {"exact": lambda: _model(512, 1024, 23, "f32", "f32"), "x3": lambda: _model(512, 1024, 23, "f32", "x3"), "default": lambda: _model(512, 1024, 23, "x3", "x3"),
         "fast": lambda: _model(512, 1024, 23, "f16", "f16")}[flavour]()
    if flavour == "default":
        from busca_amd.network import BUSCA
        import types
        assert m.precision == "x3" and m.reid_precision == "x3"
        dflt = BUSCA(types.SimpleNamespace(**{k: v for k, v in vars(_args(512, 1024)).items() if k != "precision"}))
        assert dflt.precision == "x3" and dflt.reid_precision == "x3"         # what a caller who sets nothing gets
    tol = 6e-2 if flavour == "fast" else 1e-3          # d=512 amplifies feature round-off ~10x (see test_oracle_golden.py)
    for mode in ("f64", "f32"):
        m.pinned_numpy = (mode == "f64")
        m._dirty = True
        pm, rel = m.associate_embeddings(tracks, dets, g[name + "_dists"], 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
        ref = g["%s_probs_%s_sel0" % (name, mode)]
        assert np.array_equal(rel, g[name + "_reliable"])
        assert np.array_equal(pm == 0, ref == 0)
        assert np.abs(pm - ref).max() <= tol, np.abs(pm - ref).max()
        pm1, _ = m.associate_embeddings(tracks, dets, g[name + "_dists"], 11, P, True, True, extra_kalman_candidates=kals, normalize_ims=True)
        ref1 = g["%s_probs_%s_sel1" % (name, mode)]
        full = m._last["probs"].cpu().numpy()
        srt = np.sort(full, axis=-1)
        clear = (srt[:, -1] - srt[:, -2]) > 2 * tol
        assert clear.sum() > 0
        assert np.array_equal(pm1[clear], ref1[clear])


def test_step_batcher_is_bit_identical_to_per_sequence_calls(model, golden_dir):
    """StepBatcher: the steps of several trackers in ONE Decision-Transformer launch (each step keeps its own two ReID
    BatchNorm batches).  Outputs are bit-identical to separate associate_embeddings calls."""
    from busca_amd.batcher import StepBatcher
    model.pinned_numpy = True
    model._dirty = True
    g = np.load(os.path.join(golden_dir, "assoc.npz"))
    cases = [_case(ci) for ci in (0, 1, 2, 3, 1)]
    singles = []
    for sel, (name, tracks, dets, kals, P) in zip((False, True, False, True, False), cases):
        singles.append(model.associate_embeddings(tracks, dets, g[name + "_dists"], 11, P, True, sel, extra_kalman_candidates=kals, normalize_ims=True))
    b = StepBatcher(model)
    tickets = []
    for sel, (name, tracks, dets, kals, P) in zip((False, True, False, True, False), cases):
        tickets.append(b.submit(tracks, dets, g[name + "_dists"], 11, P, True, sel, extra_kalman_candidates=kals, normalize_ims=True))
    empty = b.submit([], [], None, 11, 5, True, True)
    assert empty.result() == (None, None)
    b.flush()
    assert b.launches == 1 and b.steps == 5
    for (pm, rel), t in zip(singles, tickets):
        bpm, brel = t.result()
        assert np.array_equal(pm, bpm) and np.array_equal(rel, brel)


def test_repeated_crops_are_computed_once(golden_dir):
    """Candidate batches repeat crops (every track takes its P nearest detections; padding is all-zero crops): by default each
    distinct crop is computed once with weighted BatchNorm statistics.  Exact flavours: same output as the expanded batch to
    float32 round-off, and still on the reference's golden output."""
    g = np.load(os.path.join(golden_dir, "assoc.npz"))
    m = _model(64, 128, 17, "f32", "f32")
    for ci in (0, 1, 2):
        name, tracks, dets, kals, P = _case(ci)
        m.dedup_crops = True
        a, _ = m.associate_embeddings(tracks, dets, g[name + "_dists"], 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
        uniq, slots = m.last_unique
        m.dedup_crops = False
        b, _ = m.associate_embeddings(tracks, dets, g[name + "_dists"], 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
        assert m.last_unique[0] == m.last_unique[1] and uniq < slots          # the candidate batch really had repeats
        assert np.abs(a - b).max() <= 5e-5, np.abs(a - b).max()
        assert np.abs(a - g["%s_probs_f64_sel0" % name]).max() <= 2e-4


def test_step_batcher_keeps_the_kernel_flavour_of_separate_calls():
    """Merged steps whose tracks add up to more than 256 (where the f16 fused kernel would switch to two tracks per workgroup) still
    return what separate calls return, bit for bit: the batcher pins the merged launch to the flavour each step gets alone.  Also
    BUSCA.reserve: every workspace of the largest step allocated ahead of time."""
    from busca_amd.batcher import StepBatcher
    m = _model(256, 512, 31, "f16", "f16")
    m.reserve(160, 11, 16)
    jobs = []
    for k, B in enumerate((150, 140)):
        inp = synth.dt_inputs(900 + k, B, 11, 16)
        jobs.append({kk: torch.from_numpy(v).cuda() for kk, v in inp.items()})
    m._sync()
    single = [m._dt.forward(j["mem_feat"], j["can_feat"], j["mem_boxes"], j["can_boxes"])["logits"].cpu().numpy() for j in jobs]
    assert m._ctx.get_option("last_dt_ntrk") == 1
    cat = {k: torch.cat([j[k] for j in jobs], 0) for k in jobs[0]}
    auto = m._dt.forward(cat["mem_feat"], cat["can_feat"], cat["mem_boxes"], cat["can_boxes"])["logits"].cpu().numpy()
    assert m._ctx.get_option("last_dt_ntrk") == 2 and not np.array_equal(auto, np.concatenate(single))      # the automatic choice WOULD differ
    m._ctx.set_option("dt_ntrk", 1)                            # what StepBatcher.flush does around its merged launch
    try:
        pinned = m._dt.forward(cat["mem_feat"], cat["can_feat"], cat["mem_boxes"], cat["can_boxes"])["logits"].cpu().numpy()
    finally:
        m._ctx.set_option("dt_ntrk", 0)
    assert np.array_equal(pinned, np.concatenate(single))
    src = open(StepBatcher.flush.__code__.co_filename).read()       # flush pins the flavour around its merged launch and puts the caller's value back
    assert "prev if prev != 0 else ntrk" in src and "set_option(\"dt_ntrk\", prev)" in src


def test_fast_flavours_decide_like_the_exact_ones():
    """tools/decision_agreement.py on 2 000 seeded association steps at the shipped shape (d=512, P=5, Kalman candidates, complete and
    incomplete memories, candidate batches that repeat detections): the default flavour (float32 DT + fp16 ReID) and the fastest one
    (f16 DT + fp16 ReID) against the exact one (float32 ReID + float32 DT).  Stated bounds: no probability moves by more than 0.05,
    so `probs[i, N + i] > busca_thresh` (0.3 / 0.5 in the shipped configs) can only flip for a track whose exact probability lies
    within 0.05 of the threshold, and the one-hot winner only where the exact top-2 margin is below 0.1.  Random weights put a fifth of
    all tracks that close to 0.3, i.e. the measured flip rates (about 1 %) are an upper bound for any sharper, trained model."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    import decision_agreement as da
    r = da.run(2000)
    assert r["steps"] == 2000 and r["tracks"] > 8000 and r["tracks_with_incomplete_memory"] > 3000
    for c in r["comparisons"]:
        print(c["flavour"], {k: c[k] for k in ("kalman_gt_0.3", "kalman_gt_0.5", "winner")}, c["abs_delta_prob"]["max"])
        x3 = "x3" in c["flavour"]
        # the split-fp16 ReID flavour is float32-class: probabilities within 1e-3 of the exact flavour's (the fp16 ReID moves them by up to 0.03)
        assert c["abs_delta_prob"]["max"] <= (1e-3 if x3 else 0.05) and c["abs_delta_prob"]["p99"] <= (3e-4 if x3 else 0.025)
        for t in ("kalman_gt_0.3", "kalman_gt_0.5"):
            assert c[t]["flip_rate"] <= (1e-3 if x3 else 0.02)
            assert c[t]["largest_distance_to_threshold_among_flips"] <= c["abs_delta_kalman_prob"]["max"] + 1e-12
        assert c["winner"]["largest_exact_margin_among_flips"] <= 2 * c["abs_delta_prob"]["max"] + 1e-12


def test_decisions_next_to_the_half_threshold():
    """The same comparison with a SHARPENED model (decoder output layer x 12): the exact Kalman probabilities now spread over (0, 1)
    and a share of the tracks lies within 0.05 of the 0.5 threshold of config/StrongSORT, so 'flips at 0.5' measures something
    (with the plain random weights every Kalman probability is below 0.31).  The float32-equivalent flavour must decide like the
    exact one; the fp16 ReID flavours' flip rate is reported and bounded by the share of tracks that close to the threshold."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools"))
    import decision_agreement as da
    r = da.run(600, decoder_gain=da.SHARP_GAIN)
    q = r["exact_kalman_prob_quantiles"]
    print("exact Kalman probability quantiles (sharpened):", q)
    assert q["0.05"] < 0.1 and q["0.95"] > 0.6, q
    for c in r["comparisons"]:
        near = c["kalman_gt_0.5"]["tracks_within_0.05_of_threshold"]
        print(c["flavour"], c["kalman_gt_0.5"], c["winner"], c["abs_delta_prob"]["max"])
        assert near >= 0.02 * r["tracks"], near                      # the threshold is populated
        if "x3" in c["flavour"]:
            assert c["abs_delta_prob"]["max"] <= 5e-3 and c["kalman_gt_0.5"]["flip_rate"] <= 2e-3 and c["winner"]["flip_rate"] <= 2e-3
        else:
            assert c["kalman_gt_0.5"]["flips"] <= near and c["kalman_gt_0.5"]["flip_rate"] <= 0.05


def test_associate_never_returns_a_clipped_x3_step(golden_dir):
    """A checkpoint whose activations leave the split-fp16 operand range (LayerNorm weights x 3000 -> |x| ~ 9000 > 1023.5): the default x3 Decision Transformer
    clips, raises `dt_status` 2, and associate_embeddings - already synchronised on the probabilities - runs that step again in exact float32 and returns THAT:
    the f32 flavour's probabilities bit for bit, no exception in this frame or the next (busca/network.py:401-405 cannot fail there)."""
    from busca_amd.network import BUSCA

    def build(prec):
        a = _args(precision=prec)
        a.reid_precision = "x3"
        m = BUSCA(a).to(torch.device("cuda:0")).eval()
        sd = dict(synth.dt_state_dict(17, d=64, ff=128))
        sd["transformer_encoder.layers.1.norm1.weight"] = sd["transformer_encoder.layers.1.norm1.weight"] * 3000.0
        sd.update({"reid_encoder.model." + k: v for k, v in synth.reid_state_dict(17).items()})
        m.load_state_dict(sd)
        return m

    g = np.load(os.path.join(golden_dir, "assoc.npz"))
    name, tracks, dets, kals, P = _case(1)
    dists = g[name + "_dists"]
    m32, m3 = build("f32"), build("x3")
    want, rel32 = m32.associate_embeddings(tracks, dets, dists, 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
    for frame in range(2):               # twice: the re-run leaves nothing behind for the next frame to trip over
        got, rel = m3.associate_embeddings(tracks, dets, dists, 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
        assert np.array_equal(got, want) and np.array_equal(rel, rel32)
        assert m3._dt.exact_reruns == frame + 1 and m3._ctx.get_option("dt_status") == 0
    got1, _ = m3.associate_embeddings(tracks, dets, dists, 11, P, True, True, extra_kalman_candidates=kals, normalize_ims=True)
    want1, _ = m32.associate_embeddings(tracks, dets, dists, 11, P, True, True, extra_kalman_candidates=kals, normalize_ims=True)
    assert np.array_equal(got1, want1)


def test_associate_reruns_an_overflowed_x3_reid_pass_in_f32(golden_dir):
    """A ReID checkpoint whose activations leave the split-fp16 operand range (a BatchNorm affine x 4000): the default x3 extractor reports it (`reid_status` 2), and
    associate_embeddings - synchronised on the probabilities - computes both BatchNorm batches of the step again on the exact-f32 extractor: the result is
    what a model built with reid_precision="f32" returns, bit for bit, in this frame and the next, nothing raised; a healthy checkpoint is never re-run."""
    from busca_amd.network import BUSCA

    def build(reid_prec, hot):
        a = _args(precision="x3")
        a.reid_precision = reid_prec
        m = BUSCA(a).to(torch.device("cuda:0")).eval()
        sd = dict(synth.dt_state_dict(17, d=64, ff=128))
        rsd = dict(synth.reid_state_dict(17))
        if hot:
            rsd["layer2.0.bn1.weight"] = rsd["layer2.0.bn1.weight"] * 4000.0
            rsd["layer2.0.bn1.bias"] = rsd["layer2.0.bn1.bias"] * 4000.0
        sd.update({"reid_encoder.model." + k: v for k, v in rsd.items()})
        m.load_state_dict(sd)
        return m

    g = np.load(os.path.join(golden_dir, "assoc.npz"))
    name, tracks, dets, kals, P = _case(2)
    dists = g[name + "_dists"]
    m32, m3, healthy = build("f32", True), build("x3", True), build("x3", False)
    want, rel32 = m32.associate_embeddings(tracks, dets, dists, 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
    assert np.isfinite(want).all()
    for frame in range(2):
        got, rel = m3.associate_embeddings(tracks, dets, dists, 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
        assert np.array_equal(got, want) and np.array_equal(rel, rel32)
        assert m3.reid_exact_reruns == frame + 1 and m3._ctx.get_option("reid_status") == 0
    healthy.associate_embeddings(tracks, dets, dists, 11, P, True, False, extra_kalman_candidates=kals, normalize_ims=True)
    assert getattr(healthy, "reid_exact_reruns", 0) == 0 and healthy._dt.exact_reruns == 0
