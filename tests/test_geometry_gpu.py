"""GPU parity (bit-exact): pairwise distance / IoU / top-P / crop kernels vs the numpy oracle."""
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


def _boxes(seed, n, big=False):
    cx = synth.uniform(seed, "cx", (n,), 0, 1920).astype(np.float64)
    cy = synth.uniform(seed, "cy", (n,), 0, 1080).astype(np.float64)
    h = synth.uniform(seed, "h", (n,), 10, 400).astype(np.float64)
    w = h * synth.uniform(seed, "ar", (n,), 0.2, 0.6).astype(np.float64)
    return np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)


@pytest.mark.parametrize("nA,nB", [(1, 1), (32, 150), (128, 150), (17, 65), (300, 1000), (0, 5), (5, 0)])
@pytest.mark.parametrize("mode", ["center", "center_w", "iou", "iou_cost", "fuse"])
def test_pairwise_bit_exact(ctx, nA, nB, mode):
    from busca_amd import _lib, geometry as G
    from oracle import geometry as og
    a, b = _boxes(1 + nA, nA), _boxes(2 + nB, nB)
    if nA > 4 and nB > 4:
        b[3] = a[2]            # identical boxes (IoU 1, distance 0)
        b[4, 2:] = b[4, :2]    # zero-area box
    sc = synth.uniform(9, "scores", (nB,), 0.1, 1.0).astype(np.float64)
    if mode == "center":
        got, ref = G.pairwise(ctx, a, b, _lib.PAIR_CENTER), og.center_distance(a, b)
    elif mode == "center_w":
        got, ref = G.pairwise(ctx, a, b, _lib.PAIR_CENTER_WEIGHTED), og.center_distance(a, b, weight_size=True)
    elif mode == "iou":
        got, ref = G.pairwise(ctx, a, b, _lib.PAIR_IOU), og.iou_matrix(a, b)
    elif mode == "iou_cost":
        got, ref = G.pairwise(ctx, a, b, _lib.PAIR_IOU_COST), og.iou_distance(a, b)
    else:
        got, ref = G.pairwise(ctx, a, b, _lib.PAIR_IOU_COST, scores_b=sc), og.fuse_score(og.iou_distance(a, b), sc)
    got = got.cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref, equal_nan=True)
    # the host-table route of the per-frame tracker calls (boxes in, matrix out through one pinned table; tracking.center_distance / iou_distance)
    hm = {"center": _lib.PAIR_CENTER, "center_w": _lib.PAIR_CENTER_WEIGHTED, "iou": _lib.PAIR_IOU}.get(mode, _lib.PAIR_IOU_COST)
    host = G.pairwise_host(ctx, a, b, hm, scores_b=sc if mode == "fuse" else None)
    assert isinstance(host, np.ndarray) and host.shape == ref.shape and np.array_equal(host, ref, equal_nan=True)


def test_center_distance_matches_scipy(ctx):
    """The oracle's sqrt(dx*dx+dy*dy) is what scipy's cdist computes (busca/tracking.py:48)."""
    from scipy.spatial.distance import cdist
    from oracle import geometry as og
    a, b = _boxes(5, 40), _boxes(6, 77)
    ac = (a[:, :2] + a[:, 2:]) / 2.0
    bc = (b[:, :2] + b[:, 2:]) / 2.0
    assert np.array_equal(og.center_distance(a, b), cdist(ac, bc, metric="euclidean"))


@pytest.mark.parametrize("B,N,P", [(32, 150, 16), (128, 150, 32), (3, 4, 16), (7, 0, 5), (64, 1000, 64), (1, 1, 1),
                                   (5, 255, 32), (5, 256, 300), (5, 257, 7), (9, 511, 64), (9, 512, 64), (4, 1024, 33), (6, 1100, 40)])
def test_topk_rows_bit_exact(ctx, B, N, P):
    """Every flavour of the top-P kernel (rank kernel with 1 / 2 / 4 keys per thread, odd and even rows, the round-by-round kernel beyond 1 024 columns):
    exact ties, +-0.0, inf, NaN, all-equal rows, more slots than columns."""
    from busca_amd import geometry as G
    from oracle import geometry as og
    d = synth.uniform(B + N + P, "d", (B, N), 0, 500).astype(np.float64)
    if N > 8:
        d[:, 5] = d[:, 2]          # exact ties -> lower index first
        d[0, 3] = np.inf
        d[0, 4] = -0.0
        d[0, 6] = 0.0
        d[1, 7] = np.nan           # np.argsort: NaN last
        d[1, N - 1] = np.nan
        d[2, :] = 3.25             # a whole row of ties: indices in order
    got = G.topk_rows(ctx, d, P).cpu().numpy()
    ref = og.topk_rows(d, P)
    assert np.array_equal(got, ref)
    if B and N:                  # the host-table route of associate_embeddings (pinned rows in, pinned indices out): same kernel, same answer
        assert np.array_equal(G.topk_rows_host(ctx, d, P), ref)


def _frame(seed, H, W):
    return synth.randint_u8(seed, "frame", (H, W, 3))


def test_crop_gather_bit_exact(ctx):
    """Device crops vs the oracle's restatement of cutout + OpenCV fixed-point bilinear (bit-exact between
    the two restatements; both are +-1 LSB-unpinned against the real cv2, see oracle/geometry.py)."""
    from busca_amd import geometry as G
    from oracle import geometry as og
    H, W = 540, 960
    fr = _frame(3, H, W)
    boxes = np.array([
        [100.3, 50.2, 180.9, 300.7],      # interior, upscale in x, mixed in y
        [-20.5, -30.0, 60.2, 200.0],      # clipped top-left -> mean padding
        [900.0, 400.0, 1000.0, 600.0],    # clipped bottom-right
        [10.0, 10.0, 138.0, 394.0],       # exactly 128 x 384 -> copy
        [200.0, 100.0, 456.0, 868.0],     # 256 x 768 (clipped) -> exact 2x path with padding
        [300.0, 20.0, 556.0, 532.0],      # 256 x 512
        [5.5, 5.5, 6.2, 6.1],             # tiny 1x1 box
        [2000.0, 2000.0, 2100.0, 2200.0], # fully outside: empty clipped crop, all padding (fill 0)
        [50.0, 60.0, 50.0, 60.0],         # zero extent -> empty cutout -> zeros
        [400.0, 100.0, 1000.0, 539.5],    # big downscale
    ], dtype=np.float32)
    u8, f16 = G.crop_gather(ctx, fr, boxes, want_u8=True, want_f16=True)
    torch.cuda.synchronize()
    got = u8.cpu().numpy()
    for i, bx in enumerate(boxes):
        ref = og.get_bbox_crop(fr, bx)
        assert ref.shape == (384, 128, 3)
        assert np.array_equal(got[i], ref), "crop %d differs (max %d)" % (i, np.abs(got[i].astype(int) - ref.astype(int)).max())
    # normalised fp16 RGB0 layout == fp16(normalize_bgr(u8))[..., ::-1]
    ref_n = og.normalize_bgr(got)[..., ::-1].astype(np.float16)
    gn = f16.cpu().numpy()
    assert np.array_equal(gn[..., :3], ref_n)
    assert (gn[..., 3] == 0).all()


def test_crops_of_a_few_boxes_upload_only_their_part_of_the_frame(ctx):
    """The unchanged StrongSORT adapter calls get_image_crops once per detection (deep_sort/tracker.py:126,273,291).  Outside a frame scope the host frame is
    read live on every call, but only the sub-frame spanned by the clipped boxes is uploaded (geometry._frame_for_rects): the crops must be the oracle's bytes
    for boxes inside the frame, across each edge, across two edges, wholly outside, of zero extent - alone and in small groups - and equal to what the
    whole-frame route (a frame that is already a device tensor) cuts."""
    from busca_amd import geometry as G
    from oracle import geometry as og
    H, W = 540, 960
    fr = _frame(7, H, W)
    dev_fr = torch.from_numpy(fr).cuda()
    boxes = np.array([
        [100.3, 50.2, 180.9, 300.7], [-20.5, -30.0, 60.2, 200.0], [900.0, 400.0, 1000.0, 600.0], [930.2, -12.0, 975.5, 90.0], [-5.0, 500.0, 40.0, 560.0],
        [5.5, 5.5, 6.2, 6.1], [2000.0, 2000.0, 2100.0, 2200.0], [-300.0, 100.0, -200.0, 300.0], [50.0, 60.0, 50.0, 60.0], [400.0, 100.0, 520.0, 439.5],
    ], dtype=np.float64)
    seen_sub = 0
    groups = [[i] for i in range(len(boxes))] + [[0, 9], [1, 4], [2, 3, 6], [6, 7], [5, 8, 0]]
    for gsel in groups:
        bx = boxes[gsel]
        t, shifted = G._frame_for_rects(ctx, fr, __import__("busca_amd.tracking", fromlist=["box_extents"]).box_extents(bx), torch.device("cuda", ctx.device))
        seen_sub += int(tuple(t.shape[:2]) != (H, W))
        got = G.crop_gather(ctx, fr, bx, want_u8=True)[0].cpu().numpy()
        full = G.crop_gather(ctx, dev_fr, bx, want_u8=True)[0].cpu().numpy()
        for k, i in enumerate(gsel):
            assert np.array_equal(got[k], og.get_bbox_crop(fr, boxes[i])), (gsel, i)
        assert np.array_equal(got, full), gsel
        sized = G.crop_gather_sized(ctx, fr, bx, 64, 192).cpu().numpy()
        for k, i in enumerate(gsel):
            assert np.array_equal(sized[k], og.get_bbox_crop(fr, boxes[i], output_size=(64, 192))), (gsel, i)
    assert seen_sub >= 10                      # the sub-frame route really ran (boxes wholly outside / of zero extent fall back to the whole frame)
    fr2 = fr.copy()
    fr2[60:200, 110:170] = 255 - fr2[60:200, 110:170]              # the live array changed between two calls: the second call must see it
    assert not np.array_equal(G.crop_gather(ctx, fr2, boxes[:1], want_u8=True)[0].cpu().numpy(), G.crop_gather(ctx, fr, boxes[:1], want_u8=True)[0].cpu().numpy())
    assert np.array_equal(G.crop_gather(ctx, fr2, boxes[:1], want_u8=True)[0].cpu().numpy()[0], og.get_bbox_crop(fr2, boxes[0]))


@pytest.mark.parametrize("out_wh", [(64, 192), (96, 96), (128, 256), (50, 37), (256, 768)])
def test_crops_of_other_output_sizes_bit_exact(ctx, out_wh):
    """`get_image_crops(output_size=(w, h))` / `get_bbox_crop(output_size=...)` for sizes other than the ReID crop (busca/network.py:492-507,
    busca/tracking.py:62-78): the generic kernel against the oracle's cut-out + OpenCV fixed-point bilinear restatement, bit for bit - including the
    copy (box extent == output size), the exact 2x shrink, padded, tiny, empty and fully-outside boxes; normalised crops and the empty-list shape too."""
    from busca_amd import tracking
    from oracle import geometry as og
    w, h = out_wh
    fr = _frame(5, 540, 960)
    boxes = np.array([
        [100.3, 50.2, 180.9, 300.7],
        [-20.5, -30.0, 60.2, 200.0],
        [900.0, 400.0, 1000.0, 600.0],
        [10.0, 10.0, 10.0 + w, 10.0 + h],             # exactly the output size -> copy
        [200.0, 20.0, 200.0 + 2 * w, 20.0 + 2 * h],   # exactly twice -> the 2x2 area path (clipped at the frame edge for the large sizes)
        [5.5, 5.5, 6.2, 6.1],
        [2000.0, 2000.0, 2100.0, 2200.0],
        [50.0, 60.0, 50.0, 60.0],
        [400.0, 100.0, 1000.0, 539.5],
    ], dtype=np.float64)
    got = tracking.get_image_crops(fr, boxes, normalize=False, ctx=ctx, output_size=(w, h))
    assert type(got) is np.ndarray and got.dtype == np.uint8 and got.shape == (len(boxes), h, w, 3)
    for i, bx in enumerate(boxes):
        ref = og.get_bbox_crop(fr, bx, output_size=(w, h))
        assert np.array_equal(got[i], ref), "crop %d differs (max %d)" % (i, np.abs(got[i].astype(int) - ref.astype(int)).max())
    one = tracking.get_bbox_crop(fr, boxes[0], output_size=(w, h), normalize=True, ctx=ctx)
    assert one.dtype == np.float32 and np.array_equal(one, og.get_bbox_crop(fr, boxes[0], output_size=(w, h), normalize=True))
    assert np.array_equal(tracking.get_image_crops(fr, boxes[:2], normalize=True, ctx=ctx, output_size=(w, h)), og.normalize_bgr(got[:2]))
    assert tracking.get_image_crops(fr, [], normalize=False, ctx=ctx, output_size=(w, h)).shape == (0, w, h, 3)      # the reference's transposed empty shape


def test_crop_gather_empty(ctx):
    from busca_amd import geometry as G
    u8, _ = G.crop_gather(ctx, _frame(1, 64, 64), np.zeros((0, 4), np.float32))
    assert tuple(u8.shape) == (0, 384, 128, 3)


def test_detection_coverage_bit_exact(ctx):
    """Reliability-gate coverage (union area of filled rectangles) vs the oracle's boolean-canvas restatement."""
    import types
    from busca_amd import tracking
    from oracle import geometry as og
    H, W = 1080, 1920
    b = _boxes(77, 300)
    b[0] = [-50.7, -20.2, 30.9, 80.1]          # clipped at the top-left, negative coordinates truncate toward zero
    b[1] = [1900.0, 1000.0, 2500.0, 1500.0]    # clipped bottom-right
    b[2] = [3000.0, 100.0, 3100.0, 200.0]      # outside
    b[3] = [400.9, 300.9, 400.1, 300.1]        # degenerate: a single pixel after truncation
    b[4] = [500.0, 600.0, 450.0, 550.0]        # corners swapped
    tracks = [types.SimpleNamespace(tlbr=bb / 1.25, scale=1.25) for bb in b]
    got = tracking.get_detection_coverage((H, W, 3), tracks[:200], tracks[200:], ctx=ctx)
    ref = og.detection_coverage((H, W, 3), [np.array(t.tlbr) * t.scale for t in tracks])
    assert got["area_covered"] == ref["area_covered"] and got["area_covered_per_obj"] == ref["area_covered_per_obj"]
    assert got["max_bbox_area"] == ref["max_bbox_area"] and got["average_bbox_area"] == ref["average_bbox_area"]
    assert got["bbox_areas"] == ref["bbox_areas"]
    empty = tracking.get_detection_coverage((H, W, 3), [], [], ctx=ctx)
    assert empty["area_covered"] == 0.0 and empty["bbox_areas"] == []
    assert tracking.is_reliable((H, W, 3), tracks[:50], (3.0, 0.0), ctx=ctx) in (True, False)


def test_crops_always_come_from_the_live_frame(ctx):
    """busca/network.py:492-507 cuts from the array it is given.  Rounds 2-4 cached the uploaded frame behind object identity + a sparse pixel
    fingerprint, which an edit between two calls could miss (and a recycled id / buffer address could alias): the cache is gone.  An edit of one
    8 x 8 block between two calls on the SAME array - placed so that a 29 x 43 sampling grid would have missed it - and a freshly allocated array
    (the allocator is free to hand the old buffer out again) both give the new pixels; only inside an explicit `begin_frame` scope is an upload reused
    (adapters/ByteTrack/yolox/tracker/byte_tracker.py:280-282 cuts three times from one frame)."""
    from busca_amd import geometry, tracking
    H, W = 1080, 1920
    frame = synth.randint_u8(77, "frame", (H, W, 3))
    box = np.array([[100.0, 100.0, 164.0, 292.0]])           # covers the edited block
    first = np.asarray(tracking.get_image_crops(frame, box, normalize=False, ctx=ctx, host_copy="eager"))
    sy, sx = H // 29, W // 43
    y0, x0 = 3 * sy + 5, 3 * sx + 3                           # strictly between grid lines: rows 116..123, columns 135..142
    assert y0 % sy >= 5 and (y0 + 7) % sy < sy and x0 % sx >= 3 and (x0 + 7) // sx == x0 // sx
    frame[y0:y0 + 8, x0:x0 + 8] = 255 - frame[y0:y0 + 8, x0:x0 + 8]
    second = np.asarray(tracking.get_image_crops(frame, box, normalize=False, ctx=ctx, host_copy="eager"))
    want = np.asarray(tracking.get_image_crops(frame.copy(), box, normalize=False, ctx=ctx, host_copy="eager"))
    assert not np.array_equal(first, second) and np.array_equal(second, want)
    # a new array on (very likely) the old address, other pixels
    addr = frame.__array_interface__["data"][0]
    del frame
    fresh = np.empty((H, W, 3), np.uint8)
    fresh[:] = synth.randint_u8(78, "frame", (H, W, 3))
    third = np.asarray(tracking.get_image_crops(fresh, box, normalize=False, ctx=ctx, host_copy="eager"))
    assert np.array_equal(third, np.asarray(tracking.get_image_crops(fresh.copy(), box, normalize=False, ctx=ctx, host_copy="eager")))
    assert not np.array_equal(third, second), (addr, fresh.__array_interface__["data"][0])
    # the explicit scope: one upload, reused by identity until end_frame; other arrays are still uploaded
    geometry.begin_frame(ctx, fresh)
    try:
        up = ctx._frame_scope[1]
        assert geometry._frame_on_device(ctx, fresh, up.device) is up
        inside = np.asarray(tracking.get_image_crops(fresh, box, normalize=False, ctx=ctx, host_copy="eager"))
        other = fresh.copy(); other[100:300, 100:170] = 7
        assert geometry._frame_on_device(ctx, other, up.device) is not up
        assert not np.array_equal(np.asarray(tracking.get_image_crops(other, box, normalize=False, ctx=ctx, host_copy="eager")), inside)
    finally:
        geometry.end_frame(ctx)
    assert np.array_equal(inside, third)
    assert getattr(ctx, "_frame_scope", None) is None


@pytest.mark.parametrize("hw", [(540, 960), (487, 957), (1080, 1920)])
def test_band_crop_kernel_bit_exact(ctx, hw):
    """Round 5: the LDS-staged crop kernel (crop_band_kernel: one workgroup per crop and band of 16 output rows, source rows staged with aligned
    16-byte loads, 16-byte stores) against the oracle's cut-out + OpenCV fixed-point bilinear restatement (busca/tracking.py:62-113) AND against the
    one-thread-per-pixel kernel it replaces, bit for bit: interior / clipped / copy / exact 2x / tiny / outside / empty boxes, boxes at every byte
    alignment, a frame whose row stride is not a multiple of 16, boxes too big for the LDS band (global fall-back inside the kernel), both output
    routes (packed batch, per-crop destinations)."""
    from busca_amd import geometry as G
    from oracle import geometry as og
    H, W = hw
    fr = _frame(31 + W, H, W)
    fixed = [[100.3, 50.2, 180.9, 300.7], [-20.5, -30.0, 60.2, 200.0], [W - 60.0, H - 140.0, W + 40.0, H + 60.0], [10.0, 10.0, 138.0, 394.0],
             [200.0, 100.0, 456.0, 868.0], [300.0, 20.0, 556.0, 532.0], [5.5, 5.5, 6.2, 6.1], [2000.0, 2000.0, 2100.0, 2200.0], [50.0, 60.0, 50.0, 60.0],
             [40.0, 1.0, W - 2.0, H - 0.5], [-300.0, -200.0, W + 300.0, H + 200.0], [W - 1.0, 0.0, W + 50.0, 90.0], [-40.0, 30.0, 1.0, 130.0], [0.0, 0.0, float(W), float(H)]]
    rnd = _boxes(5 + H, 40)
    rnd[:, [0, 2]] *= W / 1920.0; rnd[:, [1, 3]] *= H / 1080.0
    al = np.array([[17.0 + k, 9.0, 17.0 + k + 37.0 + (k % 5), 9.0 + 111.0 + k] for k in range(16)])        # every byte alignment of the first source pixel
    boxes = np.concatenate([np.array(fixed), rnd, al]).astype(np.float64)
    assert ctx.get_option("crop_band") == 1
    u8, _ = G.crop_gather(ctx, fr, boxes, want_u8=True)
    got = u8.cpu().numpy()
    ctx.set_option("crop_band", 0)
    try:
        old, _ = G.crop_gather(ctx, fr, boxes, want_u8=True)
    finally:
        ctx.set_option("crop_band", 1)
    assert np.array_equal(got, old.cpu().numpy())
    for i, bx in enumerate(boxes):
        ref = og.get_bbox_crop(fr, bx)
        assert np.array_equal(got[i], ref), "crop %d %s differs (max %d)" % (i, bx, np.abs(got[i].astype(int) - ref.astype(int)).max())
    # per-crop destinations (pool slots) + packed copy in one launch
    dst = torch.zeros(len(boxes), 384, 128, 3, dtype=torch.uint8, device="cuda")
    order = np.random.default_rng(3).permutation(len(boxes))
    ptrs = (dst.data_ptr() + order.astype(np.uint64) * np.uint64(384 * 128 * 3)).astype(np.uint64)
    packed, _ = G.crop_gather(ctx, fr, boxes, want_u8=True, dst_ptrs=ptrs)
    torch.cuda.synchronize()
    assert np.array_equal(packed.cpu().numpy(), got)
    assert np.array_equal(dst.cpu().numpy()[order], got)
