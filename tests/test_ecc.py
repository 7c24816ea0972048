"""Camera-motion compensation: busca_ecc_align against the oracle restatement of cv2.findTransformECC and
against known transforms.  (cv2 itself is third-party and absent: parity unpinned, see oracle/ecc.py.)"""
import numpy as np
import pytest


def _pair(H=120, W=160, th=0.01, tx=2.3, ty=-1.4, seed=0):
    """Smooth random image and a copy moved by a known Euclidean transform: im1(x) ~ im2(M x).  BGR frames."""
    from scipy.ndimage import affine_transform, zoom
    rng = np.random.default_rng(seed)
    chans1, chans2 = [], []
    c, s = np.cos(th), np.sin(th)
    A = np.array([[c, -s], [s, c]])
    Ainv = np.linalg.inv(A)
    R = np.array([[Ainv[1, 1], Ainv[1, 0]], [Ainv[0, 1], Ainv[0, 0]]])
    off = -(R @ np.array([ty, tx]))
    for _ in range(3):
        big = zoom(rng.uniform(0, 255, (H // 8 + 4, W // 8 + 4)), 8, order=3)[16:16 + H, 16:16 + W]
        chans1.append(np.clip(big, 0, 255))
        chans2.append(np.clip(affine_transform(big, R, offset=off, order=3, mode="nearest"), 0, 255))
    im1, im2 = np.stack(chans1, -1).astype(np.uint8), np.stack(chans2, -1).astype(np.uint8)
    return im1, im2, np.array([[c, -s, tx], [s, c, ty]])


def test_oracle_recovers_known_transform():
    from oracle import ecc
    im1, im2, M = _pair()
    rho, W = ecc.find_transform_ecc(ecc.bgr2gray(im1), ecc.bgr2gray(im2), motion="euclidean")
    assert rho > 0.99 and np.abs(W - M).max() < 0.02
    rho6, W6 = ecc.find_transform_ecc(ecc.bgr2gray(im1), ecc.bgr2gray(im2), motion="affine")
    assert rho6 > 0.99 and np.abs(W6 - M).max() < 0.03
    g = ecc.bgr2gray(np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8))
    assert g.tolist() == [[29, 150, 76, 22]]                   # cv2 BGR2GRAY known answers (0.114 B + 0.587 G + 0.299 R)
    assert np.allclose(ecc.warp_pos([10.0, 20.0], np.array([[1, 0, 2.5], [0, 1, -1.5]])), [12.5, 18.5])


@pytest.mark.gpu
@pytest.mark.parametrize("motion", ["MOTION_EUCLIDEAN", "MOTION_AFFINE"])
def test_ecc_matches_oracle(motion):
    """Iterate by iterate: after k = 1..6 iterations (termination test off) the kernel's warp and correlation coefficient equal
    the oracle's to float32 round-off.  The free-running call (eps 1e-5, <= 100 iterations, the reference's settings) is then
    compared on what matters - where the warp sends points - because near the optimum the rho increments hover around the
    termination threshold and round-off decides at which iteration the loop stops (and, for the affine model on a
    near-identity pair, the iterates themselves separate after ~20 iterations)."""
    from busca_amd import tracking
    from oracle import ecc
    pts = np.array([[x, y, 1.0] for x in (0, 80, 159) for y in (0, 60, 119)]).T
    for seed, (th, tx, ty) in enumerate([(0.01, 2.3, -1.4), (-0.02, -3.1, 0.8), (0.0, 0.4, 0.2)]):
        im1, im2, M = _pair(th=th, tx=tx, ty=ty, seed=seed)
        g1, g2 = ecc.bgr2gray(im1), ecc.bgr2gray(im2)
        _, _, trace = ecc.find_transform_ecc(g1, g2, motion=motion[7:].lower(), iters=6, eps=-1.0, return_trace=True)
        for k in (1, 2, 4, 6):
            cc, W = tracking.find_transform_ecc(im1, im2, motion=motion, number_of_iterations=k, termination_eps=-1.0)
            rho_k, W_k = trace[k - 1]
            assert tracking.find_transform_ecc.last_iterations == k
            assert abs(cc - rho_k) < 1e-5, (seed, k, cc, rho_k)
            assert np.abs(W - W_k).max() < 2e-5, (seed, k, np.abs(W - W_k).max())
        cc, W = tracking.find_transform_ecc(im1, im2, motion=motion)                  # the reference's settings
        rho, Wo = ecc.find_transform_ecc(g1, g2, motion=motion[7:].lower())
        assert cc > 0.99 and rho > 0.99
        lim = 0.1 if motion == "MOTION_EUCLIDEAN" else 0.25
        assert np.abs(W.astype(np.float64) @ pts - M @ pts).max() < lim               # both sit on the true transform
        assert np.abs(Wo.astype(np.float64) @ pts - M @ pts).max() < lim


@pytest.mark.gpu
def test_ecc_full_hd_and_track_update():
    """1080p frames; tracks are moved like STrack.apply_camera_motion does (byte_tracker.py:123-137)."""
    import types
    from busca_amd import tracking
    im1, im2, M = _pair(H=1080, W=1920, th=0.004, tx=5.5, ty=-2.25, seed=3)
    trk = [types.SimpleNamespace(mean=np.array([400.0, 300.0, 0.4, 200.0, 0, 0, 0, 0]), _tlwh=np.zeros(4), scale=1.0),
           types.SimpleNamespace(mean=None, _tlwh=np.array([100.0, 50.0, 40.0, 90.0]), scale=2.0)]
    cc = tracking.camera_motion_compensation(trk, im1, im2, frame_id=5)
    assert cc > 0.98
    want0 = M @ np.array([400.0, 300.0, 1.0])
    want1 = (M @ np.array([200.0, 100.0, 1.0])) / 2.0
    assert np.abs(trk[0].mean[:2] - want0).max() < 0.2 and np.abs(trk[1]._tlwh[:2] - want1).max() < 0.2
    assert tracking.camera_motion_compensation(trk, None, im2, frame_id=1) == 1.0            # first frame: nothing to align
    with pytest.raises(ValueError):
        tracking.find_transform_ecc(im1, im2, motion="MOTION_HOMOGRAPHY")
