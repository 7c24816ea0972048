"""Oracle (CPU checker, test infrastructure only): camera-motion compensation of
/root/reference/adapters/ByteTrack/yolox/tracker/byte_tracker.py:626-650 -
`cv2.cvtColor(BGR2GRAY)` on both frames + `cv2.findTransformECC(template=previous, input=current, eye(2,3), EUCLIDEAN|AFFINE,
(EPS|COUNT, 100, 1e-5))`.

PARITY UNPINNED: the arithmetic lives in opencv_python 4.7.0.72 (requirements.txt:2), which is not in /root/reference and
not installable here.  This file restates the published algorithm (Evangelidis & Psarakis, "Parametric Image Alignment
Using Enhanced Correlation Coefficient Maximization", PAMI 2008) in the order OpenCV's video/src/ecc.cpp evaluates it:
  gray      8-bit BGR2GRAY fixed point: (1868 B + 9617 G + 4899 R + 8192) >> 14
  blur      GaussianBlur 5x5, sigma 0 -> taps [1 4 6 4 1]/16, BORDER_REFLECT_101, float32 (both images)
  gradient  filter2D with [-0.5 0 0.5] (and its transpose) on the blurred input image, BORDER_REFLECT_101
  loop      warpAffine(INTER_LINEAR | WARP_INVERSE_MAP, constant 0 border) of image + gradients with OpenCV's fixed-point
            source coordinates (1/1024 pixel, interpolation weights from a 32 x 32 table), nearest-neighbour warp of the
            all-ones mask; masked mean / std; Jacobian; Hessian = J^T J; rho; lambda; delta p; warp update;
            stop when |rho - last_rho| < eps or after `iters` iterations.
Sums are taken in float64 (OpenCV: Mat::dot / meanStdDev), the small matrices in float32 like OpenCV's CV_32F Mats.
"""
import numpy as np

AB_BITS, INTER_BITS = 10, 5
AB_SCALE, TAB = 1 << AB_BITS, 1 << INTER_BITS


def bgr2gray(img):
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * (n - 1) - i, i)


def blur5(x):
    """float32 separable [1 4 6 4 1]/16, reflect-101 borders (rows then columns, as OpenCV's separable filter)."""
    k = np.array([1, 4, 6, 4, 1], np.float32) / np.float32(16)
    H, W = x.shape
    cols = _reflect101(np.arange(-2, W + 2), W)
    xp = x[:, cols]
    t = sum(k[i] * xp[:, i:i + W] for i in range(5)).astype(np.float32)
    rows = _reflect101(np.arange(-2, H + 2), H)
    tp = t[rows, :]
    return sum(k[i] * tp[i:i + H, :] for i in range(5)).astype(np.float32)


def gradients(x):
    H, W = x.shape
    c = _reflect101(np.arange(-1, W + 1), W)
    r = _reflect101(np.arange(-1, H + 1), H)
    gx = (np.float32(0.5) * x[:, c[2:]] - np.float32(0.5) * x[:, c[:-2]]).astype(np.float32)
    gy = (np.float32(0.5) * x[r[2:], :] - np.float32(0.5) * x[r[:-2], :]).astype(np.float32)
    return gx, gy


def _round_int(v):
    return np.rint(v).astype(np.int64)                    # saturate_cast<int>(double): round half to even


def warp_coords(M, H, W):
    """OpenCV warpAffine (inverse map) fixed-point source coordinates: integer part (sx, sy) and 5-bit fractions."""
    M = M.astype(np.float64)
    xs = np.arange(W)
    adelta = _round_int(M[0, 0] * xs * AB_SCALE)
    bdelta = _round_int(M[1, 0] * xs * AB_SCALE)
    rd = AB_SCALE // TAB // 2
    ys = np.arange(H)
    X0 = _round_int((M[0, 1] * ys + M[0, 2]) * AB_SCALE) + rd
    Y0 = _round_int((M[1, 1] * ys + M[1, 2]) * AB_SCALE) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    return X >> INTER_BITS, Y >> INTER_BITS, X & (TAB - 1), Y & (TAB - 1)


def warp_linear(img, coords):
    sx, sy, ax, ay = coords
    H, W = img.shape
    fx, fy = ax.astype(np.float32) / np.float32(TAB), ay.astype(np.float32) / np.float32(TAB)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        return np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0))
    w00, w01 = (1 - fy) * (1 - fx), (1 - fy) * fx
    w10, w11 = fy * (1 - fx), fy * fx
    return (tap(sy, sx) * w00 + tap(sy, sx + 1) * w01 + tap(sy + 1, sx) * w10 + tap(sy + 1, sx + 1) * w11).astype(np.float32)


def warp_mask_nearest(coords, H, W):
    """INTER_NEAREST of the all-ones mask: source = (X + 16) >> 5 on the same fixed-point grid."""
    sx, sy, ax, ay = coords
    nx, ny = sx + (ax >= TAB // 2), sy + (ay >= TAB // 2)
    return (nx >= 0) & (nx < W) & (ny >= 0) & (ny < H)


def find_transform_ecc(template_gray, input_gray, warp=None, motion="euclidean", iters=100, eps=1e-5, return_trace=False):
    """-> (rho, warp float32 [2,3]).  Raises RuntimeError where OpenCV raises StsNoConv."""
    T = blur5(template_gray.astype(np.float32))
    I = blur5(input_gray.astype(np.float32))
    gx, gy = gradients(I)
    H, W = T.shape
    M = np.eye(2, 3, dtype=np.float32) if warp is None else np.array(warp, np.float32).reshape(2, 3)
    Xg, Yg = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    rho, last_rho, trace = -1.0, -eps, []
    it = 1
    while it <= iters and abs(rho - last_rho) >= eps:
        co = warp_coords(M, H, W)
        mask = warp_mask_nearest(co, H, W)
        Iw, gxw, gyw = warp_linear(I, co), warp_linear(gx, co), warp_linear(gy, co)
        n = int(mask.sum())
        mI = Iw[mask].astype(np.float64).mean()
        mT = T[mask].astype(np.float64).mean()
        Izm = np.where(mask, Iw - np.float32(mI), Iw).astype(np.float32)
        Tzm = np.where(mask, T - np.float32(mT), np.float32(0)).astype(np.float32)
        img_norm = np.sqrt(n * Iw[mask].astype(np.float64).var())
        tmp_norm = np.sqrt(n * T[mask].astype(np.float64).var())
        if motion == "euclidean":
            h0, h1 = M[0, 0], M[1, 0]
            hatX, hatY = -(Xg * h1) - (Yg * h0), (Xg * h0) - (Yg * h1)
            J = [gxw * hatX + gyw * hatY, gxw, gyw]
        else:
            J = [gxw * Xg, gyw * Xg, gxw * Yg, gyw * Yg, gxw, gyw]
        J = [j.astype(np.float32) for j in J]
        dot = lambda a, b: float(np.dot(a.astype(np.float64).ravel(), b.astype(np.float64).ravel()))
        P = len(J)
        hess = np.array([[dot(J[i], J[j]) for j in range(P)] for i in range(P)], np.float32)
        hinv = np.linalg.inv(hess.astype(np.float64)).astype(np.float32)
        corr = dot(Tzm, Izm)
        last_rho = rho
        rho = corr / (img_norm * tmp_norm)
        if np.isnan(rho):
            raise RuntimeError("NaN encountered.")
        ip = np.array([dot(j, Izm) for j in J], np.float32)
        tp = np.array([dot(j, Tzm) for j in J], np.float32)
        iph = hinv @ ip
        lam_n = img_norm * img_norm - float(np.dot(ip.astype(np.float64), iph.astype(np.float64)))
        lam_d = corr - float(np.dot(tp.astype(np.float64), iph.astype(np.float64)))
        if lam_d <= 0.0:
            raise RuntimeError("The algorithm stopped before its convergence. The correlation is going to be minimized.")
        lam = lam_n / lam_d
        err = (np.float32(lam) * Tzm - Izm).astype(np.float32)
        ep = np.array([dot(j, err) for j in J], np.float32)
        dp = hinv @ ep
        if motion == "euclidean":
            th = np.float32(np.arcsin(M[1, 0])) + dp[0]
            M[0, 2] += dp[1]; M[1, 2] += dp[2]
            M[0, 0] = M[1, 1] = np.float32(np.cos(th)); M[1, 0] = np.float32(np.sin(th)); M[0, 1] = -M[1, 0]
        else:
            M[0, 0] += dp[0]; M[1, 0] += dp[1]; M[0, 1] += dp[2]; M[1, 1] += dp[3]; M[0, 2] += dp[4]; M[1, 2] += dp[5]
        trace.append((rho, M.copy()))
        it += 1
    return (rho, M, trace) if return_trace else (rho, M)


def warp_pos(pos, warp):
    """byte_tracker.py:653-657: float32 [2,3] @ [x, y, 1]."""
    p = np.array([pos[0], pos[1], 1.0], np.float32)
    return (np.asarray(warp, np.float32) @ p).astype(np.float32)
