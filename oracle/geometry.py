"""Oracle (CPU checker, test infrastructure only): proposal-side geometry in numpy.

Restates /root/reference/busca/tracking.py and the IoU cost of
/root/reference/adapters/ByteTrack/yolox/tracker/matching.py.  Integer / float64 work: results are
expected to match the HIP kernels bit-for-bit.
"""
import math

import numpy as np

F32_MIN = np.finfo("float32").min


# --------------------------------------------------------------------------------------------
# sentinels / distances   (busca/tracking.py:7-60)
# --------------------------------------------------------------------------------------------

def missing_candidate_bbox(seq_len=None, flavour="ltrb", pinned_numpy=True):
    """busca/tracking.py:7-20.  With the reference's pinned numpy 1.23.5 the array is float64
    (np.float32 / 100.0 -> float64 under legacy promotion); numpy >= 2 yields float32
    (SURVEY.md 7.3b).  `pinned_numpy` selects which environment is restated."""
    m = np.float64(F32_MIN) if pinned_numpy else np.float32(F32_MIN)
    hundred = 100.0 if pinned_numpy else np.float32(100.0)
    if flavour == "ltrb":
        vals = [m, m, m / hundred, m / hundred]
    elif flavour == "ltwh":
        vals = [m, m, -m / hundred, -m / hundred]
    else:
        raise ValueError("Unknown flavour: {}".format(flavour))
    bbox = np.array(vals, dtype=np.float64 if pinned_numpy else np.float32)
    if seq_len is not None:
        bbox = np.tile(bbox, (seq_len, 1))
    return bbox


def center_distance(atlbrs, btlbrs, weight_size=False):
    """busca/tracking.py:23-60 on ndarray inputs: float64 euclidean distance between box centres
    (scipy cdist == sqrt(dx*dx + dy*dy), no FMA), optional size-ratio weighting (:50-58)."""
    a = np.asarray(atlbrs, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(btlbrs, dtype=np.float64).reshape(-1, 4)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)), dtype=np.float64)
    ac = (a[:, :2] + a[:, 2:]) / 2.0
    bc = (b[:, :2] + b[:, 2:]) / 2.0
    dx = ac[:, None, 0] - bc[None, :, 0]
    dy = ac[:, None, 1] - bc[None, :, 1]
    dist = np.sqrt(dx * dx + dy * dy)
    if weight_size:
        asz = np.sqrt((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))
        bsz = np.sqrt((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
        w = np.maximum(asz[:, None] / bsz[None, :], bsz[None, :] / asz[:, None])
        dist = dist * w
    return dist


def iou_matrix(atlbrs, btlbrs):
    """`cython_bbox.bbox_overlaps(a, b)` as called from matching.py:53-70 (third-party, PARITY
    UNPINNED; published algorithm = Fast R-CNN bbox.pyx): float64, '+1' pixel-inclusive extents,
    ua = (area_a + area_b) - iw*ih, zero where the boxes do not overlap."""
    a = np.asarray(atlbrs, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(btlbrs, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((len(a), len(b)), dtype=np.float64)
    if out.size == 0:
        return out
    area_a = (a[:, 2] - a[:, 0] + 1) * (a[:, 3] - a[:, 1] + 1)
    area_b = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    iw = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]) + 1
    ih = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]) + 1
    ok = (iw > 0) & (ih > 0)
    inter = iw * ih
    ua = (area_a[:, None] + area_b[None, :]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(ok, inter / ua, 0.0)
    return out


def iou_distance(atlbrs, btlbrs):
    """matching.py:73-91: cost = 1 - IoU."""
    return 1 - iou_matrix(atlbrs, btlbrs)


def fuse_score(cost_matrix, det_scores):
    """matching.py:173-186: 1 - (1 - cost) * score[col]."""
    if cost_matrix.size == 0:
        return cost_matrix
    iou_sim = 1 - cost_matrix
    s = np.asarray(det_scores, dtype=np.float64)[None, :].repeat(cost_matrix.shape[0], axis=0)
    return 1 - iou_sim * s


def topk_rows(dists, P):
    """busca/network.py:333: per row `np.argsort(row)[:P]`.  numpy's default sort is unstable, so tie
    order is implementation-defined in the reference; the oracle (and the kernel) break ties by the
    lower column index.  Rows shorter than P are padded with -1 (the reference pads with None, :334-338)."""
    dists = np.asarray(dists, dtype=np.float64)
    B, N = dists.shape
    idx = np.full((B, P), -1, dtype=np.int32)
    if N:
        order = np.argsort(dists, axis=1, kind="stable")[:, :P]
        idx[:, : order.shape[1]] = order
    return idx


# --------------------------------------------------------------------------------------------
# crops   (busca/tracking.py:62-113)
# --------------------------------------------------------------------------------------------

def crop_geometry(im_h, im_w, bbox):
    """Integer geometry of `_cutout_with_pad` (tracking.py:80-100).
    Returns (y1, y2, x1, x2) unclipped, (cy1, cy2, cx1, cx2) clipped to the image."""
    x1, y1, x2, y2 = [float(v) for v in bbox]
    x1 = int(math.floor(x1)); y1 = int(math.floor(y1))
    x2 = int(math.ceil(x2)); y2 = int(math.ceil(y2))
    box = np.array([y1, y2, x1, x2])
    lim = np.array([im_h, im_h, im_w, im_w])
    clipped = np.clip(box, 0, lim)
    return box, clipped


def cutout_with_pad(im, bbox):
    """tracking.py:80-113: clip to the image, slice, pad back to the box extent with uint8(mean(crop))."""
    box, c = crop_geometry(im.shape[0], im.shape[1], bbox)
    crop = im[c[0]:c[1], c[2]:c[3]]
    pad = np.abs(c - box).astype(np.int32)
    if crop.size:
        fill = np.uint8(np.mean(crop))  # np.pad casts the float mean into the uint8 array (truncation)
    else:
        fill = np.uint8(0)
    crop = np.pad(crop, [[pad[0], pad[1]], [pad[2], pad[3]], [0, 0]], mode="constant", constant_values=fill)
    if crop.shape[0] == 0 or crop.shape[1] == 0:
        crop = np.zeros((1, 1, 3), dtype=im.dtype)
    return crop


_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def _linear_taps(ssize, dsize, clamp_x):
    """OpenCV resize.cpp tap computation for INTER_LINEAR (float32 fractional part, 11-bit taps)."""
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_x:
        lo = s < 0
        f[lo] = 0.0
        s[lo] = 0
        hi = s >= ssize - 1
        f[hi] = 0.0
        s[hi] = ssize - 1
    a0 = np.rint((np.float32(1.0) - f) * np.float32(_COEF_SCALE)).astype(np.int64)
    a1 = np.rint(f * np.float32(_COEF_SCALE)).astype(np.int64)
    return s, a0, a1


def resize_linear_u8(src, dw, dh):
    """`cv2.resize(src, (dw, dh), interpolation=cv2.INTER_LINEAR)` for uint8 HxWxC (tracking.py:71).
    Third-party arithmetic (opencv-python 4.7.0.72, requirements.txt:2), restated from OpenCV's
    resize.cpp: 11-bit fixed-point taps, horizontal pass in int32, vertical pass
    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2; exact 2x2 shrink takes the area-fast path; equal
    sizes copy.  PARITY UNPINNED (cv2 is not installable here); tolerance stated in tests: +-1 LSB."""
    src = np.ascontiguousarray(src)
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src.copy()
    if sw == 2 * dw and sh == 2 * dh:
        s = src.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, ax0, ax1 = _linear_taps(sw, dw, True)
    sy, by0, by1 = _linear_taps(sh, dh, False)
    s = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    hor = s[:, sx, :] * ax0[None, :, None] + s[:, sx1, :] * ax1[None, :, None]  # [sh, dw, C]
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    S0 = hor[y0]
    S1 = hor[y1]
    out = (((by0[:, None, None] * (S0 >> 4)) >> 16) + ((by1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


PIXEL_MEAN_BGR = np.array([0.406, 0.456, 0.485])
PIXEL_STD_BGR = np.array([0.225, 0.224, 0.299])  # 0.299 is the reference's "ghost" normalisation, tracking.py:63-65


def normalize_bgr(u8):
    """network.py:470-478 / tracking.py:73-76: float32 /255, then -= mean, /= std (float64 constants
    applied in place to the float32 array)."""
    x = u8.astype(np.float32) / 255.0
    x -= PIXEL_MEAN_BGR
    x /= PIXEL_STD_BGR
    return x


def get_bbox_crop(im, bbox, output_size=(128, 384), normalize=False):
    """tracking.py:62-78."""
    cut = cutout_with_pad(im, bbox)
    crop = resize_linear_u8(cut, output_size[0], output_size[1])
    return normalize_bgr(crop) if normalize else crop


def get_image_crops(im, bboxes, normalize=False):
    """network.py:492-507."""
    crops = [get_bbox_crop(im, b, normalize=normalize) for b in bboxes]
    if not crops:
        return np.zeros([0, 128, 384, 3])
    return np.stack(crops, axis=0)


def detection_coverage(frame_shape, boxes_tlbr_scaled):
    """adapters/ByteTrack/yolox/tracker/byte_tracker.py:574-623 restated with a boolean canvas instead of cv2.rectangle
    (third-party drawing, PARITY UNPINNED: filled rectangle = both int()-truncated corners inclusive, either corner
    order, clipped to the canvas).  Returns the same dict as the reference."""
    H, W = int(frame_shape[0]), int(frame_shape[1])
    canvas = np.zeros((H, W), bool)
    areas = []
    for bb in boxes_tlbr_scaled:
        x1, y1, x2, y2 = int(bb[0]), int(bb[1]), int(bb[2]), int(bb[3])
        xa, xb, ya, yb = min(x1, x2), max(x1, x2), min(y1, y2), max(y1, y2)
        canvas[max(ya, 0):max(yb + 1, 0), max(xa, 0):max(xb + 1, 0)] = True
        areas.append(max(min(((bb[2] - bb[0]) / H) * ((bb[3] - bb[1]) / W), 1.0), 0.0))
    pct = np.count_nonzero(canvas) / (H * W)
    n = len(areas)
    return {"area_covered": pct, "area_covered_per_obj": pct / n if n else 0.0, "max_bbox_area": max(areas) if areas else 0.0,
            "average_bbox_area": (np.sqrt(np.array(areas)).mean() ** 2) if n else 0.0, "bbox_areas": areas}
