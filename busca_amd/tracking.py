"""Proposal-side geometry with the reference's names and semantics (busca/tracking.py), computed by the
HIP kernels behind the C-ABI (busca_pairwise / busca_crop_gather).  Inputs and outputs are host numpy
arrays exactly like the reference; there is no CPU implementation here."""
import os

import numpy as np
import torch

from . import _lib, geometry
from .crop_pool import Slot

_F32_MIN = np.finfo("float32").min


def missing_candidate_bbox(seq_len=None, flavour="ltrb", pinned_numpy=True):
    """Sentinel box of a padded candidate (busca/tracking.py:7-20).  `pinned_numpy=True` reproduces the
    reference's pinned numpy 1.23.5, where np.float32 / 100.0 is float64 and the array therefore float64."""
    dt = np.float64 if pinned_numpy else np.float32
    m = dt(_F32_MIN)
    if flavour == "ltrb":
        bbox = np.array([m, m, m / dt(100.0), m / dt(100.0)], dtype=dt)
    elif flavour == "ltwh":
        bbox = np.array([m, m, -m / dt(100.0), -m / dt(100.0)], dtype=dt)
    else:
        raise ValueError("Unknown flavour: {}".format(flavour))
    if seq_len is not None:
        bbox = np.tile(bbox, (seq_len, 1))
    return bbox


def _tlbrs(items):
    if len(items) > 0 and isinstance(items[0], np.ndarray):
        return np.asarray(items, dtype=np.float64).reshape(-1, 4)
    return np.array([t.tlbr for t in items], dtype=np.float64).reshape(-1, 4)


def center_distance(atracks, btracks, weight_size=False, ctx=None):
    """Centre-to-centre distance matrix [len(a), len(b)] float64 (busca/tracking.py:23-60); accepts track
    objects (`.tlbr`) or ndarrays.  Empty input -> zeros (the reference's np.float alias is float64)."""
    a, b = _tlbrs(atracks), _tlbrs(btracks)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(atracks), len(btracks)), dtype=np.float64)
    ctx = ctx or geometry.default_context()
    mode = _lib.PAIR_CENTER_WEIGHTED if weight_size else _lib.PAIR_CENTER
    return geometry.pairwise_host(ctx, a, b, mode)


def iou_distance(atlbrs, btlbrs, det_scores=None, ctx=None):
    """1 - IoU cost matrix with the '+1' pixel convention of cython_bbox (adapters/ByteTrack/yolox/tracker/
    matching.py:53-91); with `det_scores` also applies fuse_score (:173-186)."""
    a, b = _tlbrs(atlbrs), _tlbrs(btlbrs)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)), dtype=np.float64)
    ctx = ctx or geometry.default_context()
    return geometry.pairwise_host(ctx, a, b, _lib.PAIR_IOU_COST, scores_b=det_scores)

TRACKED = 1     # TrackState.Tracked (adapters/*/mot_online/basetrack.py:5-9)


def multi_predict(stracks, ctx=None):
    """STrack.multi_predict (adapters/ByteTrack/yolox/tracker/byte_tracker.py:50-61) on the GPU: constant-velocity Kalman
    prediction of every track's (mean [8], covariance [8,8]) in place; `mean[7]` of non-Tracked tracks is zeroed first."""
    if len(stracks) == 0:
        return
    ctx = ctx or geometry.default_context()
    dev = torch.device("cuda", ctx.device)
    mean = torch.from_numpy(np.asarray([st.mean for st in stracks], dtype=np.float64)).to(dev)
    cov = torch.from_numpy(np.asarray([st.covariance for st in stracks], dtype=np.float64)).to(dev)
    nt = torch.from_numpy(np.asarray([st.state != TRACKED for st in stracks], dtype=np.uint8)).to(dev)
    ctx.check(ctx.lib.busca_kalman_multi_predict(ctx.h, mean.data_ptr(), cov.data_ptr(), nt.data_ptr(), len(stracks),
                                                 torch.cuda.current_stream(dev).cuda_stream))
    both = torch.cat([mean, cov.view(len(stracks), 64)], 1).cpu().numpy()       # one device->host copy for both
    mean, cov = np.ascontiguousarray(both[:, :8]), np.ascontiguousarray(both[:, 8:]).reshape(-1, 8, 8)
    for i, st in enumerate(stracks):
        st.mean = mean[i]
        st.covariance = cov[i]


def remove_duplicate_stracks(stracksa, stracksb, ctx=None, thresh=0.15):
    """remove_duplicate_stracks (byte_tracker.py:685-698): of two tracks whose IoU cost is below 0.15 the one alive for
    fewer frames goes (ties: the one of the first list).  IoU cost and the marking both run on the GPU."""
    if len(stracksa) == 0 or len(stracksb) == 0:
        return list(stracksa), list(stracksb)
    ctx = ctx or geometry.default_context()
    dev = torch.device("cuda", ctx.device)
    cost = geometry.pairwise(ctx, _tlbrs(stracksa), _tlbrs(stracksb), _lib.PAIR_IOU_COST)
    age_a = torch.tensor([t.frame_id - t.start_frame for t in stracksa], dtype=torch.int32, device=dev)
    age_b = torch.tensor([t.frame_id - t.start_frame for t in stracksb], dtype=torch.int32, device=dev)
    keep_a = torch.empty(len(stracksa), dtype=torch.uint8, device=dev)
    keep_b = torch.empty(len(stracksb), dtype=torch.uint8, device=dev)
    ctx.check(ctx.lib.busca_duplicate_masks(ctx.h, cost.data_ptr(), len(stracksa), len(stracksb), age_a.data_ptr(), age_b.data_ptr(),
                                            float(thresh), keep_a.data_ptr(), keep_b.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    kab = torch.cat([keep_a, keep_b]).cpu().numpy()                              # one device->host copy for both masks
    ka, kb = kab[:len(stracksa)], kab[len(stracksa):]
    return [t for i, t in enumerate(stracksa) if ka[i]], [t for i, t in enumerate(stracksb) if kb[i]]


_PIXEL_MEAN = np.array([0.406, 0.456, 0.485])   # BGR
_PIXEL_STD = np.array([0.225, 0.224, 0.299])    # BGR; 0.299 is the reference's "ghost" normalisation


def normalize_crops(u8):
    """(x/255 - mean)/std in BGR order, float32 (busca/network.py:470-478)."""
    x = np.asarray(u8).astype(np.float32) / 255.0
    x -= _PIXEL_MEAN
    x /= _PIXEL_STD
    return x


class DeviceBackedCrops(np.ndarray):
    """Host uint8 crops that remember their device-resident twins (device-resident track memory).

    `get_image_crops(..., normalize=False)` returns this ndarray subclass: to the trackers it is an ordinary uint8
    array ([N,384,128,3] with REAL host bytes; `crops[i]` is what they append to `images_mem`), but `crops[i]` also
    carries `.slot`, its slot in the bounded device pool (busca_amd/crop_pool.py).  `associate_embeddings` gathers
    slots on the GPU and skips the per-frame host->device copy of B*(L+P) crops (147 KB each).  Anything that loses
    the slot (np.array(...) copies, pickling, arithmetic) is still a correct host crop and takes the host path."""

    def __new__(cls, host, slots):
        host = np.asarray(host)
        obj = host.view(cls)
        obj._slots = slots
        obj._host = host                    # the plain array: slot.host views must not reference this subclass (no cycles)
        obj.slot = None
        return obj

    def __array_finalize__(self, obj):
        self._slots = None                  # generic views/copies do not know which pool slots they cover
        self._host = None
        self.slot = None

    def __getitem__(self, key):
        if self._slots is not None and self.ndim == 4 and isinstance(key, (int, np.integer)):
            # the crop is a view of the PLAIN host array: it keeps the frame's host bytes alive (as the reference's
            # np.stack'ed crops do) but not this container, so the other crops' pool slots are free to go
            k = int(key)
            out = self._host[k].view(DeviceBackedCrops)
            out.slot = self._slots[k]
            if out.slot.host is None:
                out.slot.host = self._host[k]             # a spill of this crop needs no device->host copy
            return out
        return super().__getitem__(key)

    def __reduce__(self):                   # pickling / copy.deepcopy: a plain host array (the slot stays with this process)
        return np.asarray(self).__reduce__()

    @property
    def dev(self):
        """cuda u8 view of this crop's slot ([384,128,3]); None for non-crop views and after a spill."""
        return self.slot.tensor() if self.slot is not None else None


class FrameHostCopy:
    """The host bytes of one get_image_crops call, ON DEMAND (round 5): the crops live in pool slots; nothing is copied to the host until somebody
    reads a pixel.  The first host read of ANY crop of the call fetches the WHOLE batch in one go - one index-gather launch over the call's slots
    (busca_gather_crops) and one device->host copy - so a tracker that looks at every crop pays one transfer per call, and one that never looks
    (the five adapters only store the crops and hand them back to associate_embeddings) pays nothing: no pinned buffer, no copy, no side stream
    (rounds 3-4 enqueued a 17 MB pinned copy per call whether or not anyone read it: 0.12 ms of host time per call).
    A slot that was released and reused before the read shows another crop's bytes in its row - a row nobody can ask for any more."""
    __slots__ = ("ctx", "ptrs", "event", "_np", "expired")

    def __init__(self, ctx=None, ptrs=None, event=None):
        self.ctx, self.ptrs, self.event, self._np, self.expired = ctx, ptrs, event, None, False

    def rows(self):
        """uint8 [n,384,128,3] host view of the batch (fetched on the first call)."""
        if self._np is None:
            dev = torch.device("cuda", self.ctx.device)
            if self.event is not None:
                torch.cuda.current_stream(dev).wait_event(self.event)      # the crop kernel may have run on another stream
            self._np = geometry.gather_crops(self.ctx, self.ptrs).cpu().numpy()
            self.event = None
        return self._np


DeviceCrop = Slot                       # one object per crop: the pool slot IS what the tracker stores (busca_amd/crop_pool.py)


class DeviceCrops:
    """Sequence of DeviceCrop ([N,384,128,3] uint8 to duck-typing callers); `crops[i]` is what a tracker stores."""
    dtype, ndim = np.dtype(np.uint8), 4

    def __init__(self, slots):
        self._items = list(slots)

    @property
    def shape(self):
        return (len(self._items), 384, 128, 3)

    def __len__(self):
        return len(self._items)

    def __iter__(self):
        return iter(self._items)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self._items[int(key)]
        return np.asarray(self)[key]

    def __array__(self, dtype=None, copy=None):
        a = np.stack([np.asarray(c) for c in self._items]) if self._items else np.zeros((0, 384, 128, 3), np.uint8)
        return a if dtype is None else a.astype(dtype)


def box_extents(bboxes):
    """[n,4] x1y1x2y2 -> int32 (floor(x1), floor(y1), ceil(x2), ceil(y2)) on the caller's float64 values, as
    busca/tracking.py:84-87 does with math.floor / math.ceil (a float32 copy of the box can land on the other side of
    an integer).  Clamped to +-2^30 so absurd boxes cannot overflow the kernel's int arithmetic."""
    b = np.nan_to_num(np.asarray(bboxes, dtype=np.float64).reshape(-1, 4), nan=0.0, posinf=2.0 ** 30, neginf=-2.0 ** 30)
    r = np.empty(b.shape, np.float64)
    r[:, :2] = np.floor(b[:, :2])
    r[:, 2:] = np.ceil(b[:, 2:])
    return np.clip(r, -2.0 ** 30, 2.0 ** 30).astype(np.int32)


def get_image_crops(im, bboxes, normalize=True, ctx=None, device_only=False, host_copy=None, output_size=None):
    """All crops of one frame in one launch: u8 BGR [N,384,128,3] (float32 normalised if `normalize`).
    With normalize=False every crop is written into a slot of the device crop pool and the returned crops remember
    their slot.  `host_copy` says what happens to the HOST bytes of those crops (147 KB each - the bulk of this call when it
    is waited for; busca/network.py:492-507 returns host arrays):
      "lazy"  (default) nothing is copied until somebody reads a pixel: the returned `DeviceCrops` fetch the whole batch with one
              gather + one device->host copy on the first host read of any of its crops;
      "eager" wait for it: a real uint8 ndarray (`DeviceBackedCrops`) - for callers that need ndarray instances;
      "never" (`device_only=True`) no copy at all; a host read copies that one crop back synchronously."""
    rects = box_extents(bboxes)
    sized = output_size is not None and tuple(int(v) for v in output_size) != (128, 384)
    if len(rects) == 0:
        return np.zeros([0, int(output_size[0]), int(output_size[1]), 3]) if sized else np.zeros([0, 128, 384, 3])     # the reference's (transposed) empty shape, network.py:503
    ctx = ctx or geometry.default_context()
    if sized:
        # `output_size` = (width, height) as cv2.resize takes it (tracking.py:71): plain host arrays [N, height, width, 3], no pool slots -
        # only the 384 x 128 crops are ReID inputs
        u8 = geometry.crop_gather_sized(ctx, im, rects, int(output_size[0]), int(output_size[1])).cpu().numpy()
        return normalize_crops(u8) if normalize else u8
    if normalize:
        u8, _ = geometry.crop_gather(ctx, im, rects, want_u8=True)
        return normalize_crops(u8.cpu().numpy())
    if host_copy is None:
        host_copy = "never" if device_only else "lazy"
    if host_copy not in ("lazy", "eager", "never"):
        raise ValueError("host_copy must be 'lazy', 'eager' or 'never', not %r" % (host_copy,))
    pool = geometry.crop_pool(ctx)
    frame = FrameHostCopy() if host_copy == "lazy" else None
    slots = pool.alloc(len(rects), frame)
    ptrs = np.fromiter((s.ptr for s in slots), dtype=np.uint64, count=len(slots))
    if host_copy == "never":
        geometry.crop_gather(ctx, im, rects, want_u8=False, dst_ptrs=ptrs)
        return DeviceCrops(slots)
    if host_copy == "eager":
        packed, _ = geometry.crop_gather(ctx, im, rects, want_u8=True, dst_ptrs=ptrs)    # the same launch also writes the batch as one contiguous buffer (the slots need not be adjacent)
        return DeviceBackedCrops(packed.cpu().numpy(), slots)
    # lazy: pool slots only; the host bytes are fetched by the first host read (FrameHostCopy)
    geometry.crop_gather(ctx, im, rects, want_u8=False, dst_ptrs=ptrs)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(torch.device("cuda", ctx.device)))
    frame.ctx, frame.ptrs, frame.event = ctx, ptrs, ev
    return DeviceCrops(slots)


def get_bbox_crop(im, bbox_real_scale, output_size=(128, 384), normalize=True, ghost_normalize=True, ctx=None):
    """Single crop (busca/tracking.py:62-78); `output_size` = (width, height) as cv2.resize takes it."""
    crop = np.asarray(get_image_crops(im, [bbox_real_scale], normalize=False, ctx=ctx, output_size=output_size)[0])
    if normalize:
        crop = crop.astype(np.float32) / 255.0
        crop -= _PIXEL_MEAN
        crop /= (_PIXEL_STD if ghost_normalize else np.array([0.225, 0.224, 0.229]))
    return crop


def get_detection_coverage(frame_shape, active_stracks, inactive_stracks=(), ctx=None):
    """Share of the frame covered by the tracks' boxes and the per-object area statistics of the reliability gate
    (adapters/ByteTrack/yolox/tracker/byte_tracker.py:574-623) - same dict as the reference; the pixel count comes
    from busca_coverage instead of drawing filled rectangles on an H x W x 3 canvas.  `frame_shape` = frame.shape.
    The reference's area normalisation divides the box WIDTH by the frame height and the box HEIGHT by the frame
    width (:589); that is reproduced as is."""
    H, W = int(frame_shape[0]), int(frame_shape[1])
    rects, areas = [], []
    for track in list(active_stracks) + list(inactive_stracks):
        bb = np.array(track.tlbr) * track.scale
        x1, y1, x2, y2 = int(bb[0]), int(bb[1]), int(bb[2]), int(bb[3])          # int() truncates toward zero
        xa, xb, ya, yb = min(x1, x2), max(x1, x2), min(y1, y2), max(y1, y2)
        if xb >= 0 and yb >= 0 and xa <= W - 1 and ya <= H - 1:                  # clipped rectangle is not empty
            rects.append([max(xa, 0), max(ya, 0), min(xb, W - 1), min(yb, H - 1)])
        areas.append(max(min(((bb[2] - bb[0]) / H) * ((bb[3] - bb[1]) / W), 1.0), 0.0))
    n_obj = len(areas)
    covered = 0
    if rects:
        ctx = ctx or geometry.default_context()
        dev = torch.device("cuda", ctx.device)
        r = torch.tensor(rects, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.busca_coverage(ctx.h, r.data_ptr(), len(rects), H, W, cnt.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        covered = int(cnt.item())
    pct = covered / (H * W)
    if n_obj > 0:
        avg_cov = pct / n_obj
        avg_area = np.sqrt(np.array(areas)).mean() ** 2
    else:
        avg_cov, avg_area = 0.0, 0.0
    return {"area_covered": pct, "area_covered_per_obj": avg_cov, "max_bbox_area": max(areas) if areas else 0.0,
            "average_bbox_area": avg_area, "bbox_areas": areas}


def is_reliable(frame_shape, active_stracks, p, ctx=None):
    """byte_tracker.py:459-465: the frame is 'reliable' when the covered area exceeds p[0] * per-object area + p[1]."""
    cov = get_detection_coverage(frame_shape, active_stracks, (), ctx=ctx)
    return bool(cov["area_covered"] > cov["area_covered_per_obj"] * p[0] + p[1])


def find_transform_ecc(prev_frame, cur_frame, warp_matrix=None, motion="MOTION_EUCLIDEAN", number_of_iterations=100,
                       termination_eps=1e-5, ctx=None):
    """cv2.cvtColor(BGR2GRAY) + cv2.findTransformECC(templateImage=prev, inputImage=cur, ...) on the GPU (busca_ecc_align;
    third-party OpenCV arithmetic restated, see oracle/ecc.py).  Frames: u8 BGR [H,W,3] (numpy or cuda tensors).
    Returns (cc, warp float32 [2,3]); raises BuscaError where OpenCV raises (no convergence / NaN)."""
    import ctypes as C
    ctx = ctx or geometry.default_context()
    dev = torch.device("cuda", ctx.device)
    if motion not in ("MOTION_EUCLIDEAN", "MOTION_AFFINE"):
        raise ValueError("Invalid warp_mode: {}".format(motion))
    fr = []
    for f in (prev_frame, cur_frame):
        if not torch.is_tensor(f):
            f = torch.from_numpy(np.ascontiguousarray(f))
        f = f.to(dev).contiguous()
        assert f.dtype == torch.uint8 and f.dim() == 3 and f.shape[2] == 3
        fr.append(f)
    assert fr[0].shape == fr[1].shape
    H, W = fr[0].shape[:2]
    warp = np.eye(2, 3, dtype=np.float32) if warp_matrix is None else np.ascontiguousarray(warp_matrix, dtype=np.float32).reshape(2, 3).copy()
    cc, iters = C.c_double(0.0), C.c_int32(0)
    ctx.check(ctx.lib.busca_ecc_align(ctx.h, fr[0].data_ptr(), fr[1].data_ptr(), H, W, fr[0].stride(0), fr[1].stride(0),
                                      0 if motion == "MOTION_EUCLIDEAN" else 1, int(number_of_iterations), float(termination_eps),
                                      warp.ctypes.data, C.byref(cc), C.byref(iters), torch.cuda.current_stream(dev).cuda_stream))
    find_transform_ecc.last_iterations = iters.value
    return cc.value, warp


def warp_pos(pos, warp_matrix):
    """BYTETracker.warp_pos (byte_tracker.py:653-657): float32 [2,3] @ [x, y, 1]."""
    p = np.array([pos[0], pos[1], 1.0], dtype=np.float32)
    return (np.asarray(warp_matrix, dtype=np.float32).reshape(2, 3) @ p).astype(np.float32)


def camera_motion_compensation(track_pool, last_image, current_frame, frame_id=2, number_of_iterations=100, termination_eps=0.00001,
                               warp_mode="MOTION_EUCLIDEAN", ctx=None):
    """BYTETracker.camera_motion_compensation (byte_tracker.py:626-650): estimate the previous->current frame warp and move every
    track of `track_pool` by it (`STrack.apply_camera_motion`, :123-137: position (mean[:2] or _tlwh[:2]) * scale -> warp ->
    / scale).  Returns the correlation coefficient (1.0 on the first frame, as the reference does)."""
    cc = 1.0
    if frame_id > 1 and last_image is not None:
        cc, warp = find_transform_ecc(last_image, current_frame, None, warp_mode, number_of_iterations, termination_eps, ctx=ctx)
        for t in track_pool:
            if hasattr(t, "apply_camera_motion"):
                t.apply_camera_motion(warp)
                continue
            holder = t.mean if getattr(t, "mean", None) is not None else t._tlwh
            new_pos = warp_pos(np.asarray(holder[:2], dtype=np.float64) * t.scale, warp) / t.scale
            holder[:2] = new_pos
    return cc


def recover_with_busca(probs_matrix, reliable, n_dets, busca_thresh):
    """Caller-side decision rule shared by the adapters (byte_tracker.py:504-527, StrongSORT tracker.py:347-371,
    GHOST tracker.py:776-800): lost track i is recovered at its own Kalman prediction iff its memory is reliable
    and probs[i, n_dets + i] > busca_thresh.  Returns (matches [[i, prob], ...], unmatched track indices)."""
    matches, unmatched = [], []
    if probs_matrix is None:
        return matches, unmatched
    for i in range(probs_matrix.shape[0]):
        pr = probs_matrix[i, n_dets + i]
        if reliable[i] and pr > busca_thresh:
            matches.append([i, pr])
        else:
            unmatched.append(i)
    return matches, unmatched
