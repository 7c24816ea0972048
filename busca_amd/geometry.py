"""Thin device-tensor wrappers over the geometry / crop entry points of libbusca_hip.so."""
import numpy as np
import torch

from . import _lib

_default_ctx = None


def default_context(device=0):
    """Process-wide Context (one per process/GPU, as the C-ABI prescribes)."""
    global _default_ctx
    if _default_ctx is None or _default_ctx.device != device:
        _default_ctx = _lib.Context(device)
    return _default_ctx


def _dev(ctx):
    return torch.device("cuda", ctx.device)


def _stream(ctx):
    return torch.cuda.current_stream(_dev(ctx)).cuda_stream


def _f64(x, ctx):
    if not torch.is_tensor(x):
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64)))
    return x.to(device=_dev(ctx), dtype=torch.float64).contiguous().view(-1, 4) if x.numel() else torch.zeros(0, 4, dtype=torch.float64, device=_dev(ctx))


def h2d_async(arr, dev):
    """Small host array -> device tensor without the host waiting for the stream: pinned staging (torch's caching host allocator keeps the block until the copy has
    run) + an asynchronous copy.  A plain `.to(dev)` of a pageable array is a SYNCHRONOUS copy on the current stream: enqueued behind a ReID pass it parks the host
    until the pass is done, and everything the host still has to enqueue (index gathers, the Decision-Transformer launch) then starts late."""
    t = arr if torch.is_tensor(arr) else torch.from_numpy(np.ascontiguousarray(arr))
    if t.device.type != "cpu":
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


def pairwise(ctx, a, b, mode, scores_b=None):
    """[nA,4],[nB,4] float64 ltrb -> float64 [nA,nB] on the GPU (mode: _lib.PAIR_*)."""
    a, b = _f64(a, ctx), _f64(b, ctx)
    out = torch.zeros(a.shape[0], b.shape[0], dtype=torch.float64, device=_dev(ctx))
    sc = None
    if scores_b is not None:
        sc = torch.as_tensor(np.asarray(scores_b, dtype=np.float64)).to(_dev(ctx)).contiguous()
    ctx.check(ctx.lib.busca_pairwise(ctx.h, a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], int(mode),
                                     _lib.ptr(sc), out.data_ptr(), _stream(ctx)))
    return out


def topk_rows(ctx, dist, P):
    """float64 [B,N] -> int32 [B,P] indices of the P smallest per row (ties: lower index; -1 padding)."""
    if not torch.is_tensor(dist):
        dist = torch.from_numpy(np.ascontiguousarray(np.asarray(dist, dtype=np.float64)))
    dist = dist.to(device=_dev(ctx), dtype=torch.float64).contiguous()
    B, N = dist.shape
    idx = torch.empty(B, P, dtype=torch.int32, device=_dev(ctx))
    ctx.check(ctx.lib.busca_topk_rows(ctx.h, dist.data_ptr(), B, N, P, idx.data_ptr(), _stream(ctx)))
    return idx


def pairwise_host(ctx, a, b, mode, scores_b=None):
    """Host boxes in, host matrix out: [nA,4],[nB,4] float64 ltrb (numpy) -> float64 [nA,nB] (numpy), for the per-frame calls of a tracker
    (busca/tracking.py:23-60 center_distance, matching.py:53-91 iou_distance) whose operands are a few KB.  Boxes, scores and the result live in ONE
    pinned host table the GPU maps: the kernel stages the boxes into LDS straight from it and writes the matrix straight into it, so the call is one
    launch + one stream synchronisation - no H2D copies, no zero-fill kernel, no D2H copy (each ~20-70 us of host time on this stack, five of them
    around a 4 us kernel were the 0.18-0.31 ms of round 5)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1, 4)
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64)).reshape(-1, 4)
    nA, nB = a.shape[0], b.shape[0]
    if nA == 0 or nB == 0:
        return np.zeros((nA, nB), dtype=np.float64)
    ns = nB if scores_b is not None else 0
    table = torch.empty(4 * (nA + nB) + ns + nA * nB, dtype=torch.float64, pin_memory=True)
    tab = table.numpy()
    tab[:4 * nA] = a.reshape(-1)
    tab[4 * nA:4 * (nA + nB)] = b.reshape(-1)
    if ns:
        tab[4 * (nA + nB):4 * (nA + nB) + ns] = np.asarray(scores_b, dtype=np.float64).reshape(-1)
    base, o = table.data_ptr(), 4 * (nA + nB) + ns
    ctx.check(ctx.lib.busca_pairwise(ctx.h, base, nA, base + 32 * nA, nB, int(mode), (base + 32 * (nA + nB)) if ns else None,
                                     base + 8 * o, _stream(ctx)))
    torch.cuda.current_stream(_dev(ctx)).synchronize()
    return tab[o:].reshape(nA, nB).copy()


def topk_rows_host(ctx, dist, P):
    """Host float64 [B,N] in, host int32 [B,P] out (same kernel as topk_rows; rows and indices in one pinned table the GPU maps - one launch + one
    stream synchronisation, no copies): the route of BUSCA.associate_embeddings, whose distance matrix arrives as a numpy array (network.py:333-338)."""
    d = np.ascontiguousarray(np.asarray(dist, dtype=np.float64))
    B, N = d.shape
    P = int(P)
    if B == 0:
        return np.zeros((0, P), np.int32)
    table = torch.empty(B * N + (B * P + 1) // 2, dtype=torch.float64, pin_memory=True)
    tab = table.numpy()
    tab[:B * N] = d.reshape(-1)
    base = table.data_ptr()
    ctx.check(ctx.lib.busca_topk_rows(ctx.h, base, B, N, P, base + 8 * B * N, _stream(ctx)))
    torch.cuda.current_stream(_dev(ctx)).synchronize()
    return tab[B * N:].view(np.int32)[:B * P].reshape(B, P).copy()


def crop_pool(ctx):
    """The context's device crop pool (busca_amd/crop_pool.py), created on first use."""
    if getattr(ctx, "_crop_pool", None) is None:
        from .crop_pool import CropPool
        ctx._crop_pool = CropPool(ctx.device)
    return ctx._crop_pool


def crop_gather(ctx, frame, boxes, want_u8=True, want_f16=False, dst_ptrs=None):
    """frame: u8 [H,W,3] BGR (numpy or cuda tensor); boxes [n,4] x1y1x2y2 (float: rounded to extents in float64 on the
    host; int32: already extents) -> (u8 [n,384,128,3] | None, fp16 [n,384,128,4] RGB0 normalised | None) on the GPU.
    `dst_ptrs` (uint64 [n]): write crop i to that device address instead (crop-pool slots)."""
    from .tracking import box_extents
    dev = _dev(ctx)
    if torch.is_tensor(boxes):
        boxes = boxes.detach().cpu().numpy()
    boxes = np.asarray(boxes)
    rects = boxes.reshape(-1, 4) if boxes.dtype == np.int32 else box_extents(boxes)
    frame, rects = _frame_for_rects(ctx, frame, rects, dev)
    assert frame.dtype == torch.uint8 and frame.dim() == 3 and frame.shape[2] == 3
    n = rects.shape[0]
    # The box extents (4 x int32 per box) and the destination slots (1 x int64) are NOT copied to the device: they are written into PINNED host
    # memory, which the GPU maps at the same address, and the kernel reads its few KB straight from there (an asynchronous 3 KB copy costs
    # ~70 us of host time per call on this stack, two device-side repack kernels another ~20 us).  The table stays referenced (with an event
    # behind the launch) until 32 later calls have been issued and its own launch has completed.
    H, W = frame.shape[:2]
    u8 = torch.empty(n, 384, 128, 3, dtype=torch.uint8, device=dev) if want_u8 else None
    f16 = torch.empty(n, 384, 128, 4, dtype=torch.float16, device=dev) if want_f16 else None
    if n:
        table = torch.empty(n * 3, dtype=torch.int64, pin_memory=True)           # [n][2] int64 = [n][4] int32 extents, then [n] int64 slot addresses
        tab = table.numpy()
        tab[:2 * n] = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1).view(np.int64)
        if dst_ptrs is not None:
            tab[2 * n:] = np.ascontiguousarray(dst_ptrs, dtype=np.uint64).view(np.int64)
        base = table.data_ptr()
        ctx.check(ctx.lib.busca_crop_gather_ex(ctx.h, frame.data_ptr(), H, W, frame.stride(0), base, n,
                                               (base + 16 * n) if dst_ptrs is not None else None, _lib.ptr(u8), _lib.ptr(f16), _stream(ctx)))
        _keep_until_done(ctx, table, dev)
    return u8, f16


def crop_gather_sized(ctx, frame, boxes, out_w, out_h):
    """Crops of an arbitrary output size (busca_crop_gather_sized): u8 [n, out_h, out_w, 3] on the GPU."""
    from .tracking import box_extents
    dev = _dev(ctx)
    boxes = np.asarray(boxes.detach().cpu().numpy() if torch.is_tensor(boxes) else boxes)
    rects = boxes.reshape(-1, 4) if boxes.dtype == np.int32 else box_extents(boxes)
    frame, rects = _frame_for_rects(ctx, frame, rects, dev)
    assert frame.dtype == torch.uint8 and frame.dim() == 3 and frame.shape[2] == 3
    n = rects.shape[0]
    out = torch.empty(n, int(out_h), int(out_w), 3, dtype=torch.uint8, device=dev)
    if n:
        table = torch.empty(n * 2, dtype=torch.int64, pin_memory=True)
        table.numpy()[:] = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1).view(np.int64)
        H, W = frame.shape[:2]
        ctx.check(ctx.lib.busca_crop_gather_sized(ctx.h, frame.data_ptr(), H, W, frame.stride(0), table.data_ptr(), n, int(out_h), int(out_w), out.data_ptr(), _stream(ctx)))
        _keep_until_done(ctx, table, dev)
    return out


def _keep_until_done(ctx, host_tensor, dev, depth=32):
    """Keep a pinned host tensor a launched kernel reads alive: torch's caching host allocator would hand the block out again as soon as the
    last Python reference dies, and it only knows about uses by its own copies."""
    from collections import deque
    q = getattr(ctx, "_pinned_in_flight", None)
    if q is None:
        q = ctx._pinned_in_flight = deque()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    q.append((host_tensor, ev))
    while len(q) > depth:
        _, old = q.popleft()
        old.synchronize()                                                        # long complete in practice: one query


def begin_frame(ctx, frame):
    """Upload `frame` (u8 [H,W,3] host array) ONCE for every crop call of one tracker update.  Adapters cut crops from the same host frame several
    times per update (detections of both confidence bands, Kalman boxes - adapters/ByteTrack/yolox/tracker/byte_tracker.py:280-282; StrongSORT once
    per detection, deep_sort/tracker.py:126).  The scope is EXPLICIT: between begin_frame and end_frame a crop call whose `image` IS this object
    (identity; the scope holds a reference, so the id cannot be recycled) reuses the upload, and the caller promises not to edit the array inside the
    scope.  Without a scope every call uploads the array it is given - what busca/network.py:492-507 does with the live array - so no result ever
    depends on a guess about whether two host buffers hold the same pixels (rounds 2-4 guessed with a sparse pixel fingerprint: removed)."""
    dev = _dev(ctx)
    if torch.is_tensor(frame):
        t = frame.to(dev).contiguous()                    # (a CUDA tensor is used where it is; a CPU tensor is uploaded once)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(frame))).to(dev)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    stack = getattr(ctx, "_frame_stack", None)
    if stack is None:
        stack = ctx._frame_stack = []
    stack.append(getattr(ctx, "_frame_scope", None))        # scopes nest: leaving an inner scope restores the outer one
    ctx._frame_scope = (frame, t, ev, torch.cuda.current_stream(dev))
    return t


def end_frame(ctx):
    stack = getattr(ctx, "_frame_stack", None)
    ctx._frame_scope = stack.pop() if stack else None


class frame_scope:
    """`with geometry.frame_scope(ctx, frame): ...` = begin_frame / end_frame."""

    def __init__(self, ctx, frame):
        self.ctx, self.frame = ctx, frame

    def __enter__(self):
        begin_frame(self.ctx, self.frame)
        return self

    def __exit__(self, *exc):
        end_frame(self.ctx)
        return False


def _frame_on_device(ctx, frame, dev):
    """The frame as a contiguous cuda u8 tensor: the scope's upload when `frame` is the scoped object, else a fresh upload of the live array."""
    scope = getattr(ctx, "_frame_scope", None)
    if scope is not None and scope[0] is frame:
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(scope[2])                                     # the upload may have been enqueued on another stream ...
        if cur != scope[3]:
            scope[1].record_stream(cur)                              # ... whose allocator must not hand the block out again while this stream's crop kernel reads it
        return scope[1]
    if torch.is_tensor(frame):
        return frame.to(dev).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(frame))).to(dev)


def _frame_for_rects(ctx, frame, rects, dev):
    """-> (cuda u8 frame tensor, the box extents relative to it).  A host frame outside a frame scope is read LIVE on every call (what busca/network.py:492-507
    does), but only the part the boxes need goes over PCIe: `_cutout_with_pad` (busca/tracking.py:80-113) looks at nothing but the clipped box - slice, mean of
    the slice, padding back to the box extent - so cutting from the sub-frame spanned by the union of the clipped boxes, with the extents shifted to its origin,
    gives the same bytes (a box that crosses a frame edge makes that edge the sub-frame's edge; a box wholly outside the frame is wholly outside the
    sub-frame).  Taken when the union covers at most a quarter of the frame: the unchanged StrongSORT adapter calls get_image_crops once per detection
    (deep_sort/tracker.py:126,273,291) and uploaded the 6 MB frame each time (0.16-0.25 ms); one pedestrian's box is ~50 KB."""
    rects = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4)
    scope = getattr(ctx, "_frame_scope", None)
    if torch.is_tensor(frame) or (scope is not None and scope[0] is frame) or len(rects) == 0:
        return _frame_on_device(ctx, frame, dev), rects
    arr = np.asarray(frame)
    if arr.ndim == 3 and arr.dtype == np.uint8 and arr.shape[2] == 3:
        H, W = arr.shape[:2]
        r = rects.astype(np.int64)
        cx1, cy1, cx2, cy2 = np.clip(r[:, 0], 0, W), np.clip(r[:, 1], 0, H), np.clip(r[:, 2], 0, W), np.clip(r[:, 3], 0, H)
        ok = (cx2 > cx1) & (cy2 > cy1)
        if ok.any():
            x0, y0, x1, y1 = int(cx1[ok].min()), int(cy1[ok].min()), int(cx2[ok].max()), int(cy2[ok].max())
            if 4 * (x1 - x0) * (y1 - y0) <= H * W:
                sub = torch.empty((y1 - y0, x1 - x0, 3), dtype=torch.uint8, pin_memory=True)
                np.copyto(sub.numpy(), arr[y0:y1, x0:x1])
                t = sub.to(dev, non_blocking=True)
                _keep_until_done(ctx, sub, dev)
                return t, (r - np.array([x0, y0, x0, y0], np.int64)).astype(np.int32)
    return _frame_on_device(ctx, frame, dev), rects


def gather_crops(ctx, src_ptrs):
    """uint64 [n] device addresses of 147 456-byte crops (0 = all-zero crop) -> cuda u8 [n,384,128,3] (busca_gather_crops)."""
    dev = _dev(ctx)
    n = len(src_ptrs)
    out = torch.empty(n, 384, 128, 3, dtype=torch.uint8, device=dev)
    if n:
        stage = torch.empty(n, dtype=torch.int64, pin_memory=True)                # pinned staging: the upload is asynchronous
        stage.numpy()[:] = np.ascontiguousarray(src_ptrs, dtype=np.uint64).view(np.int64)
        src = stage.to(dev, non_blocking=True)
        ctx.check(ctx.lib.busca_gather_crops(ctx.h, src.data_ptr(), n, out.data_ptr(), _stream(ctx)))
    return out
