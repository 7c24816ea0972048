"""Device-resident track memory: a bounded pool of crop slots in HBM.

The reference keeps every crop a track has ever produced in per-track Python lists of host arrays
(`STrack.images_mem`, adapters/ByteTrack/yolox/tracker/byte_tracker.py:40-42,90-94,117-121; never trimmed - GHOST
works around the growth with `avoid_memory_leak`, adapters/GHOST/src/tracker.py:249-258) and ships B*(L+P) of them to
the device every frame.  Here `get_image_crops` writes each crop straight into a slot of this pool and hands the tracker
an object that remembers its slot; `associate_embeddings` then gathers slots with one index-gather kernel
(busca_gather_crops).

Lifetime and bound:
  * a slot belongs to exactly one crop; it returns to the free list when the LAST Python reference to that crop dies
    (track removed, list trimmed) - eviction follows track lifetime, no bookkeeping in the tracker;
  * the pool grows slab by slab up to `budget_bytes` (BUSCA_CROP_POOL_MB, default 16 GiB of the 288 GB); beyond that the
    OLDEST live crops are spilled: their bytes move to host memory (already there unless `device_only_crops`) and their
    slot is reused.  A spilled crop keeps working through the host path of `associate_embeddings` (bit-identical).
HBM use is therefore flat once the budget is reached, whatever the sequence length.
"""
import os
import weakref

import numpy as np
import torch

CROP_H, CROP_W = 384, 128
CROP_BYTES = CROP_H * CROP_W * 3


class Slot(np.lib.mixins.NDArrayOperatorsMixin):
    """One live crop of the pool AND the object a tracker stores (`crops[i]` of the lazy / device-only get_image_crops modes; alias
    tracking.DeviceCrop): round 5 merged the two - one Python object per crop instead of a Slot, a DeviceCrop wrapper and a (frame, row) tuple.
    `ptr` is its device address (0 once spilled); `host` its host bytes (None until needed).  To duck-typing callers it is a uint8
    [384,128,3] array: any host read (`np.array(crop)`, arithmetic, `.astype`, indexing, pickling) returns the real pixels - from the
    frame's asynchronous host copy (lazy mode) or by copying the slot back on demand (device-only mode)."""
    __slots__ = ("pool", "slab", "index", "ptr", "host", "host_src", "__weakref__")
    shape, dtype, ndim, size = (CROP_H, CROP_W, 3), np.dtype(np.uint8), 3, CROP_BYTES

    def __init__(self, pool, slab, index, ptr, host_src=None):
        self.pool, self.slab, self.index, self.ptr, self.host = pool, slab, index, ptr, None
        self.host_src = host_src            # (frame host copy in flight, row): tracking.FrameHostCopy of get_image_crops' lazy mode

    @property
    def slot(self):
        return self

    @property
    def dev(self):
        return self.tensor()

    def tensor(self):
        """cuda u8 [384,128,3] view of the slot, or None after a spill."""
        return self.pool.slabs[self.slab][self.index] if self.ptr else None

    def host_bytes(self):
        """uint8 [384,128,3] host copy (device -> host on first use when the crop only lived in HBM)."""
        if self.host is None and self.host_src is not None:
            frame, k = self.host_src            # the frame's asynchronous device->host copy: wait for its event, then it is a plain view
            rows = frame.rows()                 # (None once that copy has been retired: the slot itself is read below)
            self.host_src = None
            if rows is not None:
                self.host = rows[k]
        if self.host is None:
            if not self.ptr:
                raise RuntimeError("crop slot lost both its device and its host copy")       # cannot happen: spill copies first
            t = self.tensor()
            self.host = t.cpu().numpy() if t.is_cuda else t.numpy().copy()     # (.cpu() of a host tensor would alias the slot)
        return self.host

    # ---- the array face ------------------------------------------------------------------------------------------------
    def __array__(self, dtype=None, copy=None):
        a = self.host_bytes()
        return a if dtype is None else a.astype(dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(np.asarray(x) if isinstance(x, Slot) else x for x in inputs)
        return getattr(ufunc, method)(*inputs, **kwargs)

    def astype(self, dtype, **kw):
        return np.asarray(self).astype(dtype, **kw)

    def copy(self):
        return np.array(self.host_bytes())

    def __getitem__(self, key):
        return np.asarray(self)[key]

    def __len__(self):
        return CROP_H

    def __reduce__(self):
        return np.asarray(self).__reduce__()

    def __del__(self):
        try:
            self.pool._release(self)
        except Exception:
            pass


class CropPool:
    def __init__(self, device, budget_bytes=None, slab_crops=512):
        if budget_bytes is None:
            budget_bytes = int(float(os.environ.get("BUSCA_CROP_POOL_MB", "16384")) * (1 << 20))
        self.device = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        self.slab_crops = int(slab_crops)
        self.max_slabs = max(1, int(budget_bytes) // (self.slab_crops * CROP_BYTES))
        self.slabs = []
        self.slab_ptrs = []                 # data_ptr() of every slab (asked once, not per slot)
        self.free = []                      # (slab, index)
        self.live = {}                      # id -> weakref(Slot); dicts keep insertion order: allocation order = eviction order
        self.spilled = 0                    # crops moved to the host because the budget was reached
        self.peak_live = 0

    # ---- accounting ----------------------------------------------------------------------------------------------------
    @property
    def capacity(self):
        return len(self.slabs) * self.slab_crops

    @property
    def device_bytes(self):
        return self.capacity * CROP_BYTES

    @property
    def n_live(self):
        return len(self.live)

    # ---- allocation ----------------------------------------------------------------------------------------------------
    def _grow(self):
        t = torch.empty(self.slab_crops, CROP_H, CROP_W, 3, dtype=torch.uint8, device=self.device)
        self.slabs.append(t)
        self.slab_ptrs.append(t.data_ptr())
        s = len(self.slabs) - 1
        self.free.extend((s, i) for i in range(self.slab_crops - 1, -1, -1))

    def _spill_oldest(self, k):
        """Move the k oldest live crops to the host and reuse their slots."""
        if k > len(self.live):
            raise RuntimeError("crop pool budget (%d crops) is smaller than one request" % (self.max_slabs * self.slab_crops))
        for _ in range(k):
            ref = self.live.pop(next(iter(self.live)))
            slot = ref()
            if slot is None or not slot.ptr:
                continue
            slot.host_bytes()
            self.free.append((slot.slab, slot.index))
            slot.ptr = 0
            self.spilled += 1

    def alloc(self, n, host_frame=None):
        """n fresh slots (list of Slot).  Their contents are undefined until a kernel writes them.  `host_frame`: the FrameHostCopy whose row k will
        hold slot k's host bytes (lazy mode)."""
        while len(self.free) < n:
            if len(self.slabs) < self.max_slabs:
                self._grow()
            else:
                self._spill_oldest(n - len(self.free))
        out = []
        free, live, ptrs, ref = self.free, self.live, self.slab_ptrs, weakref.ref
        for k in range(n):
            s, i = free.pop()
            slot = Slot(self, s, i, ptrs[s] + i * CROP_BYTES, None if host_frame is None else (host_frame, k))
            live[id(slot)] = ref(slot)
            out.append(slot)
        self.peak_live = max(self.peak_live, len(self.live))
        return out

    def _release(self, slot):
        if self.live.pop(id(slot), None) is not None and slot.ptr:
            self.free.append((slot.slab, slot.index))
            slot.ptr = 0
