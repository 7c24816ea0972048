"""CPU: host logic of the device crop pool (busca_amd/crop_pool.py) - slot lifetime, budget, spill - on a CPU torch device
(the pool only needs an allocator and addresses; the kernels that fill / gather slots are covered by the -m gpu tests)."""
import gc

import numpy as np
import torch

from busca_amd.crop_pool import CROP_BYTES, CropPool
from busca_amd.tracking import DeviceBackedCrops, DeviceCrop, DeviceCrops, box_extents


def _pool(crops, slab=4):
    return CropPool(torch.device("cpu"), budget_bytes=crops * CROP_BYTES, slab_crops=slab)


def test_slots_return_when_the_last_reference_dies():
    pool = _pool(8)
    a = pool.alloc(5)
    assert pool.n_live == 5 and pool.capacity == 8 and len({s.ptr for s in a}) == 5
    host = np.zeros((5, 384, 128, 3), np.uint8)
    crops = DeviceBackedCrops(host, a)
    del a
    keep = crops[2]
    assert keep.slot is not None and keep.slot.host is not None
    del crops
    gc.collect()
    assert pool.n_live == 1                       # only the crop a "track" still holds
    del keep
    gc.collect()
    assert pool.n_live == 0 and len(pool.free) == pool.capacity


def test_budget_spills_oldest_to_host_and_stays_flat():
    pool = _pool(8)
    held = []
    for i in range(10):
        s = pool.alloc(2)
        for k, sl in enumerate(s):
            sl.tensor().fill_(10 * i + k)         # "kernel" writes the crop
        held.extend(s)
        assert pool.device_bytes <= 8 * CROP_BYTES
    assert pool.capacity == 8 and pool.spilled == 12 and pool.n_live == 8
    for j, sl in enumerate(held):                 # spilled crops kept their pixels on the host; resident ones are intact
        want = 10 * (j // 2) + j % 2
        assert (sl.host_bytes() == want).all()
        assert (sl.ptr == 0) == (j < 12)


def test_copies_lose_the_slot_but_keep_pixels():
    import pickle
    pool = _pool(4)
    s = pool.alloc(2)
    host = np.arange(2 * 384 * 128 * 3, dtype=np.uint8).reshape(2, 384, 128, 3)
    crops = DeviceBackedCrops(host, s)
    c = crops[1]
    for copy in (np.array(c), np.ascontiguousarray(c), c.astype(np.float32), pickle.loads(pickle.dumps(c)), np.stack([c])[0]):
        assert getattr(copy, "slot", None) is None and np.array_equal(np.asarray(copy, dtype=np.uint8), host[1])
    assert type(np.array(c)) is np.ndarray


def test_device_only_crop_reads_real_pixels():
    pool = _pool(4)
    s = pool.alloc(3)
    for k, sl in enumerate(s):
        sl.tensor().fill_(7 + k)
    crops = DeviceCrops(s)
    assert crops.shape == (3, 384, 128, 3) and isinstance(crops[1], DeviceCrop) and s[1].host is None
    assert (np.array(crops[1]) == 8).all() and s[1].host is not None
    assert ((crops[2] / 1.0) == 9).all() and (crops[0].astype(np.int32) == 7).all()
    assert np.asarray(crops).shape == (3, 384, 128, 3)


def test_box_extents_round_in_float64():
    # 99.99999999 floors to 99 in float64; its float32 copy is exactly 100.0 and would floor to 100 (one pixel wider cut)
    r = box_extents([[99.99999999, 10.2, 200.00000001, 50.0]])
    assert r.tolist() == [[99, 10, 201, 50]] and r.dtype == np.int32
    assert int(np.floor(np.float32(99.99999999))) == 100
    assert box_extents(np.zeros((0, 4))).shape == (0, 4)
    assert box_extents([[np.nan, -1e30, 1e30, 3.5]]).tolist() == [[0, -2 ** 30, 2 ** 30, 4]]
