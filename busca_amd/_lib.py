"""ctypes binding of libbusca_hip.so (include/busca_hip.h).  No fallback: if the library is missing the
import of any compute path raises - the product never computes on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbusca_hip.so")

ACT_RELU, ACT_GELU = 0, 1
LAYOUT_CAN_FIRST, LAYOUT_NO_BAD, LAYOUT_SEP_AS_CAN = 1, 2, 4      # busca_dt_cfg.layout bits
PREC_F32, PREC_F16, PREC_F16X3 = 0, 1, 2
ABI_MAJOR = 2                  # busca_version() // 1000 this binding was written for (include/busca_hip.h)
PAIR_CENTER, PAIR_CENTER_WEIGHTED, PAIR_IOU, PAIR_IOU_COST = 0, 1, 2, 3


class DTCfg(C.Structure):
    _fields_ = [("d", C.c_int32), ("ff", C.c_int32), ("nhead", C.c_int32), ("nlayers", C.c_int32),
                ("E", C.c_int32), ("activation", C.c_int32), ("fake_bbox_f64", C.c_int32),
                ("precision", C.c_int32), ("layout", C.c_int32)]


# name -> (restype, argtypes); mirrors include/busca_hip.h one to one
_vp, _i32, _sz = C.c_void_p, C.c_int32, C.c_size_t
SIGNATURES = {
    "busca_version": (C.c_int, []),
    "busca_build_info": (C.c_char_p, []),
    "busca_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "busca_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(_i32)]),
    "busca_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "busca_ctx_destroy": (None, [_vp]),
    "busca_last_error": (C.c_char_p, [_vp]),
    "busca_dt_blob_floats": (_sz, [C.POINTER(DTCfg)]),
    "busca_dt_load_weights": (C.c_int, [_vp, C.POINTER(DTCfg), _vp, _sz, _vp, _vp, _vp, _i32]),
    "busca_dt_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "busca_dt_reserve": (C.c_int, [_vp, _i32, _i32, _i32, _vp]),
    "busca_dt_bucket_ids": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "busca_timing_enable": (C.c_int, [_vp, _i32]),
    "busca_timing_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), _i32]),
    "busca_pairwise": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "busca_topk_rows": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "busca_coverage": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "busca_ecc_align": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_double, _vp, C.POINTER(C.c_double), C.POINTER(_i32), _vp]),
    "busca_kalman_multi_predict": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "busca_duplicate_masks": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, C.c_double, _vp, _vp, _vp]),
    "busca_crop_gather": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "busca_crop_gather_ex": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "busca_crop_gather_sized": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    "busca_gather_crops": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "busca_reid_forward_ex": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "busca_reid_forward_w": (C.c_int, [_vp, _vp, _i32, _vp, _vp, C.c_double, _vp, _vp]),
    "busca_reid_blob_floats": (_sz, []),
    "busca_reid_load_weights": (C.c_int, [_vp, _vp, _sz]),
    "busca_reid_load_weights_ex": (C.c_int, [_vp, _vp, _sz, _i32]),
    "busca_reid_forward": (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
    "busca_reid_workspace_bytes": (_sz, [_i32]),
    "busca_reid_reserve": (C.c_int, [_vp, _i32, _vp]),
    "busca_bn_stats_1x1": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """dlopen the in-tree library and type every entry point.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing - run `python -m busca_amd.build` (hipcc, gfx950). "
                              "busca_amd has no CPU fallback." % LIB_PATH)
        # torch must be imported first: it bundles its own libamdhip64 (SONAME libamdhip64.so.7); when that
        # copy is already loaded, this library's NEEDED libamdhip64.so.7 binds to it and both share ONE HIP
        # runtime (same device pointers, same streams).  Loading in the other order gives two runtimes.
        import torch  # noqa: F401
        lib = C.CDLL(os.environ.get("BUSCA_HIP_LIB", LIB_PATH))      # override: experiment builds of the same ABI
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        v = lib.busca_version()
        if v // 1000 != ABI_MAJOR:
            raise ImportError("libbusca_hip.so has ABI version %d, this binding needs major %d - rebuild (python -m busca_amd.build --force)" % (v, ABI_MAJOR))
        _lib = lib
    return _lib


class BuscaError(RuntimeError):
    pass


def build_info(lib=None):
    """The flag set the loaded library was compiled with (busca_build_info)."""
    return (lib or load()).busca_build_info().decode()


class Context:
    """One busca_ctx per process/GPU."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.busca_ctx_create(int(device), C.byref(h))
        if rc != 0 or not h.value:
            raise BuscaError("busca_ctx_create(device=%d) failed with %d: %s" % (
                device, rc, self.lib.busca_last_error(None).decode()))
        self.h = h
        self.device = int(device)

    def check(self, rc):
        if rc != 0:
            raise BuscaError("libbusca_hip error %d: %s" % (rc, self.lib.busca_last_error(self.h).decode()))

    def set_option(self, name, value):
        """Developer option of this context (include/busca_hip.h: busca_set_option)."""
        self.check(self.lib.busca_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name):
        v = _i32(0)
        self.check(self.lib.busca_get_option(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def close(self):
        # host-side attachments of busca_amd.geometry (frame scope, pinned tables still referenced behind their launches, crop pool)
        self._frame_scope = None
        q = getattr(self, "_pinned_in_flight", None)
        if q:
            for _, ev in q:
                ev.synchronize()
            q.clear()
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.busca_ctx_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
