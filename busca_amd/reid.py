"""Host handle of the ReID feature extractor kernels (busca_reid_* in include/busca_hip.h)."""
import weakref

import numpy as np
import torch

from . import weights


PRECISIONS = {"f32": 0, "f16": 1, "x3": 2}          # BUSCA_PREC_F32 / _F16 / _F16X3 (include/busca_hip.h)


def _ptr(t):
    return t.data_ptr() if t is not None else None


class ReIDEncoderHIP:
    """ResNet-50 (max pool, red=4) with batch-statistics BatchNorm, fp16 MFMA convs.
    `forward(crops_u8)`: u8 [n,384,128,3] BGR -> f32 [n,512] L2-normalised.  One call == one BN batch."""
    PRETRAINED_SIZE = (384, 128)

    def __init__(self, ctx, state_dict, prefix="", precision="f16"):
        """precision "f16": fp16 activations/weights (fastest, features ~6e-3 from the reference); "f32": exact float32 convs on the
        f32 MFMA (reference-exact ~1e-5, ~6x slower); "x3": float32 activations with split-fp16 products, three fp16 MFMAs per block
        (BUSCA_PREC_F16X3: the same ~1e-5 at a third of the fp16 matrix rate)."""
        if precision not in PRECISIONS:
            raise ValueError("ReID precision %r (one of %s)" % (precision, sorted(PRECISIONS)))
        self.ctx = ctx
        self.precision = precision
        self._blob = weights.reid_blob(state_dict, prefix)
        if precision == "x3":
            # a HINT only (a loose static bound: 48 sigma, summed over a layer's bottlenecks): the kernels themselves report an operand that leaves the
            # split-fp16 range at run time (`take_status`), and busca_amd.network re-runs such a batch on the exact-f32 extractor
            self.x3_activation_bound, self.x3_activation_bound_where = weights.x3_activation_bound(state_dict, prefix)
        want = ctx.lib.busca_reid_blob_floats()
        assert self._blob.size == want, (self._blob.size, want)
        self._upload()

    def _upload(self):
        ctx = self.ctx
        ctx.check(ctx.lib.busca_reid_load_weights_ex(ctx.h, self._blob.ctypes.data, self._blob.size, PRECISIONS[self.precision]))
        ctx.reid_owner = weakref.ref(self)

    def _ensure_loaded(self):
        """A busca_ctx holds ONE ReID weight set; restore this model's if another handle replaced it (see dt.py)."""
        owner = getattr(self.ctx, "reid_owner", None)
        if owner is None or owner() is not self:
            self._upload()

    def take_status(self):
        """Call once the streams of this extractor's forwards are SYNCHRONISED.  True: a split-fp16 (x3) forward since the last call staged an activation beyond
        |x| = 1023.5 (`reid_status` 2, include/busca_hip.h) - its BatchNorm statistics, hence the features of that batch, are not finite / not valid, and the caller
        must compute the batch again on an exact-f32 extractor (BUSCA._assoc_finish does).  The status is cleared.  Always False for the f32 / f16 flavours."""
        if self.precision != "x3":
            return False
        st = self.ctx.get_option("reid_status")
        if st:
            self.ctx.set_option("reid_status", 0)
        return st != 0

    def reserve(self, n, stream=None):
        """Size the workspace that forwards on `stream` (default: the current stream) use for batches of up to n crops NOW
        (busca_reid_reserve), so that no later forward synchronises the device and allocates."""
        self._ensure_loaded()
        dev = torch.device("cuda", self.ctx.device)
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        self.ctx.check(self.ctx.lib.busca_reid_reserve(self.ctx.h, int(n), s))

    def forward(self, crops_u8, stream=None, zero_norm=None, weights=None):
        """`zero_norm` (cuda u8 [n] or None): crops flagged 1 are 0.0 after normalisation (busca_reid_forward_ex).
        `weights` (numpy / sequence of n multiplicities, or None): crop i stands for weights[i] identical crops of the BatchNorm
        batch - each distinct crop is computed once, the batch statistics count it weights[i] times (busca_reid_forward_w)."""
        self._ensure_loaded()
        dev = torch.device("cuda", self.ctx.device)
        if not torch.is_tensor(crops_u8):
            crops_u8 = torch.from_numpy(np.ascontiguousarray(crops_u8))
        crops_u8 = crops_u8.to(dev).contiguous()
        assert crops_u8.dtype == torch.uint8 and tuple(crops_u8.shape[1:]) == (384, 128, 3), crops_u8.shape
        n = crops_u8.shape[0]
        feats = torch.empty(n, 512, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        zn = zero_norm.data_ptr() if zero_norm is not None else None
        wd, wsum = None, 0.0
        if weights is not None and not (np.asarray(weights) != 1).any():
            weights = None                      # every crop once: the plain forward (same schedule, bit for bit)
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float32)
            assert w.shape == (n,) and (w >= 1).all()
            wsum = float(w.astype(np.float64).sum())
            if stream is not None:      # the copy must be ordered on the stream the pass runs on
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                    wd = torch.from_numpy(w).pin_memory().to(dev, non_blocking=True)
            else:
                wd = torch.from_numpy(w).pin_memory().to(dev, non_blocking=True)
        self.ctx.check(self.ctx.lib.busca_reid_forward_w(self.ctx.h, crops_u8.data_ptr(), n, zn, _ptr(wd), wsum, feats.data_ptr(), s))
        return feats
