"""Host handle of the fused Decision-Transformer kernel (busca_dt_* in include/busca_hip.h).

Inputs/outputs are torch tensors on the GPU; torch only provides device memory and the stream."""
import ctypes as C

import weakref

import numpy as np
import torch

from . import _lib, weights

_PREC = {"f32": _lib.PREC_F32, "f16": _lib.PREC_F16, "x3": _lib.PREC_F16X3}      # x3: float32-equivalent split-fp16 GEMMs (dt_kernel.hip.inc, Prec<2>)
_ACT = {"relu": _lib.ACT_RELU, "gelu": _lib.ACT_GELU}
FLAVOURS = ("MEM-SEP-CAN-BAD", "MEM-SEP-CAN", "MEM-CAN-SEP-BAD", "MEM-CAN-SEP")


def layout_bits(input_flavour, encode_separator_as_reference=True):
    """busca_dt_cfg.layout for an input flavour of network.py:103-165 (the CLS-* ones fail inside the reference itself,
    encodings.py:161, and are refused here)."""
    if input_flavour not in FLAVOURS:
        raise NotImplementedError('Input flavour "{}" not implemented'.format(input_flavour))
    return ((_lib.LAYOUT_CAN_FIRST if "MEM-CAN-SEP" in input_flavour else 0) | (0 if "BAD" in input_flavour else _lib.LAYOUT_NO_BAD)
            | (0 if encode_separator_as_reference else _lib.LAYOUT_SEP_AS_CAN))


class DecisionTransformerHIP:
    def __init__(self, ctx, state_dict, activation="relu", fake_bbox_f64=True, precision="f32", input_flavour="MEM-SEP-CAN-BAD",
                 encode_separator_as_reference=True, nhead=4):
        self.ctx = ctx
        self.input_flavour = input_flavour
        self.nspec = 2 if "BAD" in input_flavour else 1        # appended candidates: NON [, BAD]
        self.can_pos = 0 if "MEM-CAN-SEP" in input_flavour else 1
        # the float64 promotion of the candidate-side bucket math comes from torch.cat with the float64 BAD box (encodings.py:21,
        # 127-140): no BAD token, no promotion
        fake_bbox_f64 = bool(fake_bbox_f64) and self.nspec == 2
        d, ff, nl, E = weights.dt_dims(state_dict)
        self.d, self.ff, self.nlayers, self.E, self.nhead = d, ff, nl, E, int(nhead)    # (nhead is not in the state_dict: in_proj is [3d, d] whatever it is)
        self.precision = precision
        self.cfg = _lib.DTCfg(d, ff, self.nhead, nl, E, _ACT[activation], 1 if fake_bbox_f64 else 0, _PREC[precision],
                              layout_bits(input_flavour, encode_separator_as_reference))
        self._blob = weights.dt_blob(state_dict, nl)
        want = ctx.lib.busca_dt_blob_floats(C.byref(self.cfg))
        if want == 0:
            raise _lib.BuscaError("unsupported Decision-Transformer shape d=%d ff=%d nhead=%d layers=%d E=%d (built: d in 64/256/512, "
                                  "d/nhead in 16/32/64/128, ff a multiple of d up to 8 d, E = 512)" % (d, ff, self.nhead, nl, E))
        assert self._blob.size == want, (self._blob.size, want)
        self._luts = weights.encoding_luts(d)
        self._rerun = {}            # x3: id(logits) -> (weakref, inputs) of forwards not settled yet (see settle)
        self.exact_reruns = 0       # steps settle() re-ran in exact float32 because the x3 forward reported a clipped operand
        self._upload()

    def _upload(self):
        ctx, (lxy, lsz, lt, c) = self.ctx, self._luts
        ctx.check(ctx.lib.busca_dt_load_weights(ctx.h, C.byref(self.cfg), self._blob.ctypes.data, self._blob.size,
                                                lxy.ctypes.data, lsz.ctypes.data, lt.ctypes.data, c))
        ctx.dt_owner = weakref.ref(self)

    def _ensure_loaded(self):
        """A busca_ctx holds ONE Decision-Transformer weight set (include/busca_hip.h).  If another handle loaded its
        own weights into this context since, put this model's back before computing - never run on someone else's."""
        owner = getattr(self.ctx, "dt_owner", None)
        if owner is None or owner() is not self:
            self._upload()

    @staticmethod
    def _f32(t, dev):
        if not torch.is_tensor(t):
            t = torch.as_tensor(np.asarray(t))
        if t.device.type == "cpu":      # host boxes / features: pinned staging + asynchronous copy (a pageable `.to` parks the host behind everything queued on the stream)
            t = t.to(torch.float32).contiguous().pin_memory().to(dev, non_blocking=True)
        return t.to(device=dev, dtype=torch.float32).contiguous()

    def forward(self, mem_feat, can_feat, mem_ltrb, can_ltrb, want_hidden=False, want_att=False, stream=None):
        """-> dict(logits[B,P+2], probs[B,P+2], argmax[B] int32, hidden?[B,T,d], att?[nl,B,nhead,T,T]); P+1 columns without the BAD token."""
        self._ensure_loaded()
        dev = torch.device("cuda", self.ctx.device)
        mem_feat, can_feat = self._f32(mem_feat, dev), self._f32(can_feat, dev)
        mem_ltrb, can_ltrb = self._f32(mem_ltrb, dev), self._f32(can_ltrb, dev)
        B, L, E = mem_feat.shape
        P = can_feat.shape[1]
        assert E == self.E and can_feat.shape == (B, P, E) and mem_ltrb.shape == (B, L, 4) and can_ltrb.shape == (B, P, 4)
        n = P + self.nspec
        T = L + 2 * n
        out = dict(logits=torch.empty(B, n, device=dev), probs=torch.empty(B, n, device=dev),
                   argmax=torch.empty(B, dtype=torch.int32, device=dev))
        if want_hidden:
            out["hidden"] = torch.empty(B, T, self.d, device=dev)
        if want_att:
            out["att"] = torch.empty(self.nlayers, B, self.nhead, T, T, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        self.ctx.check(self.ctx.lib.busca_dt_forward(
            self.ctx.h, mem_feat.data_ptr(), can_feat.data_ptr(), mem_ltrb.data_ptr(), can_ltrb.data_ptr(), B, L, P,
            out["logits"].data_ptr(), out["probs"].data_ptr(), out["argmax"].data_ptr(),
            _lib.ptr(out.get("hidden")), _lib.ptr(out.get("att")), s))
        if self.precision == "x3":     # what settle() needs to run the step once more (keeps the inputs alive as long as the outputs)
            self._rerun[id(out["logits"])] = (weakref.ref(out["logits"]), (mem_feat, can_feat, mem_ltrb, can_ltrb, want_hidden, want_att, stream))
            if len(self._rerun) > 64:
                self._rerun = {k: v for k, v in self._rerun.items() if v[0]() is not None}
        return out

    def settle(self, out):
        """Call once the stream of `out`'s forward is SYNCHRONISED (the caller has just copied logits / probabilities to the host) and before the results are
        used.  Reads the status word the kernels leave in host-mapped memory (`dt_status`, include/busca_hip.h):
          0  -> `out` as it is;
          2  -> the x3 forward had to clip an operand beyond |x| = 1023.5, its results are not float32-equivalent: the SAME step is run again in exact float32
                on the f32 packing of the same weights (`dt_exact_f32`), synchronised, and THAT result is returned - the caller never sees a clipped step and
                nothing is raised (the reference, busca/network.py:401-405, cannot fail there either);
          1  -> a token-split launch lost a partner workgroup: raised for THIS call.
        The status is cleared either way, so the C-side backstop of the next busca_dt_forward stays silent."""
        st = self.ctx.get_option("dt_status")
        if st == 0:
            self._rerun.pop(id(out["logits"]), None)
            return out
        self.ctx.set_option("dt_status", 0)
        if st != 2:
            raise _lib.BuscaError("a token-split Decision-Transformer launch gave up waiting for a partner workgroup: the results of this forward are invalid")
        ent = self._rerun.pop(id(out["logits"]), None)
        if ent is None or ent[0]() is not out["logits"]:
            raise _lib.BuscaError("an x3 Decision-Transformer forward clipped an operand (|x| > 1023.5) and its inputs are gone: load the model with precision='f32'")
        mem_feat, can_feat, mem_ltrb, can_ltrb, want_hidden, want_att, stream = ent[1]
        self.exact_reruns += 1
        self.ctx.set_option("dt_exact_f32", 1)
        try:
            self._ensure_loaded()
            fixed = self.forward(mem_feat, can_feat, mem_ltrb, can_ltrb, want_hidden=want_hidden, want_att=want_att, stream=stream)
        finally:
            self.ctx.set_option("dt_exact_f32", 0)
        self._rerun.pop(id(fixed["logits"]), None)
        if stream is None:
            torch.cuda.current_stream(torch.device("cuda", self.ctx.device)).synchronize()
        else:
            torch.cuda.synchronize(torch.device("cuda", self.ctx.device))
        return fixed

    def can_positions(self, L, P):
        """Rows of the candidate tokens (incl. NON [, BAD]) in the token sequence (network.py:142,154)."""
        return [L + 2 * j + self.can_pos for j in range(P + self.nspec)]

    def reserve(self, B, L, P):
        """Size the layer-wise path's HBM workspace for (B, L, P) now, so that no later forward allocates (busca_dt_reserve)."""
        self._ensure_loaded()
        s = torch.cuda.current_stream(torch.device("cuda", self.ctx.device)).cuda_stream
        self.ctx.check(self.ctx.lib.busca_dt_reserve(self.ctx.h, int(B), int(L), int(P), s))

    def bucket_ids(self, mem_ltrb, can_ltrb):
        self._ensure_loaded()
        dev = torch.device("cuda", self.ctx.device)
        mem_ltrb, can_ltrb = self._f32(mem_ltrb, dev), self._f32(can_ltrb, dev)
        B, L, _ = mem_ltrb.shape
        P = can_ltrb.shape[1]
        ids = torch.empty(B, L + 2 * (P + self.nspec), 3, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        self.ctx.check(self.ctx.lib.busca_dt_bucket_ids(self.ctx.h, mem_ltrb.data_ptr(), can_ltrb.data_ptr(), B, L, P,
                                                       ids.data_ptr(), s))
        return ids
