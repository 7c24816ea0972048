"""`BUSCA` with the reference's public surface (busca/network.py:11-507), running on libbusca_hip.so.

What stays on the host (Python): walking the tracker's track objects (`images_mem`, `tlwh_mem`, `scale`,
`tlwh`) and scattering P+2 probabilities into the [B, N_det + B] matrix the trackers consume.
What runs on the MI355X: top-P proposal selection, the ReID extractor (two train-mode-BN batches per
step), token embed/assembly, bucket encoding, the 4 encoder layers, decoder, softmax and argmax.
There is no CPU compute path: without the HIP library / a GPU every compute call raises.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, geometry, synth, tracking, weights
from .dt import DecisionTransformerHIP
from .reid import ReIDEncoderHIP

_FLAVOURS = ("MEM-SEP-CAN-BAD", "MEM-SEP-CAN", "MEM-CAN-SEP-BAD", "MEM-CAN-SEP")   # network.py:103-165 without the CLS- ones
_REID_PREFIX = "reid_encoder.model."
_PIX_MEAN_RGB = np.array([0.485, 0.456, 0.406], dtype=np.float64)
_PIX_STD_RGB = np.array([0.299, 0.224, 0.225], dtype=np.float64)


_SIDE_STREAMS = {}


def _side_stream_of(dev):
    key = dev.index or 0
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(dev)
    return _SIDE_STREAMS[key]


def memory_indices(n_hist, seq_len, use_broader_memory):
    """Which history entries form a track's memory (busca/network.py:247-275): the last `seq_len`, or - with
    `use_broader_memory` and a long enough history - `seq_len` entries spread evenly from first to last."""
    if use_broader_memory and n_hist >= seq_len and seq_len > 1:
        step = float(n_hist - 1) / float(seq_len - 1)
        return [int(i * step) for i in range(seq_len)]
    return list(range(max(0, n_hist - seq_len), n_hist))


class _ReIDFacade:
    """`model.reid_encoder`: callable like the reference's ReID_Encoder (network.py:510-575): returns
    (None, feats[n,512]); accepts u8 BGR [n,384,128,3] crops or normalised float RGB [n,3,384,128]."""
    PRETRAINED_SIZE = (384, 128)

    def __init__(self, owner):
        self._owner = owner

    def __call__(self, x):
        return None, self._owner._reid_features(x)

    forward = __call__
    get_features = __call__


class BUSCA:
    def __init__(self, args):
        self.args = args
        self.dim_embedding = args.dim_embedding
        self.dim_model = args.trans_dim
        if args.activation not in ("relu", "gelu", "tanh", "silu"):
            raise RuntimeError("activation should be relu/gelu/tanh/silu, not {}".format(args.activation))
        if args.input_flavour.startswith("CLS-"):
            # the reference cannot run these either: PositionalEncoding._get_temporal_ids replaces the index tensor by the int 0
            # and fails (encodings.py:161; recorded from the reference in tests/golden/flavours_dt.npz)
            raise NotImplementedError('Input flavour "{}" not implemented (the CLS- flavours fail inside the reference\'s own '
                                      'positional encoding, busca/encodings.py:161)'.format(args.input_flavour))
        if args.input_flavour not in _FLAVOURS:
            raise NotImplementedError('Input flavour "{}" not implemented'.format(args.input_flavour))
        if getattr(args, "output_flavour", "CAN") != "CAN":
            raise NotImplementedError("only output_flavour=CAN (all shipped configs) is built")
        if args.encode_special_tokens and args.dim_embedding != args.trans_dim:
            # the learned tokens then have dim_embedding entries and are concatenated, unencoded, with trans_dim-wide rows
            # (network.py:52-69,128-130): torch.cat raises in the reference
            raise RuntimeError("Sizes of tensors must match except in dimension 1: encode_special_tokens needs dim_embedding == trans_dim")
        d, nh, ff = int(args.trans_dim), int(args.nhead), int(args.ff_size)
        if args.dim_embedding != 512:
            raise NotImplementedError("dim_embedding must be 512 (the ReID feature, all shipped configs)")
        if d not in (64, 256, 512) or nh < 1 or d % nh or d // nh not in (16, 32, 64, 128) or ff < d or ff % d or ff > 8 * d:
            raise NotImplementedError("built: trans_dim in {64, 256, 512}, trans_dim / nhead in {16, 32, 64, 128}, ff_size a multiple of "
                                      "trans_dim up to 8x (nhead 4 with ff_size = 2 trans_dim - every shipped config - runs as one kernel, "
                                      "the others layer-wise)")
        # The reference's cloned encoder layers run ReLU whatever `activation` says (deepcopy +
        # TransformerEncoderLayer.__setstate__, custom_layers.py:24-27,44-45).  Set
        # args.fix_activation_quirk = True to run the configured activation instead (gelu only).
        self.effective_activation = "relu"
        if getattr(args, "fix_activation_quirk", False):
            if args.activation not in ("relu", "gelu"):
                raise NotImplementedError("only relu/gelu are built")
            self.effective_activation = args.activation
        # Decision-Transformer arithmetic: "x3" (default) = float32-EQUIVALENT GEMMs on the fp16 matrix cores (three fp16 MFMAs per product block; logits 1e-5
        # from the exact flavour, same test bars, 2.2x its speed), "f32" = exact float32 MFMA, "f16" = fp16 operands (opt-in, ~7e-3)
        self.precision = getattr(args, "precision", os.environ.get("BUSCA_AMD_PRECISION", "x3"))
        # ReID flavour.  "x3" (default since round 4): float32 activations, float32-equivalent products as three fp16 MFMAs - features within
        # 1e-5 of the reference's, association probabilities within 1e-3, identical decisions (tests/test_associate_gpu.py).  "f32": the same
        # parity on the exact f32 MFMA, 2x slower.  "f16": fp16 activations, 2.2-2.8x faster than x3, but it moves probabilities by up to 0.03
        # with random weights and flips 3.5 % of the `> 0.5` decisions of a sharp model (profiles/r04_decision_agreement.json): opt-in.
        self.reid_precision = getattr(args, "reid_precision", os.environ.get("BUSCA_AMD_REID_PRECISION", "x3"))
        self.pinned_numpy = bool(getattr(args, "pinned_numpy_semantics", True))
        # True: get_image_crops(normalize=False) keeps the crops in the device pool only; host reads copy them back on demand
        self.device_only_crops = bool(getattr(args, "device_only_crops", False))
        # host bytes of get_image_crops(normalize=False): "lazy" (default; copied to pinned memory on a side stream, waited for only by a
        # host read), "eager" (a real ndarray, the call waits for the copy) - device_only_crops=True means "never"
        self.crop_host_copy = getattr(args, "crop_host_copy", os.environ.get("BUSCA_AMD_CROP_HOST_COPY", "lazy"))
        self.store_logits = False           # set True to fill .logits / .mem_logits like the reference does
        # True: crops that occur several times in a BatchNorm batch (one detection among the candidates of many tracks, zero
        # padding) are computed once, with weighted batch statistics - same result up to summation order, less ReID work
        self.dedup_crops = bool(getattr(args, "dedup_crops", True))
        self.expected_image_size = _ReIDFacade.PRETRAINED_SIZE
        self.reid_encoder = _ReIDFacade(self)
        self.attentions = None
        self.logits = None
        self.mem_logits = None
        self._device_index = self._index_of(getattr(args, "device", None))
        self._ctx = None
        self._dt = None
        self._reid = None
        self._dirty = True
        seed = int(getattr(args, "seed", 0))
        sd = OrderedDict(synth.dt_state_dict(seed, d=args.trans_dim, ff=args.ff_size, nlayers=args.num_layer, flavour=args.input_flavour))
        reid_sd = None
        path = getattr(args, "reid_weights_file", "no")
        if path is not None and path != "no":
            reid_sd = self._read_checkpoint(path)
            reid_sd = {k: v for k, v in reid_sd.items() if "fc" not in k.split(".") and "fc_person" not in k.split(".")}
        for k, v in synth.reid_state_dict(seed).items():
            sd[_REID_PREFIX + k] = v
        self._sd = sd
        if reid_sd is not None:             # load_net (busca/reid/load_trained_net.py:43-66): update + strict load_state_dict
            self._load_checked({_REID_PREFIX + k: v for k, v in reid_sd.items()}, what="reid_weights_file %r" % path, scope=_REID_PREFIX)

    # ---- nn.Module-like surface -------------------------------------------------------------------------
    @staticmethod
    def _index_of(device):
        if device is None:
            return 0
        if isinstance(device, torch.device):
            if device.type != "cuda":
                raise RuntimeError("busca_amd has no CPU path; args.device must be a cuda device")
            return device.index or 0
        if isinstance(device, str):
            return BUSCA._index_of(torch.device("cuda" if device in ("gpu", "cuda") else device))
        return int(device)

    def to(self, device):
        idx = self._index_of(device)
        if idx != self._device_index:
            self._device_index, self._ctx, self._dt, self._reid, self._dirty = idx, None, None, None, True
        return self

    def eval(self):
        return self

    def train(self, mode=True):
        return self

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self._sd.items())

    def load_state_dict(self, sd, strict=True):
        """nn.Module.load_state_dict semantics on this model's key space (every parameter of the reference's BUSCA that
        the hot path reads; BatchNorm running statistics and the ReID classifier heads are not part of it, see
        `_ignorable`): unexpected / missing keys raise when `strict`."""
        unexpected = [k for k in sd if k not in self._sd and not self._ignorable(k)]
        missing = [k for k in self._sd if k not in sd]
        if strict and (unexpected or missing):
            raise RuntimeError("Error(s) in loading state_dict for BUSCA: missing keys {}, unexpected keys {}".format(missing[:8], unexpected[:8]))
        for k, v in sd.items():
            if k in self._sd:
                arr = np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v, dtype=np.float32)
                if arr.shape != self._sd[k].shape:
                    raise RuntimeError("size mismatch for {}: {} vs {}".format(k, arr.shape, self._sd[k].shape))
                self._sd[k] = np.ascontiguousarray(arr)
        self._dirty = True

    @staticmethod
    def _ignorable(key):
        """Keys of a reference checkpoint this implementation has no use for: BatchNorm buffers (train-mode BN at inference
        never reads running statistics, network.py:553-556), the ReID classifier heads (`plain` output discards them,
        resnet.py:319-322) and the optimiser-side cls_token."""
        parts = key.split(".")
        return (parts[-1] in ("running_mean", "running_var", "num_batches_tracked") or "fc" in parts or "fc_person" in parts
                or key == "cls_token")

    def _load_checked(self, sd, what, scope="", skip_reid=False):
        """The reference's `model_dict.update(ckpt); load_state_dict(model_dict)` (network.py:465-467, load_trained_net.py
        :64-66): keys the model does not have make the strict load RAISE (wrong flavour, DDP `module.` prefix, ...) instead
        of being dropped; parameters the checkpoint does not supply keep their current value, which is reported."""
        unexpected = [k for k in sd if k not in self._sd and not self._ignorable(k)]
        if unexpected:
            raise RuntimeError("Error(s) in loading state_dict for BUSCA from {}: unexpected key(s) {}{}".format(
                what, unexpected[:8], " ..." if len(unexpected) > 8 else ""))
        absent = [k for k in self._sd if k.startswith(scope) and k not in sd and not (skip_reid and k.startswith(_REID_PREFIX))]
        if absent:
            import warnings
            warnings.warn("{}: {} parameter(s) not in the checkpoint keep their current (randomly initialised) values: {}{}".format(
                what, len(absent), absent[:6], " ..." if len(absent) > 6 else ""))
        self.load_state_dict({k: v for k, v in sd.items() if k in self._sd}, strict=False)

    @property
    def num_params(self):
        return int(sum(v.size for v in self._sd.values()))

    num_trainable_params = num_params

    @staticmethod
    def _read_checkpoint(path):
        ck = torch.load(path, map_location=torch.device("cpu"))
        return ck["model_state_dict"] if "model_state_dict" in ck else ck

    def load_pretrained(self, path, ignore_reid=False, ignore_reid_fc=False):
        """busca/network.py:432-467: raw state_dict or {'model_state_dict': ...}; optional dropping of the
        ReID classifier / the whole ReID; keys this model lacks (cls_token, fc, BN buffers) are ignored."""
        sd = self._read_checkpoint(path)
        if ignore_reid_fc:
            sd = {k: v for k, v in sd.items() if "reid_encoder.model.fc." not in k and "reid_encoder.model.fc_person." not in k}
        if ignore_reid:
            sd = {k: v for k, v in sd.items() if "reid_encoder.model." not in k}
        if "cls_token" in sd:
            print("WARNING: Loading a model with a cls_token, but the current model does not have a cls_token. The cls_token will be ignored")
        if ignore_reid:                     # the ReID keys were dropped on purpose: only the Decision-Transformer part is checked
            self._load_checked(sd, what="load_pretrained(%r)" % (path,), skip_reid=True)
            return
        self._load_checked(sd, what="load_pretrained(%r)" % (path,))

    # ---- device state ---------------------------------------------------------------------------------------
    def _sync(self):
        if self._ctx is None:
            # one busca_ctx per model: a context holds ONE Decision-Transformer and ONE ReID weight set, so two BUSCA objects
            # on one GPU (the reference's modules are independent) must not share one
            self._ctx = _lib.Context(self._device_index)
        if self._dirty:
            dt_sd = {k: v for k, v in self._sd.items() if not k.startswith(_REID_PREFIX)}
            self._dt = DecisionTransformerHIP(self._ctx, dt_sd, activation=self.effective_activation,
                                              fake_bbox_f64=self.pinned_numpy, precision=self.precision,
                                              input_flavour=self.args.input_flavour, nhead=int(self.args.nhead),
                                              encode_separator_as_reference=bool(self.args.encode_separator_as_reference))
            self._reid = ReIDEncoderHIP(self._ctx, self._sd, prefix=_REID_PREFIX, precision=self.reid_precision)
            self._dirty = False
        return self._ctx

    def reserve(self, max_lost, seq_len=11, num_candidates=5, max_detections=None):
        """Allocate every workspace a step of up to `max_lost` lost tracks can need NOW (ReID: max_lost x seq_len memory crops on the
        current stream, max_lost x num_candidates (or max_detections + max_lost distinct) candidate crops on the side stream; layer-wise
        Decision-Transformer buffers), so that no association step synchronises the device to grow one (busca_reid_reserve /
        busca_dt_reserve).  Optional: without it the first step of a larger size pays one hipMalloc."""
        self._sync()
        dev = self._dev()
        n_can = max_lost * num_candidates if max_detections is None else min(max_lost * num_candidates, max_detections + max_lost + 1)
        self._reid.reserve(max_lost * seq_len)
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = _side_stream_of(dev)
        self._reid.reserve(max(1, n_can), stream=self._side_stream.cuda_stream)
        self._dt.reserve(max_lost, seq_len, num_candidates)

    def _dev(self):
        return torch.device("cuda", self._device_index)

    def _to_u8_hwc_bgr(self, x):
        """Accept u8 BGR [...,384,128,3] or the reference's normalised float RGB [...,3,384,128]; the latter is
        mapped back to the u8 crop it was made from (the normalisation is injective on 0..255)."""
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x))
        x = x.to(self._dev())
        if x.dtype == torch.uint8:
            return x.reshape(-1, 384, 128, 3).contiguous()
        x = x.reshape(-1, 3, 384, 128).double()
        mean = torch.tensor(_PIX_MEAN_RGB, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(_PIX_STD_RGB, device=x.device).view(1, 3, 1, 1)
        u8 = torch.round((x * std + mean) * 255.0).clamp_(0, 255).to(torch.uint8)
        return u8.flip(1).permute(0, 2, 3, 1).contiguous()          # RGB CHW -> BGR HWC

    def _reid_features(self, x):
        self._sync()
        return self._reid.forward(self._to_u8_hwc_bgr(x))

    def _reid_pair(self, mem_u8, can_u8):
        """The two BatchNorm batches of a step (memory crops, candidate crops; network.py:192-193) are independent:
        the candidate batch runs on a side stream, concurrently with the memory batch (per-stream workspaces in the
        library).  Small batches are latency-bound, so this nearly halves their ReID time."""
        return self._reid_join(self._reid_side_start(can_u8), self._reid.forward(mem_u8))

    def _reid_side_start(self, u8, zero_norm=None, weights=None):
        """Enqueue one BatchNorm batch on the side stream (ordered after everything already on the current stream)."""
        dev = self._dev()
        cur = torch.cuda.current_stream(dev)
        if getattr(self, "_side_stream", None) is None:
            # ONE side stream per device and process, shared by every model: HIP deals streams to its few hardware queues
            # round-robin, and a side stream that lands on the queue of the current stream serialises the two ReID passes of a
            # step (4.4 -> 5.9 ms; seen on about one model instance in four when every instance made its own stream - a
            # high-priority stream was worse, 7.8 ms).  The first stream a process creates sits next to the default queue.
            self._side_stream = _side_stream_of(dev)
        side = self._side_stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            feat = self._reid.forward(u8, stream=side.cuda_stream, zero_norm=zero_norm, weights=weights)
        u8.record_stream(side)
        if zero_norm is not None:
            zero_norm.record_stream(side)
        return feat

    def _reid_join(self, side_feat, cur_feat):
        """Make the current stream wait for the side-stream batch; returns (cur_feat, side_feat)."""
        cur = torch.cuda.current_stream(self._dev())
        cur.wait_stream(self._side_stream)
        side_feat.record_stream(cur)
        return cur_feat, side_feat

    # ---- forward ------------------------------------------------------------------------------------------------
    def forward(self, embeddings_memory, candidate_embedding, memory_bboxes=None, candidates_bboxes=None,
                return_att=False, return_logits=False, plot_results=False):
        """busca/network.py:176-244.  Images: u8 BGR [B,n,384,128,3] or normalised float RGB [B,n,3,384,128];
        boxes ltrb [B,n,4].  Returns the pre-softmax logits [B,P+2] (a cuda tensor)."""
        if plot_results:
            raise NotImplementedError("plot_results needs the reference's OpenCV visualisation")
        self._sync()
        B, L = int(embeddings_memory.shape[0]), int(embeddings_memory.shape[1])
        P = int(candidate_embedding.shape[1])
        mem_feat, can_feat = self._reid_pair(self._to_u8_hwc_bgr(embeddings_memory), self._to_u8_hwc_bgr(candidate_embedding))
        mem_feat, can_feat = mem_feat.view(B, L, -1), can_feat.view(B, P, -1)     # two BN batches (network.py:192-193)
        out = self._dt.forward(mem_feat, can_feat, memory_bboxes, candidates_bboxes,
                               want_hidden=return_logits, want_att=return_att)
        self._last = out
        if return_att:
            self.attentions = [out["att"][i] for i in range(out["att"].shape[0])]
        if return_logits:
            pos = self._dt.can_positions(L, P)
            self.logits = out["hidden"][:, pos]
            self.mem_logits = out["hidden"][:, :L].mean(dim=1)
        return out["logits"]

    __call__ = forward

    def forward_features(self, mem_feat, can_feat, memory_bboxes, candidates_bboxes, return_att=False, return_logits=False):
        """Decision-Transformer step on precomputed 512-d ReID features (the 'DT-step' of bench.py)."""
        self._sync()
        out = self._dt.forward(mem_feat, can_feat, memory_bboxes, candidates_bboxes, want_hidden=return_logits, want_att=return_att)
        self._last = out
        return out

    # ---- batching of tracker objects (busca/network.py:282-429) ------------------------------------------
    def associate_embeddings(self, tracks_embeddings, dets_embeddings, dists_matrix, seq_len, num_candidates,
                             use_broader_memory, select_highest_candidate, highest_candidate_minimum_thresh=None,
                             keep_highest_value=False, extra_kalman_candidates=[], plot_results=False, normalize_ims=False):
        job = self._assoc_prepare(tracks_embeddings, dets_embeddings, dists_matrix, seq_len, num_candidates, use_broader_memory,
                                  select_highest_candidate, highest_candidate_minimum_thresh, keep_highest_value,
                                  extra_kalman_candidates, plot_results, normalize_ims)
        if job is None:
            return None, None
        mem_feat, can_feat = self._assoc_features(job)
        out = self._dt.forward(mem_feat, can_feat, job["mem_ltrb"], job["can_ltrb"], want_hidden=self.store_logits)
        return self._assoc_finish(job, out)

    def _assoc_prepare(self, tracks_embeddings, dets_embeddings, dists_matrix, seq_len, num_candidates, use_broader_memory,
                       select_highest_candidate, highest_candidate_minimum_thresh=None, keep_highest_value=False,
                       extra_kalman_candidates=(), plot_results=False, normalize_ims=False):
        """Host bookkeeping of network.py:293-398 + the two ReID passes ENQUEUED (memory batch on the side stream, candidate
        batch on the current one).  Returns the job (dict) that `_assoc_features` / `_assoc_finish` complete, or None for the
        reference's early returns.  Split from the Decision-Transformer launch so that `StepBatcher` can run ONE launch for
        the steps of several trackers."""
        B, N, P, L = len(tracks_embeddings), len(dets_embeddings), int(num_candidates), int(seq_len)
        K = len(extra_kalman_candidates)
        if B == 0 or (N == 0 and K == 0):
            return None
        if plot_results:
            raise NotImplementedError("plot_results needs the reference's OpenCV visualisation")
        self._sync()

        def as_u8(img):
            img = np.asarray(img)               # DeviceCrop objects copy their real pixels back here (never placeholders)
            if img.dtype == np.uint8:
                return img
            # already-normalised float crops (normalize_ims=False callers): map back to the u8 they came from
            v = np.rint((img.astype(np.float64) * tracking._PIXEL_STD + tracking._PIXEL_MEAN) * 255.0)
            return np.clip(v, 0, 255).astype(np.uint8)

        # normalize_ims=False: the caller's crops are already normalised floats and the reference pads with float 0.0 in
        # NORMALISED space (network.py:285,306,354) - no u8 value maps there, so those crops are flagged for the extractor
        zero_is_normalised = not normalize_ims

        # crops are collected as references; zero crops are None (incomplete memory, padded candidate)
        mem_ref = [[None] * L for _ in range(B)]
        mem_box = np.empty((B, L, 4), np.float64)
        reliable = np.zeros(B, bool)
        for t, trk in enumerate(tracks_embeddings):
            hist = trk.images_mem
            idx = memory_indices(len(hist), L, use_broader_memory)
            if len(idx) == L:
                mem_ref[t] = [hist[i] for i in idx]
                mem_box[t] = np.asarray([trk.tlwh_mem[i] for i in idx], dtype=np.float64) * trk.scale
                reliable[t] = True
            else:                               # incomplete memory: zero crops, dummy box, flagged unreliable
                mem_box[t] = (250.0, 250.0, 500.0, 500.0)

        # the memory batch does not depend on the proposals: its ReID pass is enqueued NOW (side stream), so the host work
        # below (top-P selection, candidate lists) is hidden behind it
        mem_u8, mem_zn, mem_w, mem_inv = self._gather_crops(mem_ref, as_u8, zero_is_normalised)
        mem_feat_side = self._reid_side_start(mem_u8, mem_zn, mem_w)
        gathered = self.last_gather

        # top-P nearest detections per track on the GPU (ascending centre distance, ties by lower index)
        order = np.full((B, P), -1, np.int64)
        if N > 0:
            d = np.ascontiguousarray(np.asarray(dists_matrix, dtype=np.float64).reshape(B, N))
            order = geometry.topk_rows_host(self._ctx, d, P).astype(np.int64)
        miss = tracking.missing_candidate_bbox(flavour="ltwh", pinned_numpy=self.pinned_numpy).astype(np.float64)
        can_ref = [[None] * P for _ in range(B)]
        can_box = np.empty((B, P, 4), np.float64)
        can_box[:] = miss
        det_box = np.empty((N, 4), np.float64)
        for di in np.unique(order[order >= 0]):
            det = dets_embeddings[di]
            det_box[di] = np.asarray(det.tlwh_mem[-1], dtype=np.float64) * det.scale
        for t in range(B):
            for j in range(P):
                di = order[t, j]
                if di >= 0:
                    can_ref[t][j] = dets_embeddings[di].images_mem[-1]
                    can_box[t, j] = det_box[di]
        n_avail = min(N, P)
        if K > 0:                               # the track's own Kalman prediction takes slot min(N, P-1)
            n_avail = min(N + 1, P)
            slot = min(N, P - 1)
            for t in range(B):
                kd = extra_kalman_candidates[t]
                order[t, slot] = N + t
                can_box[t, slot] = np.asarray(kd.tlwh, dtype=np.float64) * kd.scale
                can_ref[t][slot] = kd.images_mem[-1]

        with np.errstate(over="ignore"):
            mem_ltrb = mem_box.astype(np.float32)
            can_ltrb = can_box.astype(np.float32)
            mem_ltrb[..., 2:] += mem_ltrb[..., :2]
            can_ltrb[..., 2:] += can_ltrb[..., :2]

        can_u8, can_zn, can_w, can_inv = self._gather_crops(can_ref, as_u8, zero_is_normalised)
        self.last_gather = (gathered[0] + self.last_gather[0], gathered[1] + self.last_gather[1])
        can_feat = self._reid.forward(can_u8, zero_norm=can_zn, weights=can_w)
        return dict(B=B, N=N, K=K, P=P, L=L, order=order, n_avail=n_avail, reliable=reliable, mem_ltrb=mem_ltrb, can_ltrb=can_ltrb,
                    mem_feat_side=mem_feat_side, can_feat=can_feat, mem_inv=mem_inv, can_inv=can_inv, select=bool(select_highest_candidate),
                    mem_in=(mem_u8, mem_zn, mem_w), can_in=(can_u8, can_zn, can_w),      # the two BatchNorm batches themselves: an x3 pass that overflowed is run again in f32
                    thresh=highest_candidate_minimum_thresh, keep=bool(keep_highest_value))

    def _assoc_features(self, job):
        """Join the side-stream ReID batch: (mem_feat [B,L,512], can_feat [B,P,512]), the two BN batches of network.py:192-193."""
        can_feat, mem_feat = self._reid_join(job["mem_feat_side"], job["can_feat"])
        return self._assoc_slots(job, mem_feat, can_feat)

    def _assoc_slots(self, job, mem_feat, can_feat):
        dev = mem_feat.device                  # features of the distinct crops -> the [B, L] / [B, P] slots they stand for
        if len(job["mem_inv"]) != mem_feat.shape[0] or (job["mem_inv"] != np.arange(len(job["mem_inv"]))).any():
            mem_feat = mem_feat[geometry.h2d_async(job["mem_inv"], dev)]
        if len(job["can_inv"]) != can_feat.shape[0] or (job["can_inv"] != np.arange(len(job["can_inv"]))).any():
            can_feat = can_feat[geometry.h2d_async(job["can_inv"], dev)]
        return mem_feat.view(job["B"], job["L"], -1), can_feat.view(job["B"], job["P"], -1)

    def _assoc_exact_reid(self, job):
        """The two BatchNorm batches of `job` on the exact-f32 extractor (its own busca_ctx: a context holds ONE ReID weight set; created on first use) and the
        Decision Transformer on those features - what `reid_precision="f32"` computes for the same step."""
        if getattr(self, "_reid_exact", None) is None:
            self._reid_exact = ReIDEncoderHIP(_lib.Context(self._device_index), self._sd, prefix=_REID_PREFIX, precision="f32")
        self.reid_exact_reruns = getattr(self, "reid_exact_reruns", 0) + 1
        (mu8, mzn, mw), (cu8, czn, cw) = job["mem_in"], job["can_in"]
        mem_feat = self._reid_exact.forward(mu8, zero_norm=mzn, weights=mw)
        can_feat = self._reid_exact.forward(cu8, zero_norm=czn, weights=cw)
        mem_feat, can_feat = self._assoc_slots(job, mem_feat, can_feat)
        self._ctx.set_option("dt_status", 0)            # (whatever the discarded forward on the invalid features left behind)
        out = self._dt.forward(mem_feat, can_feat, job["mem_ltrb"], job["can_ltrb"], want_hidden=self.store_logits)
        torch.cuda.current_stream(self._dev()).synchronize()
        return out

    def _assoc_finish(self, job, out, reid_overflow=None):
        """network.py:403-429: probabilities -> [B, N_det (+B)] matrix (one-hot / thresholded / raw) + reliability flags.
        `out`: dict(probs [B,P+2], argmax [B], hidden?) of this job's tracks."""
        B, N, K, P, L = job["B"], job["N"], job["K"], job["P"], job["L"]
        self._last = out
        if self.store_logits and "hidden" in out:
            pos = self._dt.can_positions(L, P)
            self.logits = out["hidden"][:, pos]
            self.mem_logits = out["hidden"][:, :L].mean(dim=1)
        probs = out["probs"].cpu().numpy().astype(np.float64)                 # (synchronises the forward's stream)
        if reid_overflow is None:
            reid_overflow = self._reid.take_status()
        if reid_overflow:                       # an x3 ReID pass of this step staged an activation beyond the split-fp16 range: both batches again, exact f32
            out = self._assoc_exact_reid(job)
            probs = out["probs"].cpu().numpy().astype(np.float64)
        fixed = self._dt.settle(out)            # an x3 forward that clipped an operand is run again in exact float32: the tracker never gets a clipped step
        if fixed is not out:
            out = self._last = fixed
            probs = out["probs"].cpu().numpy().astype(np.float64)
            if self.store_logits and "hidden" in out:
                self.logits = out["hidden"][:, self._dt.can_positions(L, P)]
                self.mem_logits = out["hidden"][:, :L].mean(dim=1)
        best = out["argmax"].cpu().numpy()
        cols = N if K == 0 else N + K
        probs_matrix = np.zeros((B, cols))
        rows = np.arange(B)
        if job["select"]:
            top = probs[rows, best]
            th = job["thresh"]
            ok = np.ones(B, bool) if (th is None or th == 0) else ((th > 0.0) & (top >= th))
            picked = np.zeros_like(probs)
            picked[rows[ok], best[ok]] = top[ok] if job["keep"] else 1.0
            probs = picked
        order, n_avail = job["order"], job["n_avail"]
        for t in range(B):
            probs_matrix[t, order[t, :n_avail]] = probs[t, :n_avail]
        return probs_matrix, job["reliable"]

    def _gather_crops(self, refs, as_u8, zero_is_normalised=False):
        """[B][n] crop references (None = all-zero crop) -> (cuda u8 [U,384,128,3], zero flags | None, multiplicities [U] | None,
        inverse [B*n]) with ONE index-gather launch (busca_gather_crops): crops that still own a slot of the device crop pool
        are read where they are; the rest (plain arrays, spilled slots) go through one host batch first.  Replaces
        `_get_track_mem` + np.array stacking + H2D of network.py:247-279,313-316,383-386.
        Repeated crops (the same detection among the P nearest of several tracks, network.py:340-358; all zero crops) are
        gathered ONCE: the extractor computes each distinct crop once and weights the batch statistics by its multiplicity."""
        dev = self._dev()
        flat = [r for row in refs for r in row]
        keys = np.zeros(len(flat), np.uint64)                # identity of every entry: pool address, or a host-object tag
        host_first, host_objs = {}, []
        for i, r in enumerate(flat):
            if r is None:
                continue
            slot = getattr(r, "slot", None)
            if slot is not None and slot.ptr and slot.pool.device == dev:
                keys[i] = slot.ptr
            else:
                tag = host_first.get(id(r))
                if tag is None:
                    tag = len(host_objs) + 1                 # small integers never collide with device addresses
                    host_first[id(r)] = tag
                    host_objs.append(r)
                keys[i] = tag
        if self.dedup_crops:
            # distinct crops in order of FIRST APPEARANCE (np.unique sorts by key = address): the batch the extractor sees is
            # then the same whether a crop came from the device pool or from the host, so both routes stay bit-identical
            uniq, first, inverse, counts = np.unique(keys, return_index=True, return_inverse=True, return_counts=True)
            order = np.argsort(first, kind="stable")
            rank = np.empty_like(order)
            rank[order] = np.arange(len(order))
            uniq, counts, inverse = uniq[order], counts[order], rank[inverse]
        else:
            uniq, inverse, counts = keys, np.arange(len(keys)), np.ones(len(keys), np.int64)
        ptrs = uniq.copy()
        staged = None
        is_host = (uniq > 0) & (uniq <= len(host_objs))
        if is_host.any():
            staged = torch.from_numpy(np.stack([as_u8(host_objs[int(t) - 1]) for t in uniq[is_host]])).to(dev)
            ptrs[is_host] = staged.data_ptr() + np.arange(int(is_host.sum()), dtype=np.uint64) * np.uint64(384 * 128 * 3)
        out = geometry.gather_crops(self._ctx, ptrs)
        if staged is not None:
            staged.record_stream(torch.cuda.current_stream(dev))
        n_host = int(sum(counts[is_host])) if is_host.any() else 0
        self.last_gather = (len(flat) - n_host - int((keys == 0).sum()), n_host)     # (device-resident, host) crops of the batch
        self.last_unique = (len(uniq), len(flat))
        zn = None
        if zero_is_normalised and (uniq == 0).any():
            zn = geometry.h2d_async((uniq == 0).astype(np.uint8), dev)
        wts = counts.astype(np.float32) if (counts > 1).any() else None
        return out, zn, wts, inverse

    # ---- helpers with the reference's names ------------------------------------------------------------------
    def _get_track_mem(self, track, seq_len, use_broader_memory):
        idx = memory_indices(len(track.images_mem), seq_len, use_broader_memory)
        return [track.images_mem[i] for i in idx], np.array([track.tlwh_mem[i] for i in idx]) * track.scale

    def _normalize_embeddings_batch(self, embeddings_batch):
        return tracking.normalize_crops(embeddings_batch)

    @staticmethod
    def ltwh_to_ltrb(ltwh):
        ret = torch.clone(ltwh)
        ret[..., 2:] += ret[..., :2]
        return ret

    def begin_frame(self, image):
        """Optional, not part of the reference's surface: upload `image` once for all get_image_crops calls of one tracker update (the adapters make
        2-3 per frame, byte_tracker.py:280-282,475; StrongSORT one per detection).  Until end_frame(), calls that pass this very object reuse the
        upload; the caller must not edit the array in between.  Without it every call uploads the live array (always exact, +0.16 ms per extra call
        at 1080p)."""
        self._sync()
        geometry.begin_frame(self._ctx, image)

    def end_frame(self):
        geometry.end_frame(self._ctx)

    def frame(self, image):
        """`with model.frame(image): tracker.update(...)`"""
        self._sync()
        return geometry.frame_scope(self._ctx, image)

    def get_image_crops(self, image, bboxes, output_size=None, normalize=True):
        """busca/network.py:492-507: all boxes of a frame cut, padded and resized on the GPU in one launch."""
        self._sync()
        return tracking.get_image_crops(image, bboxes, normalize=normalize, ctx=self._ctx,
                                        host_copy="never" if self.device_only_crops else self.crop_host_copy, output_size=output_size)


class ReID_Encoder(_ReIDFacade):
    """Name kept for importers of busca.network.ReID_Encoder."""
