"""Cross-sequence step batcher: the association steps of several tracker instances (sequences sharded onto ONE GPU) in
one Decision-Transformer launch.

A single `BUSCA.associate_embeddings` call hands the GPU B lost tracks = B workgroups of the fused kernel - 32 of the 256
CUs at the north-star shape.  Tracks are independent inside the Decision Transformer (attention never crosses tracks,
busca/custom_layers.py:30-41; `tests/test_dt_gpu.py::test_dt_batch_invariance`), so the tracks of S concurrent steps can
share a launch.  The ReID extractor is different: its train-mode BatchNorm statistics span a batch
(busca/network.py:553-556), so every step keeps its OWN two BN batches - the batcher never merges those; it only lets
them queue back to back on the two ReID streams.

    batcher = StepBatcher(model)
    t1 = batcher.submit(tracks_a, dets_a, dists_a, 11, 5, True, True, extra_kalman_candidates=kal_a, normalize_ims=True)
    t2 = batcher.submit(tracks_b, dets_b, dists_b, 11, 5, True, True, extra_kalman_candidates=kal_b, normalize_ims=True)
    batcher.flush()
    probs_a, reliable_a = t1.result()        # bit-identical to model.associate_embeddings(tracks_a, ...)

Bit-identical because the kernel FLAVOUR of a merged launch is pinned to what each step would get alone: the f16 fused kernel
switches to two tracks per workgroup from 257 tracks on (where that needs fewer rounds of workgroups) (agrees with the one-track flavour only to f16 rounding, ~3e-3 in the
logits), so the merged launch of steps that are each below that size is forced to the one-track flavour (`dt_ntrk` option of the
context) and a step that is itself larger runs in its own launch; the row-tile geometry of the layer-wise path also follows the
merged size, but its GEMMs sum every row in the same order whatever the tile (`tests/test_bench_gpu.py`, cfg5 split leg).

`submit` takes exactly the arguments of `associate_embeddings` (busca/network.py:282); `result()` flushes if needed.
"""
import torch


class Ticket:
    def __init__(self, batcher, job):
        self._batcher, self._job, self._value, self._done = batcher, job, (None, None), job is None

    def result(self):
        if not self._done:
            self._batcher.flush()
        return self._value


class StepBatcher:
    def __init__(self, model):
        self.model = model
        self._pending = []
        self.launches = 0                   # Decision-Transformer launches issued so far
        self.steps = 0                      # association steps served so far

    def submit(self, *args, **kwargs):
        """Queue one association step (arguments of BUSCA.associate_embeddings).  Its crops are gathered and its two ReID
        passes enqueued immediately; the Decision Transformer waits for `flush`."""
        job = self.model._assoc_prepare(*args, **kwargs)
        t = Ticket(self, job)
        if job is not None:
            self._pending.append(t)
        return t

    def flush(self):
        """Run every queued step: one busca_dt_forward per (L, P) group, then the per-step scatter."""
        m = self.model
        pending, self._pending = self._pending, []
        groups = {}
        for t in pending:
            groups.setdefault((t._job["L"], t._job["P"]), []).append(t)
        launches = []
        for (L, P), ts in groups.items():
            small = [t for t in ts if t._job["B"] <= 256]
            if small:
                launches.append((small, 1))                       # one-track flavour, as each of these steps would run alone
            launches.extend(([t], 0) for t in ts if t._job["B"] > 256)
        for ts, ntrk in launches:
            feats = [m._assoc_features(t._job) for t in ts]
            dev = feats[0][0].device
            mem_feat = torch.cat([f[0] for f in feats], 0)
            can_feat = torch.cat([f[1] for f in feats], 0)
            mem_ltrb = torch.cat([torch.from_numpy(t._job["mem_ltrb"]) for t in ts], 0).pin_memory().to(dev, non_blocking=True)
            can_ltrb = torch.cat([torch.from_numpy(t._job["can_ltrb"]) for t in ts], 0).pin_memory().to(dev, non_blocking=True)
            prev = m._ctx.get_option("dt_ntrk")                   # a pin (BUSCA_DT_NTRK / an earlier set_option) outranks the batcher's choice and survives it
            m._ctx.set_option("dt_ntrk", prev if prev != 0 else ntrk)
            try:
                out = m._dt.forward(mem_feat, can_feat, mem_ltrb, can_ltrb, want_hidden=m.store_logits)
            finally:
                m._ctx.set_option("dt_ntrk", prev)
            self.launches += 1
            torch.cuda.current_stream(dev).synchronize()
            out = m._dt.settle(out)             # a merged x3 launch that clipped an operand is run again in exact float32 before it is handed out
            reid_overflow = m._reid.take_status()       # (one status word per context: every step of this flush is suspect when an x3 ReID pass overflowed)
            lo = 0
            for t in ts:
                hi = lo + t._job["B"]
                part = {k: v[lo:hi] for k, v in out.items()}
                t._value, t._done = m._assoc_finish(t._job, part, reid_overflow), True
                lo = hi
                self.steps += 1
