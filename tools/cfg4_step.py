#!/usr/bin/env python3
"""BASELINE configs[3]: MOT20-like dense crowd, 128 lost tracks x 32 proposals, crops cut ON THE GPU.

One full step = crop gather of the frame's detections (busca_crop_gather) -> index gather of the two crop batches
(busca_gather_crops: 128 x 11 memory crops from the device pool, 128 x 32 candidate crops out of the frame's detections) ->
ReID over the two BatchNorm batches (1 408 and 4 096 crops, two streams) -> Decision Transformer (T = 79 tokens per track,
layer-wise path) -> probabilities.  The 4 096 candidate slots are filled from the frame's ~160 detections (every track takes
its 32 nearest): with dedup (default, what BUSCA.associate_embeddings does) the extractor computes each distinct detection crop
once and weights the BatchNorm statistics by its multiplicity (busca_reid_forward_w); dedup=False runs the expanded batch
exactly as the reference builds it.  python tools/cfg4_step.py [steps] [dt_precision] [nodedup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from busca_amd import _lib, geometry, synth
from busca_amd.dt import DecisionTransformerHIP
from busca_amd.reid import ReIDEncoderHIP

REID_GFLOP_PER_CROP = 8.01


def run(steps=3, precision="f16", B=128, L=11, P=32, n_det=160, d=512, check=False, seed=7, dedup=True, reid_precision="f16"):
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    reid = ReIDEncoderHIP(ctx, synth.reid_state_dict(seed), precision=reid_precision)
    sd = synth.dt_state_dict(seed, d=d, ff=2 * d)
    dt = DecisionTransformerHIP(ctx, sd, precision=precision)
    dt.reserve(B, L, P)
    frame = torch.from_numpy(synth.randint_u8(seed, "frame", (1080, 1920, 3))).to(dev)
    rng = np.random.default_rng(seed)
    h = rng.uniform(80, 320, n_det); w = h * rng.uniform(0.3, 0.5, n_det)
    x = rng.uniform(0, 1920 - 100, n_det); y = rng.uniform(0, 1080 - 200, n_det)
    det_tlbr = np.stack([x, y, x + w, y + h], 1)
    # track memories: L earlier crops per lost track, resident in the device pool (cut from earlier frames)
    pool = geometry.crop_pool(ctx)
    mem_slots = pool.alloc(B * L)
    jit = rng.normal(0, 3, (B * L, 4))
    mem_tlbr = det_tlbr[rng.integers(0, n_det, B * L)] + jit
    geometry.crop_gather(ctx, frame, mem_tlbr, want_u8=False, dst_ptrs=np.array([s.ptr for s in mem_slots], dtype=np.uint64))
    mem_ptrs = np.array([s.ptr for s in mem_slots], dtype=np.uint64)
    order = np.stack([rng.permutation(n_det)[:P] for _ in range(B)])                 # each track's P nearest detections
    inp = synth.dt_inputs(seed, B, L, P)
    mb, cb = torch.from_numpy(inp["mem_boxes"]).to(dev), torch.from_numpy(inp["can_boxes"]).to(dev)
    from busca_amd.network import _side_stream_of
    side = _side_stream_of(dev)               # the process-wide side stream (see busca_amd/network.py)

    def one():
        det_u8, _ = geometry.crop_gather(ctx, frame, det_tlbr, want_u8=True)          # this frame's detections, cut on the GPU
        base = det_u8.data_ptr()
        mem_u8 = geometry.gather_crops(ctx, mem_ptrs)
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            mf = reid.forward(mem_u8, stream=side.cuda_stream).view(B, L, -1)
        mem_u8.record_stream(side)
        flat = order.reshape(-1)
        if dedup:                                  # distinct detections in order of first appearance + multiplicities
            uniq, first, inverse, counts = np.unique(flat, return_index=True, return_inverse=True, return_counts=True)
            o = np.argsort(first, kind="stable"); rank = np.empty_like(o); rank[o] = np.arange(len(o))
            uniq, counts, inverse = uniq[o], counts[o], rank[inverse]
            can_u8 = geometry.gather_crops(ctx, (base + uniq.astype(np.uint64) * np.uint64(384 * 128 * 3)).astype(np.uint64))
            cf = reid.forward(can_u8, weights=counts)[torch.from_numpy(inverse).to(dev)].view(B, P, -1)
        else:
            can_u8 = geometry.gather_crops(ctx, (base + flat.astype(np.uint64) * np.uint64(384 * 128 * 3)).astype(np.uint64))
            cf = reid.forward(can_u8).view(B, P, -1)
        cur.wait_stream(side)
        out = dt.forward(mf, cf, mb, cb)
        return out, mf, cf

    out, mf, cf = one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out, mf, cf = one()
    torch.cuda.synchronize(dev)
    el = (time.perf_counter() - t0) / steps
    crops = B * (L + P)
    computed = B * L + (len(np.unique(order)) if dedup else B * P)
    res = {"workload": "cfg4 full step: %d lost x %d proposals x d%d, %d + %d crop slots cut/gathered on the GPU (%d detections), ReID (%s) + DT (%s)%s"
                       % (B, P, d, B * L, B * P, n_det, reid_precision, precision, "; repeated candidate crops computed once, statistics weighted" if dedup else "; expanded candidate batch"),
           "value": 1.0 / el, "unit": "steps/s", "ms_per_step": el * 1e3, "crops_per_step": crops, "crops_computed": computed, "steps": steps,
           "reid_tflops_executed": computed * REID_GFLOP_PER_CROP / el / 1e3, "frac_of_f16_mfma_peak_executed": computed * REID_GFLOP_PER_CROP / el / 1e3 / 2500.0,
           "dtype": reid_precision, "dt_dtype": precision}
    if check:
        res["_out"] = {k: v.cpu().numpy() for k, v in out.items()}
        res["_feat"] = (mf.cpu().numpy(), cf.cpu().numpy())
        res["_inp"] = inp
        res["_sd"] = sd
    ctx._crop_pool = None
    return res


if __name__ == "__main__":
    r = run(int(sys.argv[1]) if len(sys.argv) > 1 else 3, sys.argv[2] if len(sys.argv) > 2 else "f16", dedup="nodedup" not in sys.argv)
    print({k: v for k, v in r.items() if not k.startswith("_")})
