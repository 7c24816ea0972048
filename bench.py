#!/usr/bin/env python3
"""bench.py - BUSCA association-step throughput on MI355X (contract in the task statement / DESIGN.md section 5).

One "step" = one pass of the hot path over one frame's batch: `--lost` lost tracks x `--proposals` proposals
(default: the north-star shape 32 x 16, d=256, L=11, 4 layers, 4 heads, ff=512) from ReID features that are already
resident in HBM to logits/probs/argmax ("DT-step", SURVEY.md 8d - the only variant for which >= 10k steps/s is
physically possible; a full step adds 864 crops x 8 GFLOP of ReID and is reported separately under `full_step`).
`--inflight F` independent steps (frames of F different sequences sharded onto this GPU) are handed to the C-ABI in
one call, i.e. one launch processes F steps and fills the 256 CUs; the single-frame case is always measured as
`p50_latency_ms` (F = 1, one synchronised call per step).

    python bench.py                       # 1 GPU, defaults finish in about a minute
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE short JSON line on stdout (the contract keys + `roofline` + `cpu_baseline` + `variants` reduced to
{value, dtype, frac}; at most MAX_LINE characters - the driver keeps only a few KB of stdout) and writes everything it measured
(the per-config legs, full steps, end-to-end latencies, HBM-kernel table, per-rank reports, notes) to `bench_detail.json` next to
this file.  Primary line = exact-f32 MFMA arithmetic (the reference computes in float32, busca/custom_layers.py:30-41); the
library's default x3 flavour and the opt-in f16 flavour are reported under `variants`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from busca_amd import synth  # noqa: E402

# MI355X_MICROARCH.md: f32-input MFMA peak (= f32 vector peak) 157.3 TFLOP/s; dense f16/bf16 MFMA 2500 TFLOP/s
PEAK_TFLOPS = {"f32": 157.3, "f16": 2500.0, "x3": 2500.0 / 3}      # x3: float32-equivalent products as three fp16 MFMAs (BUSCA_PREC_F16X3)
REID_GFLOP_PER_CROP = 8.01          # SURVEY.md 2.1 (4.005 GMAC)


def dt_step_flops(B, L, P, d, ff, E=512, nlayers=4):
    """algorithmic FLOPs of one DT-step."""
    T = L + 2 * (P + 2)
    per_layer = 2 * B * T * d * 3 * d + 4 * B * T * T * d + 2 * B * T * d * d + 4 * B * T * d * ff
    return 2 * B * (L + P) * E * d + nlayers * per_layer + 2 * B * (P + 2) * d


def dt_step_bytes(B, L, P, d, ff, E=512, nlayers=4):
    """compulsory bytes of one DT-step (fp32 I/O, weights once)."""
    W = E * d + d + nlayers * (4 * d * d + 2 * d * ff + 9 * d + ff) + 6 * d + 1
    c = 2 * int(np.ceil(d / 6))
    lut = (211 + 211 + 61) * c * 2
    return B * (L + P) * E * 4 + B * (L + P) * 16 + W * 4 + lut + B * (P + 2) * 4


def cpu_baseline(sd, cfg_kw, inp, budget_s):
    """The oracle (CPU restatement of the reference's PyTorch path, kind "port") on this host's cores.
    A short sweep picks the torch thread count that is fastest on this host (many-core hosts are slower
    with every core on matrices this small); `cores` reports the thread count actually used."""
    from oracle import dt as odt
    from oracle import encoding as enc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cfg = odt.DTConfig(**cfg_kw)
    luts = enc.build_luts(cfg.d)
    psd = odt.prepare(sd)
    tin = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in inp.items()}

    def rate(threads, seconds, max_n):
        torch.set_num_threads(threads)
        odt.dt_forward(psd, cfg, luts=luts, **tin)
        n, t0 = 0, time.perf_counter()
        while True:
            odt.dt_forward(psd, cfg, luts=luts, **tin)
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds or n >= max_n:
                return n / el, n, el

    cands = sorted({t for t in (1, 4, 8, 16, 32, 64, avail) if 1 <= t <= avail})
    sweep = {t: rate(t, budget_s * 0.08, 50)[0] for t in cands}
    best = max(sweep, key=sweep.get)
    r, n, el = rate(best, budget_s * 0.5, 5000)
    return dict(value=r, unit="steps/s", cores=best, kind="port", host_cpus=avail,
                sample="%d DT-steps of the same workload in %.1f s (oracle.dt.dt_forward, torch CPU, %d threads = fastest of sweep %s)"
                       % (n, el, best, {k: round(v, 1) for k, v in sweep.items()}))


def split_steps(k, F):
    """k steps as ceil(k/F) launches of near-equal size (20 steps, F = 8 -> 7 + 7 + 6, never a half-empty 8 + 8 + 4)."""
    if k <= 0:
        return []
    n = -(-k // F)
    base, extra = divmod(k, n)
    return [base + 1] * extra + [base] * (n - extra)


class DTRunner:
    """F steps worth of synthetic tracks resident in HBM + one loaded Decision-Transformer flavour."""

    def __init__(self, ctx, sd, precision, tens, B, L, P, F, dev):
        from busca_amd.dt import DecisionTransformerHIP
        self.ctx, self.B, self.L, self.P, self.F, self.dev = ctx, B, L, P, F, dev
        self.precision = precision
        self.model = DecisionTransformerHIP(ctx, sd, activation="relu", fake_bbox_f64=True, precision=precision)
        self.t = tens
        self.logits = torch.empty(B * F, P + 2, device=dev)
        self.probs = torch.empty_like(self.logits)
        self.amax = torch.empty(B * F, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.current_stream(dev).cuda_stream

    def launch(self, nsteps):
        self.model._ensure_loaded()
        t, lib = self.t, self.ctx.lib
        self.ctx.check(lib.busca_dt_forward(self.ctx.h, t["mem_feat"].data_ptr(), t["can_feat"].data_ptr(),
                                            t["mem_boxes"].data_ptr(), t["can_boxes"].data_ptr(), self.B * nsteps, self.L, self.P,
                                            self.logits.data_ptr(), self.probs.data_ptr(), self.amax.data_ptr(), None, None, self.stream))

    def run_steps(self, k):
        sizes = split_steps(k, self.F)
        for n in sizes:
            self.launch(n)
        return len(sizes)

    def kernel_time(self, k):
        """(total kernel ms, kernel launches, forward calls, steps) of k steps, from HIP events recorded around every kernel
        launch on the launch stream (busca_timing_*).  The fused path is one kernel per call; the layer-wise path several."""
        lib, h = self.ctx.lib, self.ctx.h
        lib.busca_timing_read(h, None, None, 1)
        lib.busca_timing_enable(h, 1)
        calls = self.run_steps(k)
        torch.cuda.synchronize(self.dev)
        avg, n = C.c_double(0), C.c_int64(0)
        lib.busca_timing_read(h, C.byref(avg), C.byref(n), 1)
        lib.busca_timing_enable(h, 0)
        return avg.value * n.value, n.value, calls, k

    def geometry(self):
        """Launch geometry of the last forward: workgroups, and how many of its tracks ran token-split (one 16-token tile per workgroup: the tracks of
        a last, partial round of one-track workgroups - a second launch inside the same bracketed region)."""
        g = {k: int(self.ctx.get_option(o)) for k, o in (("workgroups", "last_dt_grid"), ("token_split_tracks", "last_dt_split"), ("tracks_per_workgroup", "last_dt_ntrk"))}
        if g["token_split_tracks"] == 0:
            g.pop("token_split_tracks")
        else:
            g["tracks_per_split_workgroup"] = g.pop("tracks_per_workgroup")
        return g

    def p50_latency_ms(self, samples):
        if samples <= 0:
            return None
        for _ in range(20):
            self.launch(1)
        torch.cuda.synchronize(self.dev)
        lat = []
        for _ in range(samples):
            a = time.perf_counter()
            self.launch(1)
            torch.cuda.synchronize(self.dev)
            lat.append(time.perf_counter() - a)
        return float(np.percentile(np.array(lat), 50) * 1e3)


def roofline_obj(precision, B, L, P, d, ff, timing, bracket_ms_per_call, kernel="dt_fused_kernel", geometry=None):
    """timing = DTRunner.kernel_time(...).  achieved = algorithmic FLOPs of the steps those launches REALLY processed / the
    kernel time they took (a 6-step launch counts 6 steps)."""
    tot_ms, nk, calls, steps = timing
    flops = dt_step_flops(B, L, P, d, ff) * steps
    ach = flops / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else float("nan")
    spl = steps / max(1, calls)
    traffic, traffic_note = None, None
    try:    # HBM bytes per launch from the committed PMC run (profiles/pmc_traffic.json), same workload and F
        meta = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        key = "dt_%s_F%d_B%d_P%d_d%d" % (precision, int(round(spl)), B, P, d)
        ent = meta.get(key) if abs(spl - round(spl)) < 1e-9 else None      # only a PMC run of EXACTLY this launch shape counts
        traffic = ent["hbm_bytes_per_launch"] if ent else None
        traffic_note = ent["source"] if ent else "no PMC run of this launch shape (%s, %.2f steps per launch) in profiles/pmc_traffic.json" % (key, spl)
    except Exception as e:
        traffic, traffic_note = None, "profiles/pmc_traffic.json unreadable: %r" % (e,)
    if geometry and geometry.get("token_split_tracks"):
        kernel += " (two launches in one bracketed region: <SPLIT=false> for the whole rounds of one-track workgroups + <SPLIT=true> for the %d tracks of the last round, one token tile per workgroup)" % geometry["token_split_tracks"]
    extra = {"peak_note": "x3 = float32-equivalent GEMMs as three fp16 MFMAs per product block: peak = the dense fp16 MFMA peak / 3 (the 4 % of the FLOPs in the attention run on the f32 MFMA)"} if precision == "x3" else {}
    return {**extra, "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[precision], "launch_geometry": geometry,
            "traffic": traffic, "traffic_note": traffic_note, "kernel": kernel, "kernel_avg_ms": tot_ms / max(1, nk), "kernel_launches_per_call": nk / max(1, calls),
            "kernel_ms_per_call": tot_ms / max(1, calls), "flops_per_call": flops / max(1, calls),
            "steps_per_launch": spl, "event_bracket_ms_per_launch": bracket_ms_per_call,
            "algorithmic_bytes_per_step": dt_step_bytes(B, L, P, d, ff)}


def config_leg(ctx, dev, name, B, L, P, d, precision, F, steps, seed=7):
    """One BASELINE config as its own DT-step measurement (outside the contract's timed region): value + roofline."""
    ff = 2 * d
    sd = synth.dt_state_dict(seed, d=d, ff=ff)
    big = synth.dt_inputs(seed, B * F, L, P)
    tens = {k: torch.from_numpy(v).to(dev) for k, v in big.items()}
    run = DTRunner(ctx, sd, precision, tens, B, L, P, F, dev)
    run.run_steps(max(F, steps // 10))
    torch.cuda.synchronize(dev)
    a = time.perf_counter()
    calls = run.run_steps(steps)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - a
    timing = run.kernel_time(min(steps, 20 * F))
    T = L + 2 * (P + 2)
    fused = timing[1] == timing[2]
    return {"workload": "%s DT-step: %d lost x %d proposals x d%d (L=%d, T=%d, ff=%d)" % (name, B, P, d, L, T, ff), "dtype": precision,
            "value": steps / el, "unit": "steps/s", "steps": steps, "ms_per_step": el / steps * 1e3, "steps_in_flight_per_launch": F,
            "path": "fused (one kernel per call)" if fused else "layer-wise (%d kernels per call)" % round(timing[1] / max(1, timing[2])),
            "roofline": roofline_obj(precision, B, L, P, d, ff, timing, el / calls * 1e3,
                                     kernel="dt_fused_kernel" if fused else "dt_bucket_ids + dtl_gemm<EMBED> + 4 x (dtl_qkv_attn + dtl_ffn) + dtl_decoder (whole forward, every launch bracketed)",
                                     geometry=run.geometry() if fused else None)}


def full_step(ctx, dt_model, B, L, P, n_steps, dev, n_det=None, reid_precision="f16"):
    """ReID (two train-mode-BN batches: B*L memory crops, B*P candidate crops, u8 resident in HBM) + DT.
    reid_precision "f32" = the exact float32 convs (the reference's arithmetic, busca/reid/resnet.py:266-322); the roofline is
    then priced against the f32 MFMA peak.
    n_det=None: every candidate slot holds a different crop (worst case).  n_det=k: the B*P candidate slots are filled from k
    distinct detection crops, as a tracker's are (each track takes its P nearest detections, network.py:340-358); the extractor
    then computes each distinct crop once and weights the BatchNorm statistics by its multiplicity (busca_reid_forward_w)."""
    from busca_amd.reid import ReIDEncoderHIP
    reid = ReIDEncoderHIP(ctx, synth.reid_state_dict(7), precision=reid_precision)
    mem = torch.from_numpy(synth.randint_u8(11, "mem", (B * L, 384, 128, 3))).to(dev)
    inverse_d, counts = None, None
    if n_det is None:
        can = torch.from_numpy(synth.randint_u8(12, "can", (B * P, 384, 128, 3))).to(dev)
    else:
        can = torch.from_numpy(synth.randint_u8(12, "can", (n_det, 384, 128, 3))).to(dev)
        rng = np.random.default_rng(12)
        inverse = np.concatenate([rng.permutation(n_det)[:P] for _ in range(B)])
        u, first, inv, counts = np.unique(inverse, return_index=True, return_inverse=True, return_counts=True)
        o = np.argsort(first, kind="stable"); rank = np.empty_like(o); rank[o] = np.arange(len(o))
        can, counts, inverse_d = can[torch.from_numpy(u[o]).to(dev)].contiguous(), counts[o], torch.from_numpy(rank[inv]).to(dev)
    boxes = synth.dt_inputs(7, B, L, P)
    mb, cb = torch.from_numpy(boxes["mem_boxes"]).to(dev), torch.from_numpy(boxes["can_boxes"]).to(dev)

    from busca_amd.network import _side_stream_of
    side = _side_stream_of(dev)               # the process-wide side stream (see busca_amd/network.py)

    def one():                                           # as BUSCA._reid_pair: the two BN batches run on two streams
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            cf = reid.forward(can, stream=side.cuda_stream, weights=counts)
            cf = (cf if inverse_d is None else cf[inverse_d]).view(B, P, -1)
        mf = reid.forward(mem).view(B, L, -1)
        cur.wait_stream(side)
        return dt_model.forward(mf, cf, mb, cb)

    one()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        one()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / n_steps
    slots = B * (L + P)
    crops = B * L + int(can.shape[0])               # crops the extractor really computes
    tf = crops * REID_GFLOP_PER_CROP * 1e9 / dt / 1e12
    traffic = None
    try:    # HBM bytes of the two ReID passes from the committed PMC runs, scaled per crop from the nearest measured batch
        meta = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        per_crop = meta["reid_%s_n512" % reid_precision]["hbm_bytes_per_pass"] / 512.0
        traffic = per_crop * crops
    except Exception:
        traffic = None
    peak = PEAK_TFLOPS[reid_precision]
    arith = {"f16": "fp16", "f32": "float32", "x3": "float32 split into fp16 hi + lo, three fp16 MFMAs per product block (float32-equivalent)"}[reid_precision]
    extra = {"peak_note": "peak = fp16 MFMA peak / 3 (three MFMAs per float32-equivalent product)"} if reid_precision == "x3" else {}
    return {"value": 1.0 / dt, "unit": "steps/s", "ms_per_step": dt * 1e3, "crop_slots_per_step": slots, "crops_per_step": crops,
            "candidate_crops": "all distinct" if n_det is None else "%d slots drawn from %d detections (repeats computed once, weighted statistics)" % (B * P, n_det),
            "reid_algorithmic_tflop_per_step": crops * REID_GFLOP_PER_CROP / 1e3,
            "dtype": reid_precision, "dt_dtype": dt_model.precision,
            "roofline": dict({"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                         "traffic": traffic, "hbm_time_floor_ms": (traffic / 6.3e12 * 1e3) if traffic else None,
                         "note": "ReID convs (%s MFMA operands, f32 accumulate) + DT over the whole step; u8 crops already in HBM; traffic = PMC "
                                 "FETCH_SIZE x2 + WRITE_SIZE of a 512-crop pass (profiles/*reid_n512_pmc_traffic*.txt) scaled per crop, null "
                                 "when no PMC run of this flavour is committed" % arith}, **extra),
            "steps": n_steps}


def assoc_e2e(frames):
    """Simulated tracker frame (tools/e2e_sim.py): crops cut on the GPU, device-resident track memory, centre distances,
    BUSCA.associate_embeddings on the SHIPPED model shape (d=512, L=11, P=5; config/*/*/*.yml) - the metric's
    'p50 assoc latency'.  The UNPREFIXED keys are what a user gets by default (busca_amd.network.BUSCA: float32-equivalent x3 Decision Transformer +
    float32-equivalent x3 ReID); `f16_*` = the opt-in fast flavours (fp16 ReID + f16 DT), `f32_*` = exact-f32 ReID + f32 DT."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gc
    import e2e_sim
    out = {}
    keys = ("p50_assoc_latency_ms", "p50_crop_ms", "p50_center_distance_ms", "busca_frames_per_s", "device_resident_crops", "precision", "reid_precision")
    for lost, objs in ((32, 150), (8, 60)):
        r = e2e_sim.run(lost, objs, 5, 512, frames=frames, verbose=False)            # library defaults: precision x3, reid_precision x3
        out["lost%d_dets%d" % (lost, objs - lost)] = {k: r[k] for k in keys}
        gc.collect(); torch.cuda.empty_cache()      # the previous scenes' models / crop pools go away before the next one is timed
    try:        # opt-in fast flavours: fp16 ReID + f16 Decision Transformer (moves probabilities by up to 0.03, profiles/r04_decision_agreement.json)
        for lost, objs in ((32, 150), (8, 60)):
            r = e2e_sim.run(lost, objs, 5, 512, "f16", frames, verbose=False, reid_precision="f16")
            out["f16_lost%d_dets%d" % (lost, objs - lost)] = {k: r[k] for k in keys}
            gc.collect(); torch.cuda.empty_cache()
    except Exception as e:
        out["f16_lost32_dets118"] = {"error": repr(e)}
    try:        # the reference's own arithmetic: exact-f32 ReID + f32 Decision Transformer
        for lost, objs in ((32, 150), (8, 60)):
            r = e2e_sim.run(lost, objs, 5, 512, "f32", max(5, frames // 2), verbose=False, reid_precision="f32")
            out["f32_lost%d_dets%d" % (lost, objs - lost)] = {k: r[k] for k in ("p50_assoc_latency_ms", "busca_frames_per_s", "precision", "reid_precision")}
            gc.collect(); torch.cuda.empty_cache()
    except Exception as e:
        out["f32_lost32_dets118"] = {"error": repr(e)}
    r = e2e_sim.run(8, 60, 5, 512, frames=frames, verbose=False, device_only_crops=True)     # opt-in: crops never copied back to the host (default flavour)
    out["lost8_dets52_device_only_crops"] = {k: r[k] for k in ("p50_assoc_latency_ms", "p50_crop_ms", "busca_frames_per_s", "precision", "reid_precision")}
    gc.collect(); torch.cuda.empty_cache()
    try:        # the call pattern of the UNCHANGED StrongSORT / GHOST adapters: one get_image_crops per detection, no frame scope (each call reads the live host frame
                # and uploads the sub-frame its box needs, geometry._frame_for_rects)
        r = e2e_sim.run(8, 60, 5, 512, frames=frames, verbose=False, per_detection_crops=True)
        out["lost8_dets52_one_crop_call_per_detection"] = {k: r[k] for k in ("p50_assoc_latency_ms", "p50_crop_ms", "crop_calls_per_frame", "busca_frames_per_s")}
        gc.collect(); torch.cuda.empty_cache()
    except Exception as e:
        out["lost8_dets52_one_crop_call_per_detection"] = {"error": repr(e)}
    try:        # several trackers on one GPU: the steps of one frame interval through the StepBatcher (one DT launch), default flavour
        out["multi_sequence_4x_lost8"] = e2e_sim.run_multi(4, 8, 60, 5, 512, frames=frames)
    except Exception as e:
        out["multi_sequence_4x_lost8"] = {"error": repr(e)}
    out["config"] = ("shipped model shape d=512 ff=1024 L=11 P=5, random weights, synthetic 1080p frames; p50_crop_ms = the frame's two get_image_crops calls "
                     "(detections + Kalman boxes: ONE frame upload inside an explicit model.frame(img) scope, crop kernel, lazy host copy enqueued) until the tracker's stream is done; "
                     "UNPREFIXED keys = library defaults: float32-equivalent x3 DT + float32-equivalent x3 ReID (both split-fp16 MFMA); f16_* keys: opt-in f16 MFMA DT + fp16 ReID; "
                     "f32_* keys: float32 DT + exact-f32 ReID (reference arithmetic)")
    return out


def hbm_kernels(ctx, dev):
    """The HBM-bound kernels of the path (crop gather K1, pairwise matrices + top-P K10/K11, track-memory gather) at tracker sizes:
    algorithmic bytes / kernel time (HIP events around the launches on the launch stream, busca_timing_*) against the 8 TB/s peak.  At these sizes
    every one of them is launch-latency-bound (tens of KB to tens of MB per launch), which is what the fractions say; rocprof summaries of the same
    launches: profiles/r05_hbm_kernels_*."""
    from busca_amd import geometry
    lib, h = ctx.lib, ctx.h
    rng = np.random.default_rng(5)

    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize(dev)
        lib.busca_timing_read(h, None, None, 1); lib.busca_timing_enable(h, 1)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        avg, n = C.c_double(0), C.c_int64(0)
        lib.busca_timing_read(h, C.byref(avg), C.byref(n), 1); lib.busca_timing_enable(h, 0)
        return avg.value * n.value / reps            # ms per call (sum over the call's bracketed launches)

    def entry(kernel, workload, nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else float("nan")
        return {"kernel": kernel, "workload": workload, "algorithmic_bytes": int(nbytes), "kernel_us": ms * 1e3, "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                "frac": gbs / 8000.0, "bound": "hbm"}

    out = {}
    # K1 crop gather: 150 boxes of a 1080p frame (the detections + Kalman boxes of a 32-lost / 118-detection frame)
    frame = torch.from_numpy(synth.randint_u8(5, "frame", (1080, 1920, 3))).to(dev)
    hh = rng.uniform(80, 320, 150); ww = hh * rng.uniform(0.3, 0.5, 150)
    x = rng.uniform(0, 1920 - 170, 150); y = rng.uniform(0, 1080 - 330, 150)
    tlbr = np.stack([x, y, x + ww, y + hh], 1)
    ext = np.stack([np.floor(tlbr[:, 0]), np.floor(tlbr[:, 1]), np.ceil(tlbr[:, 2]), np.ceil(tlbr[:, 3])], 1)
    src_bytes = float(((ext[:, 2] - ext[:, 0]) * (ext[:, 3] - ext[:, 1]) * 3).sum())
    out["crop_gather_150_boxes_1080p"] = entry("crop_band_kernel", "150 boxes of a 1080p frame -> u8 [150,384,128,3]",
                                               src_bytes + 150 * 147456, timed(lambda: geometry.crop_gather(ctx, frame, tlbr, want_u8=True)))
    # track-memory gather: 864 crops (32 lost x (11 memory + 16 candidate) slots) out of a resident batch
    res = torch.from_numpy(synth.randint_u8(6, "res", (256, 384, 128, 3))).to(dev)
    ptrs = (res.data_ptr() + rng.integers(0, 256, 864).astype(np.uint64) * np.uint64(147456)).astype(np.uint64)
    out["crop_ptr_gather_864_crops"] = entry("crop_ptr_gather_kernel", "864 crops of 147 456 B gathered by address", 2 * 864 * 147456, timed(lambda: geometry.gather_crops(ctx, ptrs)))
    # K10 / K11 pairwise centre distance + top-P
    for nA, nB, P in ((128, 150, 32), (300, 1000, 16)):
        a = torch.from_numpy(rng.uniform(0, 1000, (nA, 4))).to(dev); b = torch.from_numpy(rng.uniform(0, 1000, (nB, 4))).to(dev)
        a[:, 2:] += a[:, :2]; b[:, 2:] += b[:, :2]
        out["pairwise_center_%dx%d" % (nA, nB)] = entry("pairwise_kernel", "centre distance %d x %d float64" % (nA, nB), (nA + nB) * 32 + nA * nB * 8,
                                                        timed(lambda: geometry.pairwise(ctx, a, b, 0)))
        dist = geometry.pairwise(ctx, a, b, 0)
        out["topk_rows_%dx%d_P%d" % (nA, nB, P)] = entry("topk_rows_kernel", "%d smallest of each of %d rows of %d float64" % (P, nA, nB), nA * nB * 8 + nA * P * 4,
                                                         timed(lambda: geometry.topk_rows(ctx, dist, P)))
    return out


def cfg5_split_leg(ctx, dev, rank, world, dist, red_dev, steps, seed=7):
    """BASELINE configs[4] (512 lost x 64 proposals x d512, f16 MFMA) with ONE step's tracks split over the ranks
    (SURVEY.md 8e case 2): rank r runs sharding.split_tracks(512, world, r) through busca_dt_forward with replicated weights,
    the host gathers the slices in rank order (sharding.gather_track_slices) - no collective on the data path.  value = whole
    steps per second (slowest rank).  Every rank calls this; rank 0 gets the result."""
    import hashlib
    from busca_amd import sharding
    from busca_amd.dt import DecisionTransformerHIP
    B, L, P, d = 512, 11, 64, 512
    sd = synth.dt_state_dict(seed, d=d, ff=2 * d)
    full = synth.dt_inputs(seed, B, L, P)                    # every rank builds the same step, then keeps its slice
    lo, hi = sharding.split_tracks(B, world, rank)
    t = {k: torch.from_numpy(np.ascontiguousarray(v[lo:hi])).to(dev) for k, v in full.items()}
    model = DecisionTransformerHIP(ctx, sd, activation="relu", fake_bbox_f64=True, precision="f16")
    out = None
    if hi > lo:
        model.reserve(hi - lo, L, P)

    def one():
        return model.forward(t["mem_feat"], t["can_feat"], t["mem_boxes"], t["can_boxes"]) if hi > lo else None

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(2):
        out = one()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    barrier()
    el = sharding.max_over_ranks(time.perf_counter() - t0, dist, red_dev)
    local = out["logits"].cpu().numpy() if out is not None else np.zeros((0, P + 2), np.float32)
    logits = sharding.gather_track_slices(local, dist)
    amax = sharding.gather_track_slices(out["argmax"].cpu().numpy() if out is not None else np.zeros((0,), np.int32), dist)
    if rank != 0:
        return None
    assert logits.shape == (B, P + 2) and amax.shape == (B,), (logits.shape, amax.shape)
    T = L + 2 * (P + 2)
    fl = dt_step_flops(B, L, P, d, 2 * d)
    return {"workload": "cfg5 DT-step: 512 lost x 64 proposals x d512 (L=11, T=%d), tracks of ONE step split over %d rank%s" % (T, world, "" if world == 1 else "s"),
            "dtype": "f16", "value": steps / el, "unit": "steps/s", "steps": steps, "ms_per_step": el / steps * 1e3, "n_gpus": world,
            "scaling": "strong", "track_slices": sharding.split_tracks(B, world), "collective": "none on the data path (host gather of B x (P+2) logits)",
            "logits_sha256": hashlib.sha256(np.ascontiguousarray(logits).tobytes()).hexdigest(),
            "argmax_sha256": hashlib.sha256(np.ascontiguousarray(amax).tobytes()).hexdigest(),
            "roofline": {"bound": "mfma", "achieved": fl * steps / el / 1e12, "peak": PEAK_TFLOPS["f16"] * world, "unit": "TFLOP/s",
                         "frac": fl * steps / el / 1e12 / (PEAK_TFLOPS["f16"] * world), "traffic": None,
                         "note": "whole-step wall time incl. host gather; peak = %d x the per-GPU f16 MFMA peak" % world}}


MAX_LINE = 6000          # the driver keeps about 8 KB of stdout: the contract line stays well inside that


def _short(x, sig=6):
    """Floats to `sig` significant digits, recursively (the full-precision numbers are in bench_detail.json)."""
    if isinstance(x, float):
        return float("%.*g" % (sig, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _short(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, sig) for v in x]
    return x


def contract_line(res, detail_path):
    """The ONE stdout line: the contract keys, `roofline`, `cpu_baseline`, and {value, dtype, frac} per variant / config leg."""
    out = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                               "dtype", "data", "p50_latency_ms") if k in res}
    out["config"] = res["config"]
    rf = res["roofline"]
    out["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_avg_ms", "kernel_launches_per_call",
                                          "kernel_ms_per_call", "steps_per_launch", "algorithmic_bytes_per_step", "flops_per_call") if k in rf}
    out["cpu_baseline"] = res.get("cpu_baseline")

    def brief(v):
        if not isinstance(v, dict) or "error" in v or "value" not in v:
            return {"error": str(v.get("error", "no value"))[:80]} if isinstance(v, dict) else None
        b = {"value": v["value"], "dtype": v.get("dtype"), "frac": (v.get("roofline") or {}).get("frac")}
        if v.get("n_gpus", 1) != 1:
            b["n_gpus"] = v["n_gpus"]
        return b

    if res.get("variants"):
        out["variants"] = {k: brief(v) for k, v in res["variants"].items()}
    if res.get("configs"):
        out["configs"] = {k: brief(v) for k, v in res["configs"].items()}
    for k in ("full_step", "full_step_f32"):
        if isinstance(res.get(k), dict) and "value" in res[k]:
            out.setdefault("configs", {})[k] = brief(res[k])
    if detail_path:
        out["detail"] = os.path.basename(detail_path)
    out = _short(out)
    line = json.dumps(out, separators=(",", ":"))
    for drop in ("configs", "variants", "p50_latency_ms"):         # never reached with today's legs; the line must parse whatever is added later
        if len(line) <= MAX_LINE:
            break
        out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) > MAX_LINE:
        raise RuntimeError("bench.py: contract line is %d characters (> %d)" % (len(line), MAX_LINE))
    return line


def emit(result, detail_path, out=None):
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(result, f, indent=1)
                f.write("\n")
        except OSError as e:
            print("bench.py: could not write %s: %r" % (detail_path, e), file=sys.stderr)
            detail_path = None
    print(contract_line(result, detail_path), file=out or sys.stdout, flush=True)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (python -m torch.distributed.run), relay
    rank 0's JSON line and the children's exit code.  This parent never touches the GPU - it does not even count devices (every
    rank refuses to run when fewer GPUs than ranks are visible, and that failure is relayed) - and never exec()s."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if r.returncode != 0:
        print("bench.py: a rank failed (torch.distributed.run exit code %d)" % r.returncode, file=sys.stderr)
        return r.returncode
    if len(lines) != 1:
        print("bench.py: expected ONE JSON line from rank 0, got %d" % len(lines), file=sys.stderr)
        return 3
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE under a launcher, else 1)")
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--lost", type=int, default=32, help="lost tracks per step (B)")
    ap.add_argument("--proposals", type=int, default=16, help="proposals per track (P)")
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=11)
    ap.add_argument("--precision", choices=["f32", "x3", "f16"], default=os.environ.get("BUSCA_BENCH_PRECISION", "f32"),
                    help="arithmetic of the primary line (default f32 = exact float32 MFMA, the reference's own arithmetic; x3 = the library's default "
                         "flavour, float32-equivalent GEMMs as three fp16 MFMAs per product block); the others are reported under `variants`")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where rank 0 writes everything it measured ('' = nowhere)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="independent steps handed to one C-ABI call; 0 = automatic: the K timed steps as ceil(K/64) launches of "
                         "near-equal size, so a short run is ONE launch whose workgroups back-fill the CUs round after round")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--latency-samples", type=int, default=1000)
    ap.add_argument("--full-steps", type=int, default=6, help="full steps (ReID + DT) timed for `full_step` (0 = skip)")
    ap.add_argument("--no-variants", action="store_true", help="skip the secondary-precision / full-step / end-to-end measurements")
    ap.add_argument("--e2e-frames", type=int, default=20, help="frames of the simulated-tracker end-to-end leg (0 = skip)")
    ap.add_argument("--split-steps", type=int, default=20, help="steps of the cfg5 split-tracks leg (0 = skip)")
    args = ap.parse_args()
    gpus_given = args.gpus is not None
    if not gpus_given:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))      # before anything touches the GPU

    contract_out, sys.stdout = sys.stdout, sys.stderr        # stdout carries the contract line and nothing else: stray prints of any leg go to stderr
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if gpus_given and world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    dist = None
    # one rank per GPU.  BUSCA_BENCH_BACKEND=gloo is a TEST mode for boxes with fewer GPUs than ranks: ranks wrap around the
    # visible devices and the bookkeeping (barrier, max over ranks, rank reports) runs over gloo instead of RCCL, so the N > 1
    # code path can be exercised on a 1-GPU box; its numbers mean nothing.
    backend = os.environ.get("BUSCA_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1 or (backend == "nccl" and ndev < world):
        print("bench.py: %d rank(s) but %d GPU(s) visible; refusing to fall back to fewer ranks" % (world, ndev), file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from busca_amd import _lib, sharding

    B, P, L, d, ff = args.lost, args.proposals, args.seq_len, args.d, 2 * args.d
    # steps per launch: with F = 8 a 20-step run was three padded launches (7 + 7 + 6 steps = 224 + 224 + 192 workgroups on 256
    # CUs, one round each); as ONE 640-workgroup launch the hardware back-fills the CUs and the tail round is the only partial one
    F = max(1, args.inflight) if args.inflight > 0 else max(1, min(64, args.steps))
    seed = 7   # the reference configs' tracker.seed (config/*/*/*.yml:18)
    sd = synth.dt_state_dict(seed, d=d, ff=ff)
    ctx = _lib.Context(dev_index)
    # synthetic inputs resident in HBM before the timed region: F steps worth of tracks, each rank its own seed
    big = synth.dt_inputs(seed + 1000 * rank, B * F, L, P)
    tens = {k: torch.from_numpy(v).to(dev) for k, v in big.items()}
    run = DTRunner(ctx, sd, args.precision, tens, B, L, P, F, dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- warm-up, then EXACTLY K steps between barriers ---------------------------------------------------------
    run.run_steps(args.warmup)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the roofline's kernel time is measured LIVE, over the timed region itself: HIP events around every kernel launch of the K steps, on the launch stream
    # (busca_timing_*: two hipEventRecord per launch, read back after the closing barrier)
    ctx.lib.busca_timing_read(ctx.h, None, None, 1)
    ctx.lib.busca_timing_enable(ctx.h, 1)
    t0 = time.perf_counter()
    ev0.record()
    n_launch = run.run_steps(args.steps)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    k_avg, k_n = C.c_double(0), C.c_int64(0)
    ctx.lib.busca_timing_read(ctx.h, C.byref(k_avg), C.byref(k_n), 1)
    ctx.lib.busca_timing_enable(ctx.h, 0)
    my_elapsed = elapsed
    ev_ms = ev0.elapsed_time(ev1)
    elapsed = sharding.max_over_ranks(elapsed, dist, red_dev)
    # who was live: every rank reports (rank, device index, device name, library version, its own elapsed time)
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev),
          "busca_version": int(ctx.lib.busca_version()), "build": _lib.build_info(ctx.lib), "elapsed_s": my_elapsed, "steps": args.steps}
    ranks = [me]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    # ---- roofline: the timed launches' own event brackets ---------------------------------------------------------
    geom = run.geometry()
    timing = (k_avg.value * k_n.value, k_n.value, n_launch, args.steps) if k_n.value else (ev_ms, n_launch, n_launch, args.steps)
    p50 = run.p50_latency_ms(args.latency_samples)

    # ---- BASELINE configs[4] with one step's tracks split over the ranks: every rank takes part ------------------
    split = None
    if args.split_steps > 0 and (world > 1 or not args.no_variants or "WORLD_SIZE" in os.environ):
        try:
            split = cfg5_split_leg(ctx, dev, rank, world, dist, red_dev, args.split_steps)
        except Exception as e:
            if world > 1:
                raise
            split = {"error": repr(e)}

    result = None
    if rank == 0:
        total_steps = args.steps * world
        result = {
            "metric": "BUSCA association steps/sec (DT-step: features in HBM -> logits/probs/argmax), MOT17-like %d lost x %d proposals" % (B, P),
            "value": total_steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "dtype_note": {"x3": "float32-equivalent: every GEMM product block is three fp16 MFMAs on operands split into fp16 hi + lo (22 mantissa bits), f32 accumulate; attention, "
                                 "LayerNorm, softmax and the residual stream in f32 - logits 1e-5 from the exact-f32 flavour (`variants.f32`), same parity bars",
                           "f32": "exact float32 MFMA (v_mfma_f32_16x16x4_f32)", "f16": "fp16 operands, f32 accumulate (opt-in)"}[args.precision],
            "p50_latency_ms": p50,
            "config": {"workload": "cfgN DT-step: %d lost x %d proposals x d%d (L=%d, T=%d, ff=%d, 4 layers, 4 heads), ReID features "
                                   "precomputed; BASELINE.json configs[1]-shaped batch without the tracker" % (B, P, d, L, L + 2 * (P + 2), ff),
                       "lost": B, "proposals": P, "d": d, "seq_len": L, "steps_in_flight_per_launch": F, "launches_in_timed_region": n_launch,
                       "parallelism": "independent sequences sharded per GPU, no collective (%d rank%s)%s" % (
                           world, "" if world == 1 else "s", "" if backend == "nccl" else "; TEST MODE backend=%s, ranks share %d GPU(s)" % (backend, ndev))},
            "roofline": roofline_obj(args.precision, B, L, P, d, ff, timing, ev_ms / max(1, n_launch), geometry=geom),
            "ranks": ranks,
        }
        if split is not None:
            result["configs"] = {"cfg5_split": split}
    # ---- the other precisions (rank 0, outside the contract's timed region) ------------------------------------------
    if rank == 0 and not args.no_variants:
        result["variants"] = {}
        r2 = None

        def variant(prec, Fv, k, key, note):
            tv = tens
            if Fv != F:
                tv = {kk: torch.from_numpy(v).to(dev) for kk, v in synth.dt_inputs(seed + 1000 * rank, B * Fv, L, P).items()}
            rv = DTRunner(ctx, sd, prec, tv, B, L, P, Fv, dev)
            rv.run_steps(max(Fv, k // 10))
            torch.cuda.synchronize(dev)
            a = time.perf_counter()
            nl = rv.run_steps(k)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - a
            tm = rv.kernel_time(min(k, 50 * Fv))
            if not tm[1]:
                tm = (el * 1e3, nl, nl, k)
            result["variants"][key] = {"value": k / el, "unit": "steps/s", "steps": k, "n_gpus": 1, "dtype": prec, "steps_in_flight_per_launch": Fv, "note": note,
                                       "roofline": roofline_obj(prec, B, L, P, d, ff, tm, el / nl * 1e3, geometry=rv.geometry())}
            return rv

        for other in [q for q in ("f32", "x3", "f16") if q != args.precision]:
            # (a) the launch shape of the primary line (F steps per launch), launches queued back to back
            rv = variant(other, F, F * max(1, 400 // F), other, "same steps per launch as the primary line, launches queued back to back")
            result["variants"][other]["p50_latency_ms"] = rv.p50_latency_ms(min(args.latency_samples, 300))
            # (b) whole rounds of workgroups: 8 steps = 256 one-track workgroups on 256 CUs (the f16 flavour packs two tracks per workgroup from
            # 257 tracks on: 16 steps)
            F2 = 16 if other == "f16" else 8
            rw = variant(other, F2, 1600, other + "_whole_rounds", "launches of %d steps = whole rounds of workgroups on 256 CUs" % F2)
            if other == "f16":
                r2 = rw
        if F != 8:
            variant(args.precision, 8, 1600, args.precision + "_whole_rounds", "launches of 8 steps = one whole round of workgroups on 256 CUs")
        # the other BASELINE shapes as their own DT-step lines (cfgR = shipped model shape; cfg4 = BASELINE configs[3];
        # cfg5 = BASELINE configs[4], one GPU's share is the full 512-track step here)
        cfgs = result.setdefault("configs", {})
        for name, cB, cP, cd, prec, cF, csteps in (("cfgR", 32, 5, 512, "x3", 8, 600), ("cfgR_f32", 32, 5, 512, "f32", 8, 400), ("cfgR_f16", 32, 5, 512, "f16", 8, 800),
                                                   ("cfg4", 128, 32, 512, "x3", 2, 40), ("cfg4_f32", 128, 32, 512, "f32", 2, 20), ("cfg4_f16", 128, 32, 512, "f16", 2, 60),
                                                   ("cfg5", 512, 64, 512, "f16", 1, 20)):
            try:
                cfgs[name] = config_leg(ctx, dev, name, cB, L, cP, cd, prec, cF, csteps)
            except Exception as e:
                cfgs[name] = {"error": repr(e)}
        try:        # BASELINE configs[3] as a FULL step (crops cut on the GPU, 1 408 + 4 096-crop BatchNorm batches, DT at T = 79)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import cfg4_step
            cfgs["cfg4_full_step"] = cfg4_step.run(3, "x3", reid_precision="x3")               # library defaults (x3 ReID + x3 DT: at T = 79 the split-fp16 layer-wise kernels)
            cfgs["cfg4_full_step_f16"] = cfg4_step.run(3, "f16", reid_precision="f16")          # opt-in fast flavours
            cfgs["cfg4_full_step_f16_expanded_batch"] = cfg4_step.run(3, "f16", dedup=False, reid_precision="f16")
        except Exception as e:
            cfgs["cfg4_full_step"] = {"error": repr(e)}
        try:
            result["hbm_kernels"] = hbm_kernels(ctx, dev)
        except Exception as e:
            result["hbm_kernels"] = {"error": repr(e)}
        if args.full_steps > 0:
            from busca_amd.dt import DecisionTransformerHIP
            # UNPREFIXED `full_step` = the library's default flavour: float32-EQUIVALENT ReID on the fp16 matrix cores (BUSCA_PREC_F16X3: same parity bars as the
            # exact flavour, tests/test_reid_gpu.py EXACT_FLAVOURS) + the float32-equivalent (x3) Decision Transformer.  Its own context: a busca_ctx holds one ReID weight set.
            try:
                ctx3 = _lib.Context(dev_index)
                dt3 = DecisionTransformerHIP(ctx3, sd, activation="relu", fake_bbox_f64=True, precision="x3")
                result["full_step"] = full_step(ctx3, dt3, B, L, P, args.full_steps, dev, reid_precision="x3")
                result["full_step_tracker_like_candidates"] = full_step(ctx3, dt3, B, L, P, args.full_steps, dev, n_det=118, reid_precision="x3")
                ctx3.close()
            except Exception as e:
                result["full_step"] = {"error": repr(e)}
            # opt-in fast flavours: fp16 ReID + f16 Decision Transformer
            try:
                dt16 = r2.model if r2 is not None else DTRunner(ctx, sd, "f16", tens, B, L, P, 1, dev).model
                result["full_step_f16"] = full_step(ctx, dt16, B, L, P, args.full_steps, dev)
                result["full_step_f16_tracker_like_candidates"] = full_step(ctx, dt16, B, L, P, args.full_steps, dev, n_det=118)
            except Exception as e:
                result["full_step_f16"] = {"error": repr(e)}
            # the same step in the REFERENCE's arithmetic: exact-f32 ReID convs (<= 1e-4 against the reference's own features) +
            # the f32 Decision Transformer, priced against the f32 MFMA peak
            try:
                ctx32 = _lib.Context(dev_index)
                dt32 = DecisionTransformerHIP(ctx32, sd, activation="relu", fake_bbox_f64=True, precision="f32")
                n32 = max(2, args.full_steps // 2)
                result["full_step_f32"] = full_step(ctx32, dt32, B, L, P, n32, dev, reid_precision="f32")
                result["full_step_f32_tracker_like_candidates"] = full_step(ctx32, dt32, B, L, P, n32, dev, n_det=118, reid_precision="f32")
                ctx32.close()
            except Exception as e:
                result["full_step_f32"] = {"error": repr(e)}
    if rank == 0:
        if args.cpu_seconds > 0:
            one = {k: v[:B] for k, v in big.items()}
            result["cpu_baseline"] = cpu_baseline(sd, dict(d=d, ff=ff), one, args.cpu_seconds)
        else:
            result["cpu_baseline"] = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if rank == 0 and not args.no_variants and args.e2e_frames > 0:
        try:
            result["assoc_e2e"] = assoc_e2e(args.e2e_frames)
        except Exception as e:      # the end-to-end leg is informative, never fatal for the contract line
            result["assoc_e2e"] = {"error": repr(e)}
    if rank == 0:
        emit(result, args.detail, contract_out)


if __name__ == "__main__":
    main()
