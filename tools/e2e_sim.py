#!/usr/bin/env python3
"""End-to-end BUSCA step inside a simulated tracker: crops cut on the GPU, device-resident track memory, centre
distances, associate_embeddings.  python tools/e2e_sim.py [lost] [objects] [proposals] [d] [precision] [frames] [device_only]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from busca_amd.network import BUSCA
from busca_amd.sim import SimScene
from busca_amd.tracking import center_distance


def run(lost=32, n_obj=150, P=5, d=512, precision="x3", frames=30, verbose=True, device_only_crops=False, reid_precision="x3", per_detection_crops=False):
    """Defaults = the library's defaults (busca_amd.network.BUSCA: float32-equivalent x3 Decision Transformer + x3 ReID)."""
    args = types.SimpleNamespace(reid_precision=reid_precision, num_layer=4, nhead=4, dim_embedding=512, trans_dim=d, ff_size=2 * d, activation="gelu", dropout_p=0.1,
                                 input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                                 encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), precision=precision, seed=7,
                                 device_only_crops=device_only_crops)
    model = BUSCA(args).to(torch.device("cuda:0")).eval()
    scene = SimScene(model, n_objects=n_obj)
    scene.warm_up(12)
    t_crop, t_dist, t_assoc, t_sim = [], [], [], []
    for f in range(frames + 3):
        frame, boxes = scene.next_frame()            # producing the synthetic frame is not part of the crop path
        torch.cuda.synchronize()
        a = time.perf_counter()
        lost_t, dets, kal = scene.crop_inputs(frame, boxes, lost, per_detection=per_detection_crops)
        b = time.perf_counter()
        dists = center_distance(lost_t, dets)
        c = time.perf_counter()
        probs, rel = model.associate_embeddings(lost_t, dets, dists, 11, P, True, True, extra_kalman_candidates=kal, normalize_ims=True)
        torch.cuda.synchronize()
        e = time.perf_counter()
        if f >= 3:
            t_crop.append(scene.last_crop_calls_s); t_sim.append(b - a); t_dist.append(c - b); t_assoc.append(e - c)
    assert probs.shape == (lost, (n_obj - lost) + lost) and rel.all()
    res = dict(lost=lost, dets=n_obj - lost, proposals=P, d=d, precision=precision, reid_precision=reid_precision, frames=frames,
               p50_assoc_latency_ms=float(np.percentile(t_assoc, 50) * 1e3), p50_crop_ms=float(np.percentile(t_crop, 50) * 1e3),
               p50_center_distance_ms=float(np.percentile(t_dist, 50) * 1e3), p90_assoc_latency_ms=float(np.percentile(t_assoc, 90) * 1e3),
               max_assoc_latency_ms=float(np.max(t_assoc) * 1e3), p90_crop_ms=float(np.percentile(t_crop, 90) * 1e3),
               p50_crop_and_sim_objects_ms=float(np.percentile(t_sim, 50) * 1e3),
               busca_frames_per_s=float(1.0 / np.mean(np.array(t_assoc) + np.array(t_dist))),
               device_resident_crops=model.last_gather[1] == 0, device_only_crops=device_only_crops, crop_calls_per_frame=scene.last_crop_call_count)
    if verbose:
        print(res)
    return res


def run_multi(n_seq=4, lost=8, n_obj=60, P=5, d=512, precision="x3", frames=20, reid_precision="x3"):
    """S tracker instances (sequences sharded onto this GPU) stepping in the same frame interval: S separate
    associate_embeddings calls versus the same S steps through busca_amd.batcher.StepBatcher (one Decision-Transformer launch
    per interval; every step keeps its own two ReID BatchNorm batches).  Returns per-interval p50 times and checks equality."""
    from busca_amd.batcher import StepBatcher
    args = types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=d, ff_size=2 * d, activation="gelu", dropout_p=0.1,
                                 input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                                 encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), precision=precision, seed=7,
                                 reid_precision=reid_precision)
    model = BUSCA(args).to(torch.device("cuda:0")).eval()
    scenes = [SimScene(model, n_objects=n_obj, seed=7 + 13 * i) for i in range(n_seq)]
    for sc in scenes:
        sc.warm_up(12)
    batcher = StepBatcher(model)
    t_seq, t_bat, same = [], [], True
    for f in range(frames + 3):
        steps = []
        for sc in scenes:
            lost_t, dets, kal = sc.step_inputs(lost)
            steps.append((lost_t, dets, center_distance(lost_t, dets), kal))
        torch.cuda.synchronize()
        a = time.perf_counter()
        single = [model.associate_embeddings(l, dd, ds, 11, P, True, True, extra_kalman_candidates=k, normalize_ims=True) for l, dd, ds, k in steps]
        torch.cuda.synchronize()
        b = time.perf_counter()
        tickets = [batcher.submit(l, dd, ds, 11, P, True, True, extra_kalman_candidates=k, normalize_ims=True) for l, dd, ds, k in steps]
        batcher.flush()
        batched = [t.result() for t in tickets]
        torch.cuda.synchronize()
        c = time.perf_counter()
        same = same and all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(single, batched))
        if f >= 3:
            t_seq.append(b - a); t_bat.append(c - b)
    return dict(sequences=n_seq, lost=lost, dets=n_obj - lost, proposals=P, d=d, precision=precision,
                p50_interval_ms_separate_calls=float(np.percentile(t_seq, 50) * 1e3), p50_interval_ms_step_batcher=float(np.percentile(t_bat, 50) * 1e3),
                steps_per_s_separate_calls=float(n_seq / np.mean(t_seq)), steps_per_s_step_batcher=float(n_seq / np.mean(t_bat)),
                dt_launches_per_interval=1, bit_identical=bool(same))


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]) if a else 32, int(a[1]) if len(a) > 1 else 150, int(a[2]) if len(a) > 2 else 5, int(a[3]) if len(a) > 3 else 512,
        a[4] if len(a) > 4 else "f16", int(a[5]) if len(a) > 5 else 30, device_only_crops=len(a) > 6 and a[6] == "device_only")
