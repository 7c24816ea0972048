#!/usr/bin/env python3
"""Where the host time of one frame's crop calls goes (the p50_crop_ms of tools/e2e_sim.py): python tools/crop_profile.py [lost] [objects]"""
import cProfile, os, pstats, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from busca_amd.network import BUSCA
from busca_amd.sim import SimScene

lost = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 150
args = types.SimpleNamespace(reid_precision="f16", num_layer=4, nhead=4, dim_embedding=512, trans_dim=512, ff_size=1024, activation="gelu", dropout_p=0.1,
                             input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                             encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), precision="f16", seed=7)
model = BUSCA(args).to(torch.device("cuda:0")).eval()
scene = SimScene(model, n_objects=n_obj)
scene.warm_up(12)
def med(f, n=30):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); a = time.perf_counter(); f(); ts.append(time.perf_counter() - a)
    return 1e3 * float(np.median(ts))
print("next_frame (np.roll of the synthetic frame): %.3f ms" % med(lambda: scene.next_frame()))
frame, boxes = scene.next_frame()
tlbr = boxes.copy(); tlbr[:, 2:] += tlbr[:, :2]
def upload():
    t = torch.from_numpy(np.ascontiguousarray(frame)).to("cuda:0"); torch.cuda.current_stream().synchronize()
print("frame upload (pageable 1080p -> HBM, waited): %.3f ms" % med(upload))
for sync, name in ((lambda: torch.cuda.current_stream().synchronize(), "current stream"), (torch.cuda.synchronize, "device")):
    def both():
        fr = frame.copy()          # a new frame object: the upload is part of the call
        a = time.perf_counter()
        model.get_image_crops(fr, tlbr[lost:], normalize=False); model.get_image_crops(fr, tlbr[:lost], normalize=False); sync()
        return time.perf_counter() - a
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); ts.append(both())
    print("2 x get_image_crops (%d + %d boxes, new frame), %s waited: %.3f ms" % (n_obj - lost, lost, name, 1e3 * float(np.median(ts))))
def nosync():
    fr = frame.copy(); a = time.perf_counter()
    model.get_image_crops(fr, tlbr[lost:], normalize=False); model.get_image_crops(fr, tlbr[:lost], normalize=False)
    return time.perf_counter() - a
ts = []
for _ in range(30):
    torch.cuda.synchronize(); ts.append(nosync())
print("2 x get_image_crops, host time only (nothing waited): %.3f ms" % (1e3 * float(np.median(ts))))
pr = cProfile.Profile()
frames = [frame.copy() for _ in range(20)]
torch.cuda.synchronize()
pr.enable()
for fr in frames:
    model.get_image_crops(fr, tlbr[lost:], normalize=False); model.get_image_crops(fr, tlbr[:lost], normalize=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

# the same two calls inside the simulated tracker (track memories hold their crops: no slot is released, the pool grows, old host copies retire)
print("---- inside SimScene.crop_inputs (tracks keep their crops) ----")
ts = []
pr2 = cProfile.Profile()
for f in range(40):
    fr, bx = scene.next_frame()
    torch.cuda.synchronize()
    if f >= 10:
        pr2.enable()
    lost_t, dets, kal = scene.crop_inputs(fr, bx, lost)
    if f >= 10:
        pr2.disable()
        ts.append(scene.last_crop_calls_s)
    for t, d in zip(scene.tracks[lost:], dets):          # detected tracks take the new crop, as a tracker's update does
        t.update(d.tlwh, d.images_mem[0])
print("crop calls inside the scene: p50 %.3f ms, p90 %.3f ms" % (1e3 * float(np.median(ts)), 1e3 * float(np.percentile(ts, 90))))
pstats.Stats(pr2).sort_stats("tottime").print_stats(14)
