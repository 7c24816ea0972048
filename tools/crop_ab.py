#!/usr/bin/env python3
"""Crop kernel A/B (option crop_band 1 = LDS-staged band kernel, 0 = one thread per output pixel) + the host time of a frame's two get_image_crops
calls: python tools/crop_ab.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from busca_amd import _lib, geometry, synth

ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
frame = torch.from_numpy(synth.randint_u8(5, "frame", (1080, 1920, 3))).to(dev)
hh = rng.uniform(80, 320, 150); ww = hh * rng.uniform(0.3, 0.5, 150)
x = rng.uniform(0, 1920 - 170, 150); y = rng.uniform(0, 1080 - 330, 150)
tlbr = np.stack([x, y, x + ww, y + hh], 1)
lib, h = ctx.lib, ctx.h
for band in (1, 0, 1, 0):
    ctx.set_option("crop_band", band)
    geometry.crop_gather(ctx, frame, tlbr, want_u8=True); torch.cuda.synchronize()
    lib.busca_timing_read(h, None, None, 1); lib.busca_timing_enable(h, 1)
    for _ in range(50):
        geometry.crop_gather(ctx, frame, tlbr, want_u8=True)
    torch.cuda.synchronize()
    avg, n = C.c_double(0), C.c_int64(0)
    lib.busca_timing_read(h, C.byref(avg), C.byref(n), 1); lib.busca_timing_enable(h, 0)
    print("crop_band=%d: 150 boxes of a 1080p frame: %.1f us per call (fill + resize kernels)" % (band, avg.value * 1e3))
ctx.set_option("crop_band", 1)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import e2e_sim
for lost, objs in ((32, 150), (8, 60)):
    r = e2e_sim.run(lost, objs, 5, 512, frames=30, verbose=False)
    print("e2e lost %d dets %d: p50_crop_ms %.3f (p90 %.3f), p50_assoc %.2f ms" % (lost, objs - lost, r["p50_crop_ms"], r["p90_crop_ms"], r["p50_assoc_latency_ms"]))
