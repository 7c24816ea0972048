#!/usr/bin/env python3
"""Host profile of a frame's get_image_crops calls inside the simulated tracker (cProfile, top entries): python tools/crop_profile.py [lost] [objects]"""
import cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd.network import BUSCA
from busca_amd.sim import SimScene

lost, n_obj = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 150)
args = types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=512, ff_size=1024, activation="gelu", dropout_p=0.1, input_flavour="MEM-SEP-CAN-BAD",
                             output_flavour="CAN", encode_separator_as_reference=True, encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), seed=7)
model = BUSCA(args).to(torch.device("cuda:0")).eval()
scene = SimScene(model, n_objects=n_obj)
scene.warm_up(12)
frames = [scene.next_frame() for _ in range(40)]
torch.cuda.synchronize()
pr = cProfile.Profile()
ts = []
for frame, boxes in frames:
    pr.enable()
    scene.crop_inputs(frame, boxes, lost)
    pr.disable()
    ts.append(scene.last_crop_calls_s)
print("p50 crop calls %.3f ms" % (np.percentile(ts, 50) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
