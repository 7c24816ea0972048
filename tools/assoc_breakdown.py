#!/usr/bin/env python3
"""Where does associate_embeddings spend its time?  python tools/assoc_breakdown.py [lost] [objs]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd.network import BUSCA
from busca_amd.sim import SimScene
from busca_amd.tracking import center_distance
lost = int(sys.argv[1]) if len(sys.argv) > 1 else 8
objs = int(sys.argv[2]) if len(sys.argv) > 2 else 60
args = types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=512, ff_size=1024, activation="gelu", dropout_p=0.1,
                             input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                             encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), precision=os.environ.get("AB_PREC", "x3"), reid_precision=os.environ.get("AB_REID", "x3"), seed=7)
m = BUSCA(args).to(torch.device("cuda:0")).eval()
sc = SimScene(m, n_objects=objs)
sc.warm_up(12)
import cProfile, pstats
lt, dets, kal = sc.step_inputs(lost)
d = center_distance(lt, dets)
for _ in range(3):
    m.associate_embeddings(lt, dets, d, 11, 5, True, True, extra_kalman_candidates=kal, normalize_ims=True)
torch.cuda.synchronize()
# GPU-only pieces
mem = torch.zeros(lost * 11, 384, 128, 3, dtype=torch.uint8, device="cuda"); can = torch.zeros(lost * 5, 384, 128, 3, dtype=torch.uint8, device="cuda")
def t(fn, n=20):
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e3
print("reid pair   %.3f ms" % t(lambda: m._reid_pair(mem, can)))
print("reid mem    %.3f ms" % t(lambda: m._reid.forward(mem)))
print("reid can    %.3f ms" % t(lambda: m._reid.forward(can)))
mf = torch.zeros(lost, 11, 512, device="cuda"); cf = torch.zeros(lost, 5, 512, device="cuda")
mb = torch.rand(lost, 11, 4, device="cuda") * 100; mb[..., 2:] += mb[..., :2] + 10; cb = torch.rand(lost, 5, 4, device="cuda") * 100; cb[..., 2:] += cb[..., :2] + 10
print("dt forward  %.3f ms" % t(lambda: m._dt.forward(mf, cf, mb, cb)))
print("assoc total %.3f ms" % t(lambda: m.associate_embeddings(lt, dets, d, 11, 5, True, True, extra_kalman_candidates=kal, normalize_ims=True)))
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    m.associate_embeddings(lt, dets, d, 11, 5, True, True, extra_kalman_candidates=kal, normalize_ims=True)
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
