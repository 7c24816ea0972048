import os, sys, time, types, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import e2e_sim
for dev_only in (False, True):
    r = e2e_sim.run(8, 60, 5, 512, "f16", 20, verbose=False, device_only_crops=dev_only)
    print(dev_only, {k: round(v, 3) if isinstance(v, float) else v for k, v in r.items()})
from busca_amd.network import BUSCA
from busca_amd.sim import SimScene
from busca_amd.tracking import center_distance
args = types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=512, ff_size=1024, activation="gelu", dropout_p=0.1,
                             input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                             encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), precision="f16", seed=7, device_only_crops=True)
m = BUSCA(args).to(torch.device("cuda:0")).eval()
sc = SimScene(m, n_objects=60); sc.warm_up(12)
lt, dets, kal = sc.step_inputs(8); d = center_distance(lt, dets)
for _ in range(3): m.associate_embeddings(lt, dets, d, 11, 5, True, True, extra_kalman_candidates=kal, normalize_ims=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): m.associate_embeddings(lt, dets, d, 11, 5, True, True, extra_kalman_candidates=kal, normalize_ims=True)
torch.cuda.synchronize(); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(10)
