"""Wall time of the host-visible sections of associate_embeddings (synchronised after each): python tools/assoc_sections.py [lost]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from busca_amd.network import BUSCA
from busca_amd.sim import SimScene
from busca_amd.tracking import center_distance
from busca_amd import geometry
lost = int(sys.argv[1]) if len(sys.argv) > 1 else 32
args = types.SimpleNamespace(num_layer=4, nhead=4, dim_embedding=512, trans_dim=512, ff_size=1024, activation="gelu", dropout_p=0.1,
                             input_flavour="MEM-SEP-CAN-BAD", output_flavour="CAN", encode_separator_as_reference=True,
                             encode_special_tokens=False, reid_weights_file="no", device=torch.device("cuda:0"), precision="f16", seed=7)
m = BUSCA(args).to(torch.device("cuda:0")).eval()
sc = SimScene(m, n_objects=60)
sc.warm_up(12)
lt, dets, kal = sc.step_inputs(lost)
d = center_distance(lt, dets)
for _ in range(3):
    m.associate_embeddings(lt, dets, d, 11, 5, True, True, extra_kalman_candidates=kal, normalize_ims=True)
sync = torch.cuda.synchronize
def t(fn, n=30):
    sync(); a = time.perf_counter()
    for _ in range(n): r = fn()
    sync(); return (time.perf_counter() - a) / n * 1e3, r
refs_mem = [[trk.images_mem[i] for i in range(-11, 0)] for trk in lt]
refs_can = [[dets[i % len(dets)].images_mem[-1] for i in range(5)] for _ in lt]
as_u8 = lambda x: np.asarray(x)
tm, gm = t(lambda: m._gather_crops(refs_mem, as_u8))
tc, gc = t(lambda: m._gather_crops(refs_can, as_u8))
print("gather mem (%d crops) %.3f ms, can (%d) %.3f ms" % (gm.shape[0], tm, gc.shape[0], tc))
tp, _ = t(lambda: m._reid_pair(gm, gc))
print("reid pair %.3f ms" % tp)
dd = np.ascontiguousarray(np.asarray(d, dtype=np.float64))
tk, _ = t(lambda: geometry.topk_rows(m._ctx, dd, 5).cpu().numpy())
print("topk_rows + D2H %.3f ms" % tk)
ta, _ = t(lambda: m.associate_embeddings(lt, dets, d, 11, 5, True, True, extra_kalman_candidates=kal, normalize_ims=True))
print("assoc total %.3f ms" % ta)
