#!/usr/bin/env python3
"""Per-shape conv throughput from a rocprofv3 rocpd db of tools/reid_bench.py: conv_table.py db n_crops [--direct]
(--direct: the trace was taken with BUSCA_REID_GRAM=0; default assumes the automatic schedule)"""
import sqlite3, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from busca_amd import synth
db, n = sys.argv[1], int(sys.argv[2])
direct = '--direct' in sys.argv
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels where name like '%conv_gemm%' or name like '%conv3x3_halo%' or name like '%stem_halo%' order by start").fetchall()
specs = synth.reid_conv_specs()
def osz(h, w, k, s, p): return ((h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1)
out = []
name, co, ci, k, s, p, bn = specs[0]
out.append((ci, co, k, *osz(384, 128, k, s, p), "stem"))
h, w, i = 96, 32, 1
# launch order per bottleneck (capi_reid.hip.inc): conv1, conv2, [downsample], conv3 statistics pass, conv3 merge pass
for li, nb in enumerate((3, 4, 6, 3)):
    for b in range(nb):
        n1 = specs[i]; o1 = osz(h, w, n1[3], n1[4], n1[5]); out.append((n1[2], n1[1], n1[3], *o1, "conv1"))
        n2 = specs[i + 1]; o2 = osz(*o1, n2[3], n2[4], n2[5]); out.append((n2[2], n2[1], n2[3], *o2, "conv2"))
        n3 = specs[i + 2]; o3 = osz(*o2, n3[3], n3[4], n3[5])
        i += 3
        gram = (not direct) and li <= 2 and n * o3[0] * o3[1] >= 65536      # capi_reid.hip.inc: reid_use_gram
        if b == 0:
            nd = specs[i]; od = osz(h, w, nd[3], nd[4], nd[5]); i += 1
            if not gram: out.append((nd[2], nd[1], nd[3], *od, "down"))
        if gram:
            out.append((n3[2] + (nd[2] if b == 0 else 0), n3[1], n3[3], *o3, "tail+down" if b == 0 else "tail"))
        else:
            out.append((n3[2], n3[1], n3[3], *o3, "conv3-stats"))
            out.append((n3[2], n3[1], n3[3], *o3, "tail"))
        h, w = o3
last = rows[-len(out):]
agg = {}
for (ci, co, k, oh, ow, role), (kn, st, en) in zip(out, last):
    fl = 2 * n * oh * ow * co * ci * k * k
    byt = n * (oh * ow * co * 2 + (oh * ow if k == 1 else oh * ow) * ci * 2 * (1 if k == 1 else 1))   # out + in (once), fp16
    a = agg.setdefault((role, ci, co, k, oh), [0, 0, 0, 0]); a[0] += fl; a[1] += (en - st) / 1e3; a[2] += 1; a[3] += byt
print("total conv us %.0f" % sum(v[1] for v in agg.values()))
for key, (fl, us, c, byt) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-11s cin=%4d cout=%4d k=%d oh=%3d x%d %8.1f GFLOP %7.1f us %6.1f TFLOP/s  min-HBM %5.2f TB/s-equivalent" % (*key, c, fl / 1e9, us, fl / us / 1e6, byt / us / 1e6))
