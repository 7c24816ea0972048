#!/usr/bin/env python3
"""Drive busca_bn_stats_1x1 on the ReID layer shapes (run under rocprofv3 --kernel-trace, then tools/gram_bench.py --report db)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [  # n, H, W, Cin, Cout, stride, transform
    (512, 96, 32, 64, 256, 1, True), (512, 96, 32, 64, 256, 1, False),
    (512, 48, 16, 128, 512, 1, True), (512, 96, 32, 256, 512, 2, False),
    (512, 24, 8, 256, 1024, 1, True), (512, 48, 16, 512, 1024, 2, False),
    (88, 96, 32, 64, 256, 1, True), (88, 48, 16, 128, 512, 1, True), (88, 24, 8, 256, 1024, 1, True),
]
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import sqlite3
    con = sqlite3.connect(sys.argv[2])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    rows = [r for r in rows if "gram" in r[0] or "quadform" in r[0]]
    NK = 4
    per = len(rows) // len(SHAPES) // NK
    k = 0
    for sh in SHAPES:
        for it in range(per):
            trip = rows[k:k + NK]; k += NK
            if it == per - 1:
                print("%-40s " % (sh,) + "  ".join("%s %.1f us" % (r[0].split("(")[0].replace("void ", "")[:20], (r[2] - r[1]) / 1e3) for r in trip))
    sys.exit(0)
import numpy as np, torch
from busca_amd import _lib
ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
for (n, H, W, Cin, Cout, stride, tr) in SHAPES:
    x = (torch.randn(n, H, W, Cin, device=dev) * 1.5).half()
    w = (torch.randn(Cout, Cin, device=dev) / Cin ** 0.5).half()
    g = torch.ones(Cout, device=dev); b = torch.zeros(Cout, device=dev)
    ss = torch.stack([torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)], 1).contiguous()
    out = torch.zeros(Cout, 2, device=dev)
    for _ in range(3):
        ctx.check(ctx.lib.busca_bn_stats_1x1(ctx.h, x.data_ptr(), ss.data_ptr() if tr else None, n, H, W, Cin, stride, w.data_ptr(), Cout,
                                             g.data_ptr(), b.data_ptr(), out.data_ptr(), None))
    torch.cuda.synchronize()
print("done")
