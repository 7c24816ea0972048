#!/usr/bin/env python3
"""Per-kernel table (calls, total, average, share, min, max) of a `rocprofv3 --kernel-trace --output-format csv` output directory.
Usage: kstats.py <dir> [skip_first_n_calls_per_kernel]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    a = agg.setdefault(r["Kernel_Name"][:110], [0, 0, 10**18, 0])
    t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a[0] += 1; a[1] += t; a[2] = min(a[2], t); a[3] = max(a[3], t)
tot = sum(v[1] for v in agg.values()) or 1
print("%-110s %7s %13s %11s %6s %10s %10s" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-110s %7d %13d %11d %6.2f %10d %10d" % (k, v[0], v[1], v[1] // v[0], 100.0 * v[1] / tot, v[2], v[3]))
print("%-110s %7d %13d" % ("TOTAL", sum(v[0] for v in agg.values()), tot))
