#!/usr/bin/env python3
"""Per-launch timeline of the LAST pass in a rocprofv3 --kernel-trace CSV: timeline.py <kernel_trace.csv> [first-kernel-substring]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = sys.argv[2] if len(sys.argv) > 2 else 'stem_'
idx = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]
seq = rows[idx[-1]:]
t0 = int(seq[0]['Start_Timestamp'])
agg = {}
for r in seq:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = re.sub(r'\(.*$', '', re.sub(r'^void ', '', r['Kernel_Name']))[:60]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
    if '-v' in sys.argv:
        print("%8.1f %7.1f  %-60s grid=%s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d, n, r.get('Grid_Size_X', r.get('Grid_Size'))))
wall = (int(seq[-1]['End_Timestamp']) - t0) / 1e3
print("# %d launches, kernel sum %.1f us, wall %.1f us" % (len(seq), sum(v[1] for v in agg.values()), wall))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %4d %9.1f us %6.1f us/launch" % (n, c, d, d / c))
