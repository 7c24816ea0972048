#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per dispatch of one kernel, over all *_counter_collection.csv
files of a directory.  Usage: pmc_summary.py <dir> [kernel-substring] [grid_size_filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, sub="dt_fused", grid=None):
    acc = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            if sub not in row["Kernel_Name"]:
                continue
            if grid and row.get("Grid_Size") != str(grid):
                continue
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("# dir=%s kernel~%s grid=%s" % (d, sub, grid))
    for k in sorted(acc):
        v = acc[k]
        print("%-28s n=%4d avg=%16.1f" % (k, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(*sys.argv[1:])
