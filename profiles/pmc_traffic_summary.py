#!/usr/bin/env python3
"""Per-kernel HBM traffic of one pass of a program profiled by tools/pmc_traffic.sh.

  pmc_traffic_summary.py <dir> <passes>            (passes = how many times the program ran the measured pass, e.g. 4 for
                                                     `tools/reid_bench.py N 2`: 2 warm-up + 2 timed forwards)

Columns: launches per pass, kernel time per pass (us, from the un-instrumented --kernel-trace run), HBM read MB = FETCH_SIZE
x 2 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads; the counter is in KiB),
HBM write MB = WRITE_SIZE (KiB; uncalibrated per the guide - treat as indicative), and (read + write) / time."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def load(pattern, col=None):
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in sorted(glob.glob(pattern, recursive=True)):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if col is None:
                acc[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            else:
                if row["Counter_Name"] != col:
                    continue
                acc[k] += float(row["Counter_Value"])
            cnt[k] += 1
    return acc, cnt


def main(d, passes):
    passes = float(passes)
    fetch, _ = load(os.path.join(d, "**", "fetch_counter_collection.csv"), "FETCH_SIZE")
    write, _ = load(os.path.join(d, "**", "write_counter_collection.csv"), "WRITE_SIZE")
    tns, tcnt = load(os.path.join(d, "**", "trace_kernel_trace.csv"))
    keys = sorted(tns, key=lambda k: -tns[k])
    print("# %s, per pass (totals / %g passes); read = FETCH_SIZE KiB x 2 (gfx950 correction), write = WRITE_SIZE KiB" % (d, passes))
    print("%-70s %7s %10s %10s %10s %8s" % ("kernel", "calls", "time_us", "read_MB", "write_MB", "TB/s"))
    tot = [0.0, 0.0, 0.0, 0.0]
    for k in keys:
        us = tns[k] / passes / 1e3
        rd = fetch.get(k, 0.0) / passes * 1024 * 2 / 1e6
        wr = write.get(k, 0.0) / passes * 1024 / 1e6
        print("%-70s %7.1f %10.1f %10.1f %10.1f %8.2f" % (k, tcnt[k] / passes, us, rd, wr, (rd + wr) / us if us else 0.0))
        tot[0] += tcnt[k] / passes; tot[1] += us; tot[2] += rd; tot[3] += wr
    print("%-70s %7.1f %10.1f %10.1f %10.1f %8.2f" % ("TOTAL", tot[0], tot[1], tot[2], tot[3], (tot[2] + tot[3]) / tot[1] if tot[1] else 0.0))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 1)
