#!/usr/bin/env python3
"""Per-kernel SQ counter table from the rocprofv3 --pmc passes of tools/pmc_reid.sh / tools/pmc_dt.sh.

  pmc_kernel_table.py <dir> [min_dispatches]

For every kernel name: dispatches seen, and per-dispatch averages of the counters of all *_counter_collection.csv files in <dir>,
plus derived ratios (MI355X_MICROARCH.md, rocprofv3 PMC slots: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs, SQ_BUSY_CYCLES counts per shader engine):
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES/32 x 1024 SIMDs)   (matrix pipes busy, share of the kernel's duration on the whole chip)
  wait_any    = SQ_WAIT_ANY / SQ_WAVE_CYCLES          (wave parked on s_waitcnt / barrier)
  wait_inst   = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     (issue stall: MFMA dependency / pipe busy)
  valu/mfma   = SQ_INSTS_VALU / SQ_INSTS_MFMA
  lds_conf    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:58]


def main(d, min_n=1):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    cols = ["mfma_busy", "wait_any", "wait_inst", "act_valu", "act_lds", "act_vmem", "valu/mfma", "lds_conf"]
    print("# %s" % d)
    print("%-58s %6s %14s " % ("kernel", "disp", "wave_qcycles") + " ".join("%9s" % c for c in cols))
    rows = []
    for k, c in acc.items():
        a = {n: sum(v) / len(v) for n, v in c.items()}
        n = max(len(v) for v in c.values())
        if n < int(min_n):
            continue
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        busy = a.get("SQ_BUSY_CYCLES", 0.0)

        def r(x, y):
            return a.get(x, 0.0) / y if y else float("nan")
        vals = [r("SQ_VALU_MFMA_BUSY_CYCLES", busy / 32.0 * 1024.0), r("SQ_WAIT_ANY", wc), r("SQ_WAIT_INST_ANY", wc), r("SQ_ACTIVE_INST_VALU", wc),
                r("SQ_ACTIVE_INST_LDS", wc), r("SQ_ACTIVE_INST_VMEM", wc), r("SQ_INSTS_VALU", a.get("SQ_INSTS_MFMA", 0.0)),
                r("SQ_LDS_BANK_CONFLICT", a.get("SQ_LDS_IDX_ACTIVE", 0.0))]
        rows.append((wc * n, k, n, wc, vals))
    for _, k, n, wc, vals in sorted(rows, reverse=True):
        print("%-58s %6d %14.0f " % (k, n, wc) + " ".join("%9.3f" % v for v in vals))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 1)
