#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (``--kernel-trace --stats`` output, *.db) as a per-kernel table:
calls, avg/min/max duration (ns), total (ns), share.  Usage: rocpd_summary.py results.db > summary.txt"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                       "from kernels group by name order by 6 desc").fetchall()
    total = sum(r[5] for r in rows) or 1
    print("# source: %s" % path)
    print("%-90s %8s %12s %12s %12s %14s %6s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ns", "pct"))
    for name, n, avg, mn, mx, tot in rows:
        print("%-90s %8d %12.0f %12d %12d %14d %6.2f" % (name[:90], n, avg, mn, mx, tot, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
