#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) into a per-kernel stats table
(the equivalent of `--stats` kernel_stats.csv).   python tools/rocpd_kernel_stats.py <results.db> [more.db]"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else "kernel_name"
        rows = cur.execute(
            "select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
            "group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
        total = sum(r[2] for r in rows) or 1
        print("# %s" % path)
        print("%-86s %8s %14s %12s %12s %12s %7s" % ("Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct"))
        for n, c, tot, avg, mn, mx in rows:
            print("%-86s %8d %14d %12.0f %12d %12d %6.2f%%" % (n[:86], c, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main()
