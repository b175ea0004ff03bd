#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output (counter_collection.csv) per kernel name: mean counter value per dispatch.

    python tools/pmc_summary.py [--traffic] gpurun_out/<tag>/pmcN [more dirs]

--traffic: the directories hold a FETCH_SIZE and a WRITE_SIZE pass; additionally print, per kernel, HBM bytes per launch
= 2 x FETCH_SIZE (gfx950 correction for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both
reported by rocprofv3 in KiB, and a line `dominant <kernel> bytes_per_launch=<B>` for the kernel with the largest total
(bench.py parses that line from the tracked copy under profiles/)."""
import csv
import glob
import sys
from collections import defaultdict

args = sys.argv[1:]
traffic = "--traffic" in args
args = [a for a in args if a != "--traffic"]
files = []
for root in args:
    files += sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True))
per_kernel = defaultdict(dict)
for f in files:
    acc = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = (row["Kernel_Name"][:70], row["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
    print("#", f)
    for (kn, cn), (n, tot) in sorted(acc.items()):
        print("%-72s %-30s n=%-6d mean=%.5g  total=%.6g" % (kn, cn, n, tot / n, tot))
        per_kernel[kn][cn] = (n, tot / n, tot)
if traffic:
    print("# HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B)")
    best = None
    for kn, d in sorted(per_kernel.items()):
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            rd, wr = 2 * d["FETCH_SIZE"][1] * 1024, d["WRITE_SIZE"][1] * 1024
            n = d["FETCH_SIZE"][0]
            print("traffic %-72s launches=%-5d read=%.4g B write=%.4g B per launch; per pass total %.4g GB" % (kn, n, rd, wr, (rd + wr) * n / 1e9))
            if best is None or (rd + wr) * n > best[1]:
                best = (kn, (rd + wr) * n, rd + wr)
    tot = sum((2 * d["FETCH_SIZE"][2] + d["WRITE_SIZE"][2]) * 1024 for d in per_kernel.values() if "FETCH_SIZE" in d and "WRITE_SIZE" in d)
    print("total HBM bytes of the pass (all kernels): %.4g GB" % (tot / 1e9))
    if best:
        print("dominant %s bytes_per_launch=%.6g" % (best[0].replace(" ", ""), best[2]))
