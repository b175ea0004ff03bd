#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output (counter_collection.csv) per kernel name: mean counter value per dispatch.
   python tools/pmc_summary.py gpurun_out/<tag> [more dirs]"""
import csv
import glob
import sys
from collections import defaultdict

files = []
for root in sys.argv[1:]:
    files += sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True))
for f in files:
    acc = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = (row["Kernel_Name"][:60], row["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
    print("#", f)
    for (kn, cn), (n, tot) in sorted(acc.items()):
        print("%-62s %-32s n=%-6d mean=%.4g  total=%.6g" % (kn, cn, n, tot / n, tot))
