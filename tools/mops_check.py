#!/usr/bin/env python3
"""Executed matrix work of a frame from a rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 pass, against the algorithmic count:
    python tools/mops_check.py <pmc dir> <config C2|C3> <frames in the pass>
One MOPS unit = 512 FLOP (MI355X_MICROARCH.md).  Algorithmic: 35,853,696 FLOP per padded LR pixel; the dominant kernel
(the 276 dense-block convs, conv3x3_flow<1,1,false,1,true,true>) 2*9*32*(64+96+128+160)*69 = 17,805,312 FLOP per padded LR pixel."""
import csv
import glob
import sys
from collections import defaultdict

root, cfg, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
PX = {"C2": 2544000, "C3": 9211200}[cfg]
tot = defaultdict(float)
n = defaultdict(int)
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == "SQ_INSTS_VALU_MFMA_MOPS_F16":
                k = row["Kernel_Name"]
                tot[k] += float(row["Counter_Value"])
                n[k] += 1
flow = {k: v for k, v in tot.items() if "conv3x3_flow" in k}
allf = sum(flow.values()) * 512 / frames
dom = sum(v for k, v in flow.items() if "1, 1, false, 1, true, true" in k.replace("(bool)0", "false").replace("(bool)1", "true")) * 512 / frames
print("%s, %d frames in the pass, %d padded LR px per frame" % (cfg, frames, PX))
for k in sorted(flow):
    print("  %-70s launches/frame %6.1f  executed %.3f TFLOP/frame" % (k[:70], n[k] / frames, flow[k] * 512 / frames / 1e12))
alg_all, alg_dom = PX * 35853696.0, PX * 17805312.0
print("all convs:        executed %.2f TFLOP per frame vs algorithmic %.2f = %+.2f %%" % (allf / 1e12, alg_all / 1e12, 100 * (allf / alg_all - 1)))
if dom:
    print("dominant kernel:  executed %.2f TFLOP per frame vs algorithmic %.2f = %+.2f %%  (LR level: no dead-output elimination, the x quantisation shows)" % (
        dom / 1e12, alg_dom / 1e12, 100 * (dom / alg_dom - 1)))
