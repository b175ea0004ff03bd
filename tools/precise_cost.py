#!/usr/bin/env python3
"""Where does precise mode (rsr_set_option precise 1) spend its extra time?  Per-class HIP-event times of the convolutions of a C2
frame, fp16 storage vs precise, alternating in one process.
    python tools/precise_cost.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
sr = R.RealSR(0); sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")); sr.tilesize = 200
w, h = 1920, 1080
d_in = torch.from_numpy(synth.make_image(1235, w, h)).cuda()
d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cls = {}
for i in range(351):
    if i == 0: c = "conv_first"
    elif i <= 345:
        j, k = divmod(i - 1, 5)
        c = ("conv%d" % (k + 1)) if k < 4 else ("conv5 +rrdb" if j % 3 == 2 else "conv5")
    else: c = ["trunk_conv", "upconv1", "upconv2", "HRconv", "conv_last"][i - 346]
    cls.setdefault(c, []).append(i)
acc = {0: np.zeros(351), 1: np.zeros(351)}
for p in (0, 1):
    sr.set_option("precise", p); sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
for r in range(rounds):
    for p in (0, 1):
        sr.set_option("precise", p)
        sr.set_profiling(True)
        sr.get_conv_times(reset=True)
        for _ in range(3):
            sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        acc[p] += np.array(sr.get_conv_times(reset=True)) / 3
        sr.set_profiling(False)
print("class            launches   fp16 ms   precise ms   delta ms   per launch us (fp16 -> precise)")
tot = [0, 0]
for c, idx in cls.items():
    a, b = acc[0][idx].sum() / rounds, acc[1][idx].sum() / rounds
    tot[0] += a; tot[1] += b
    print("%-16s %5d   %8.3f   %8.3f   %+8.3f   %7.1f -> %7.1f" % (c, len(idx), a, b, b - a, a / len(idx) * 1e3, b / len(idx) * 1e3))
print("%-16s %5d   %8.3f   %8.3f   %+8.3f" % ("all convs", 351, tot[0], tot[1], tot[1] - tot[0]))
sr.close()
