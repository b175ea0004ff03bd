#!/usr/bin/env python3
"""Frame time, socket power and sclk for ablation variants of the C2 frame (GPU box):  python tools/power_probe.py "dbg=0;dbg=1;..."
The ablation bits 1 / 4 / 64 / 128 exist only in an experiment build of conv_flow.hip (round 5: compiled out of the product):
    tools/build_variant.sh abl conv_flow -DRSR_EXPERIMENT;  RSR_LIB=realsr-ncnn-vulkan_amd/lib/exp/abl.so python tools/power_probe.py ..."""
import os, subprocess, sys, time, re, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

d = synth.make_model_dir("/tmp/rsr_models", "models-DF2K", 42)
sr = R.RealSR(0); sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")); sr.tilesize = 200
w, h = 1920, 1080
d_in = torch.from_numpy(synth.make_image(3, w, h)).cuda()
d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            m1 = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o); m2 = re.search(r"Power \(W\): ([0-9.]+)", o)
            if m1 and m2:
                samples.append((time.time(), int(m1.group(1)), float(m2.group(1))))
        except Exception:
            pass


th = threading.Thread(target=sampler); th.start()
for var in (sys.argv[1] if len(sys.argv) > 1 else "dbg=0;dbg=1;dbg=4;dbg=5;dbg=2;dbg=0").split(";"):
    for kv in var.split(","):
        k, v = kv.split("="); sr.set_option(k, int(v))
    sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr()); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr()); n += 1
    torch.cuda.synchronize(); t1 = time.time()
    ss = [s for s in samples if t0 + 1.0 < s[0] < t1]
    print("%-14s %7.1f ms/frame  sclk %4.0f MHz  power %5.0f W  (%d samples)" % (var, (t1 - t0) / n * 1e3,
          sum(s[1] for s in ss) / max(len(ss), 1), sum(s[2] for s in ss) / max(len(ss), 1), len(ss)), flush=True)
stop = True; th.join(); sr.close()
