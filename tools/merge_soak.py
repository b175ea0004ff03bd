#!/usr/bin/env python3
"""Soak of the cross-call merging (engine.h): 24 caller threads hammer ONE context for N seconds with 25 images of random small sizes (RGB and
RGBA, one of them too large to merge), host and device API mixed, while thread 0 flips the options "merge" / "merge_mixed" -- every result must
equal the lone call's.     python tools/merge_soak.py [seconds]   (MI355X box)"""
import os, sys, time, threading, random
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth
d = synth.make_model_dir("/tmp/rsr_models", "models-DF2K", 42)
s = R.RealSR(0); s.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")); s.tilesize = 48
rng = np.random.default_rng(11)
sizes = [(int(rng.integers(8, 160)), int(rng.integers(8, 160)), int(rng.choice([3, 3, 4]))) for _ in range(24)] + [(400, 300, 3)]
imgs = [synth.make_image(100 + i, *sz) for i, sz in enumerate(sizes)]
s.set_option("merge", 1)
lone = [s.process(im) for im in imgs]
s.set_option("merge", 16); s.set_option("max_lanes", 32)
stop = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60
bad = []; calls = [0]
def work(t):
    r = random.Random(t)
    n = 0
    while time.time() < stop:
        i = r.randrange(len(imgs))
        if r.random() < 0.2:
            d_in = torch.from_numpy(imgs[i]).cuda(); d_out = torch.empty(lone[i].shape, dtype=torch.uint8, device="cuda")
            s.process_device(d_in.data_ptr(), imgs[i].shape[1], imgs[i].shape[0], imgs[i].shape[2], d_out.data_ptr())
            out = d_out.cpu().numpy()
        else:
            out = s.process(imgs[i], push_params=False)
        if not np.array_equal(out, lone[i]): bad.append((t, i))
        n += 1
        if t == 0 and n % 50 == 0:
            s.set_option("merge_mixed", r.randrange(2)); s.set_option("merge", r.choice([16, 16, 4, 1]))
    calls[0] += n
th = [threading.Thread(target=work, args=(t,)) for t in range(24)]
t0 = time.time(); [x.start() for x in th]; [x.join() for x in th]
print("soak: %d calls from 24 threads in %.0f s, %d merged batches (%d mixed, widest %d), %d mismatches" % (
    calls[0], time.time() - t0, s.get_stat("merged_batches"), s.get_stat("merged_mixed"), s.get_stat("merged_widest"), len(bad)))
s.close()
sys.exit(1 if bad else 0)
