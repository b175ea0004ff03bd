#!/usr/bin/env python3
"""Two engine contexts on one GPU, each on half the CUs with MALL-sized tile batches, frames in flight concurrently.
   python tools/concurrent_probe.py  (GPU box) -- aggregate ms/frame for several (num_cu, workspace) settings"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

d = synth.make_model_dir("/tmp/rsr_models", "models-DF2K", 42)
w, h = 1920, 1080
d_in = torch.from_numpy(synth.make_image(3, w, h)).cuda()


def run(nctx, num_cu, ws, frames=4):
    ctxs, outs = [], []
    for i in range(nctx):
        sr = R.RealSR(0); sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")); sr.tilesize = 200
        sr.set_option("num_cu", num_cu); sr.set_option("max_workspace_mb", ws)
        o = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
        sr.process_device(d_in.data_ptr(), w, h, 3, o.data_ptr())
        ctxs.append(sr); outs.append(o)
    torch.cuda.synchronize()

    def work(i):
        for _ in range(frames):
            ctxs[i].process_device(d_in.data_ptr(), w, h, 3, outs[i].data_ptr())
    t = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nctx)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize()
    dt = time.time() - t
    print("contexts=%d num_cu=%d ws=%d: %.1f ms per frame aggregate (%.1f Mpix/s)" % (nctx, num_cu, ws, dt / (nctx * frames) * 1e3, 33.1776 * nctx * frames / dt), flush=True)
    same = all(bool((outs[0] == o).all()) for o in outs[1:])
    for c in ctxs:
        c.close()
    return same


for cfg in [(1, 256, 65536), (1, 256, 6100), (2, 256, 65536), (2, 256, 9400), (2, 256, 6100), (2, 256, 4000), (2, 256, 3100), (2, 256, 2200), (3, 256, 3100), (3, 256, 2200)]:
    print("  identical outputs:", run(*cfg))
