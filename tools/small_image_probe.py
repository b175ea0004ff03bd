#!/usr/bin/env python3
"""Small images on one GPU (the reference's "-j 4:4:4 for many small images" mode, main.cpp:708-711,811-828, README.md:61):
what do (a) several caller threads on one context, (b) several contexts (= compute streams + workspaces) on one GPU and
(c) several images merged into ONE tile batch buy?  Device-resident images, 256x256 at tile 128 (4 tiles of 148x148).
    python tools/small_image_probe.py [frames]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
FR = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def mk(T=128, **opt):
    s = R.RealSR(0); s.load(pp, bp); s.tilesize = T
    for k, v in opt.items():
        s.set_option(k, v)
    return s


def bench(ctxs, threads_per_ctx, w, h, frames, host=False):
    """frames images of w x h in total, dealt to len(ctxs) * threads_per_ctx caller threads"""
    img = synth.make_image(5, w, h)
    nthr = len(ctxs) * threads_per_ctx
    if host:
        pin = R.PinnedArray((h, w, 3)); pin.array[:] = img
        outs = [R.PinnedArray((h * 4, w * 4, 3)) for _ in range(nthr)]
        call = lambda c, i: c.process(pin.array, out=outs[i].array, push_params=False)
    else:
        d_in = torch.from_numpy(img).cuda()
        outs = [torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda") for _ in range(nthr)]
        call = lambda c, i: c.process_device(d_in.data_ptr(), w, h, 3, outs[i].data_ptr())
    for i in range(nthr):
        call(ctxs[i % len(ctxs)], i)
    torch.cuda.synchronize()
    per = frames // nthr

    def work(i):
        c = ctxs[i % len(ctxs)]
        for _ in range(per):
            call(c, i)
    t = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthr)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize()
    dt = time.time() - t
    n = per * nthr
    return dt / n * 1e3, 16.0 * w * h * n / dt / 1e6


a = mk()
print("256x256 tile 128, device-resident images, %d frames" % FR)
for merge in (1, 16):
    a.set_option("merge", merge)
    for nthr in (1, 2, 4, 8, 16, 32):
        b0, i0 = a.get_stat("merged_batches"), a.get_stat("merged_images")
        ms, mp = bench([a], nthr, 256, 256, max(FR, 4 * nthr))
        nb, ni = a.get_stat("merged_batches") - b0, a.get_stat("merged_images") - i0
        print("  merge=%2d, 1 context x %2d caller threads: %.2f ms per image, %.1f Mpix/s   (%.1f images per merged batch)" % (
            merge, nthr, ms, mp, ni / nb if nb else 1.0), flush=True)
a.set_option("merge", 1)
more = [mk(merge=1) for _ in range(3)]
for nc in (2, 4):
    ms, mp = bench([a] + more[:nc - 1], 1, 256, 256, FR)
    print("  merge= 1, %d contexts x 1 thread: %.2f ms per image, %.1f Mpix/s" % (nc, ms, mp), flush=True)
for c in more:
    c.close()
print("one image of the same tile count as k 256x256 images (what a merged batch of k can reach), 1 context x 1 thread:")
for k, (w, h) in ((2, (512, 256)), (4, (512, 512)), (8, (1024, 512)), (16, (1024, 1024))):
    ms, mp = bench([a], 1, w, h, max(4, FR // k))
    print("  k=%2d (%dx%d, %d tiles): %.2f ms per batch = %.2f ms per 256x256 image, %.1f Mpix/s" % (k, w, h, 4 * k, ms, ms / k, mp), flush=True)
print("host -> host (pinned), 256x256:")
for merge in (1, 16):
    a.set_option("merge", merge)
    for nthr in (1, 4, 16):
        ms, mp = bench([a], nthr, 256, 256, max(FR, 4 * nthr), host=True)
        print("  merge=%2d, 1 context x %2d caller threads: %.2f ms per image, %.1f Mpix/s" % (merge, nthr, ms, mp), flush=True)
print("other small geometries, merge=16, 16 threads (device-resident): ")
for (w, h, T) in ((128, 128, 128), (200, 200, 200), (320, 240, 200), (512, 512, 200), (640, 480, 200)):
    for merge in (1, 16):
        a.set_option("merge", merge); a.tilesize = T
        ms, mp = bench([a], 16, w, h, 64)
        print("  %dx%d tile %d merge=%2d: %.2f ms per image, %.1f Mpix/s (merge width %d)" % (w, h, T, merge, ms, mp, a.get_stat("merged_widest")), flush=True)
# a directory of thumbnails: 64 images of 16 different sizes (48 .. 320 pixels a side), 16 caller threads, tile 200, device-resident
import numpy as np
rng = np.random.default_rng(3)
sizes = [(int(rng.integers(48, 321)), int(rng.integers(48, 321))) for _ in range(16)]
d_ins = [torch.from_numpy(synth.make_image(900 + i, *sizes[i % 16])).cuda() for i in range(64)]
d_outs = [torch.empty((sizes[i % 16][1] * 4, sizes[i % 16][0] * 4, 3), dtype=torch.uint8, device="cuda") for i in range(64)]
mpix = sum(16.0 * w * h for (w, h) in sizes) * 4 / 1e6
a.tilesize = 200
print("64 images of 16 different sizes (48 .. 320 px a side, %.1f Mpix of output), tile 200, 16 caller threads:" % mpix)
for label, merge, mixed in (("not merged", 1, 1), ("merged, one size per batch", 16, 0), ("merged across sizes", 16, 1)):
    a.set_option("merge", merge); a.set_option("merge_mixed", mixed)
    b0, m0 = a.get_stat("merged_batches"), a.get_stat("merged_mixed")

    def work(t):
        for i in range(4 * t, 4 * t + 4):
            a.process_device(d_ins[i].data_ptr(), sizes[i % 16][0], sizes[i % 16][1], 3, d_outs[i].data_ptr())
    for rep in range(3):
        if rep == 1:
            torch.cuda.synchronize(); t0 = time.time(); b0, m0 = a.get_stat("merged_batches"), a.get_stat("merged_mixed")
        th = [threading.Thread(target=work, args=(t,)) for t in range(16)]
        [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 2
    print("  %-28s %.1f ms per pass = %.1f Mpix/s  (%d batches per pass, %d mixed)" % (label, dt * 1e3, mpix / dt, (a.get_stat("merged_batches") - b0) / 2,
                                                                                     (a.get_stat("merged_mixed") - m0) / 2), flush=True)
a.close()
