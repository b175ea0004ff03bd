"""Probe: does splitting a frame's tile rows over two contexts on ONE GPU (two streams, two workspaces) beat one
context?  The tails of one stream's launches are back-filled by the other stream's workgroups.
    python tools/two_stream_probe.py"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib
R = importlib.import_module("realsr-ncnn-vulkan_amd")
synth = importlib.import_module("realsr-ncnn-vulkan_amd.synth")

MODELS = os.environ.get("RSR_MODELS", "/tmp/rsr_models")
d = synth.make_model_dir(MODELS, "models-DF2K", 42)
pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
w, h = 1920, 1080
img = synth.make_image(3, w, h)
pin = R.PinnedArray((h, w, 3)); pin.array[:] = img
out1 = R.PinnedArray((h * 4, w * 4, 3))
out2 = R.PinnedArray((h * 4, w * 4, 3))

def mk():
    s = R.RealSR(0); s.load(pp, bp); s.tilesize = 200; return s

a = mk()
a.process(pin.array, out=out1.array)
t = time.time()
for _ in range(5): a.process(pin.array, out=out1.array, push_params=False)
t1 = (time.time() - t) / 5
print("one context, whole frame: %.1f ms" % (t1 * 1e3))

b = mk()
for split in (3, 2, 4):
    def run(s, r0, r1, n):
        for _ in range(n): s.process_rows(pin.array, out2.array, r0, r1)
    for n in (1, 5):
        th = [threading.Thread(target=run, args=(a, 0, split, n)), threading.Thread(target=run, args=(b, split, 6, n))]
        t = time.time()
        for x in th: x.start()
        for x in th: x.join()
        dt = (time.time() - t) / n
    print("two contexts, rows [0,%d) | [%d,6): %.1f ms  identical=%s" % (split, split, dt * 1e3, bool((out1.array == out2.array).all())))
# frames alternating: two whole frames in flight on two contexts
def runf(s, o, n):
    for _ in range(n): s.process(pin.array, out=o, push_params=False)
th = [threading.Thread(target=runf, args=(a, out1.array, 5)), threading.Thread(target=runf, args=(b, out2.array, 5))]
t = time.time()
for x in th: x.start()
for x in th: x.join()
print("two contexts, one frame each, 5 rounds: %.1f ms per frame" % ((time.time() - t) / 10 * 1e3))
