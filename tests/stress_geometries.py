#!/usr/bin/env python3
"""One-off stress sweep (not collected by pytest; test infrastructure: it calls the oracle): N random geometries -- image sizes
2..140, RGB / RGBA, tile sizes 16..96 that divide nothing, prepadding 10, every fourth case TTA -- rsr_process against the oracle,
+-1 uint8.     python tests/stress_geometries.py [N [seed]]        (MI355X box; ~2 s per case, mostly CPU oracle time)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402,F401  (its HIP runtime first)
import oracle  # noqa: E402
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
net = oracle.OracleNet(pp, bp)
eng = {False: R.RealSR(0), True: R.RealSR(0, tta_mode=True)}
for e in eng.values():
    e.load(pp, bp)
rng = np.random.default_rng(seed)
t0 = time.time()
worst = 0
for i in range(n):
    tta = i % 4 == 3
    lim = 60 if tta else 140
    w, h, c, T = int(rng.integers(2, lim)), int(rng.integers(2, lim)), int(rng.choice([3, 4])), int(rng.integers(16, 97))
    img = synth.make_image(5000 + i, w, h, c)
    e = eng[tta]
    e.tilesize = T
    got = e.process(img)
    ref = net.process(img, T, tta=tta)
    dmax = int(np.abs(got.astype(int) - ref.astype(int)).max())
    worst = max(worst, dmax)
    if dmax > 1:
        print("FAIL case %d: %dx%dx%d tile %d tta %d: max diff %d" % (i, w, h, c, T, tta, dmax), flush=True)
        sys.exit(1)
print("%d random geometries (seed %d) against the oracle: all within +-1 uint8 (worst %d), %.0f s" % (n, seed, worst, time.time() - t0))
