#!/usr/bin/env python3
"""One-off stress sweep (not collected by pytest; test infrastructure: it calls the oracle): N random geometries -- image sizes
2..140, RGB / RGBA, tile sizes 16..96 that divide nothing, prepadding 10, every fourth case TTA -- rsr_process against the oracle,
+-1 uint8, in fp16-storage AND in precise mode (rsr_set_option precise 1); then the same images again, all at once, from 12 caller threads
(small images of concurrent calls are merged into shared tile batches, of mixed sizes): byte-identical to the lone calls.
    python tests/stress_geometries.py [N [seed]]        (MI355X box; ~2 s per case, mostly CPU oracle time)"""
import os
import sys
import threading
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
worstp = 0
cases = []
for i in range(n):
    tta = i % 4 == 3
    lim = 60 if tta else 140
    w, h, c, T = int(rng.integers(2, lim)), int(rng.integers(2, lim)), int(rng.choice([3, 4])), int(rng.integers(16, 97))
    img = synth.make_image(5000 + i, w, h, c)
    e = eng[tta]
    e.tilesize = T
    got = e.process(img)
    e.set_option("precise", 1)
    gotp = e.process(img)
    e.set_option("precise", 0)
    ref = net.process(img, T, tta=tta)
    dmax = int(np.abs(got.astype(int) - ref.astype(int)).max())
    dmaxp = int(np.abs(gotp.astype(int) - ref.astype(int)).max())
    worst = max(worst, dmax)
    worstp = max(worstp, dmaxp)
    cases.append((tta, T, img, got))
    if dmax > 1 or dmaxp > 1:
        print("FAIL case %d: %dx%dx%d tile %d tta %d: max diff %d (precise: %d)" % (i, w, h, c, T, tta, dmax, dmaxp), flush=True)
        sys.exit(1)
print("%d random geometries (seed %d) against the oracle: all within +-1 uint8 (worst %d; precise mode: worst %d), %.0f s" % (n, seed, worst, worstp, time.time() - t0))
# the same images from 12 concurrent callers per context; tile sizes differ per case, so the cases of one tile size go together
bad = []
for tta in (False, True):
    e = eng[tta]
    b0, m0 = e.get_stat("merged_batches"), e.get_stat("merged_mixed")
    byT = {}
    for k, (ct, T, img, got) in enumerate(cases):
        if ct == tta:
            byT.setdefault(T // 16, []).append((T, img, got))  # (callers of one group set the tile size once: equal T only)
    todo = [(T, img, got) for grp in byT.values() for (T, img, got) in grp]
    todo.sort(key=lambda t: t[0])
    i = 0
    while i < len(todo):
        j = i
        while j < len(todo) and todo[j][0] == todo[i][0]:
            j += 1
        e.tilesize = todo[i][0]
        e._push_params()
        chunk = todo[i:j] * 3  # every image three times: more concurrency
        outs = [None] * len(chunk)

        def work(t):
            for k in range(t, len(chunk), 12):
                outs[k] = e.process(chunk[k][1], push_params=False)
        th = [threading.Thread(target=work, args=(t,)) for t in range(12)]
        [x.start() for x in th]
        [x.join() for x in th]
        bad += [1 for k in range(len(chunk)) if not np.array_equal(outs[k], chunk[k][2])]
        i = j
    print("tta=%d: %d concurrent calls in %d merged batches (%d of mixed sizes): %s" % (
        tta, 3 * len(todo), e.get_stat("merged_batches") - b0, e.get_stat("merged_mixed") - m0, "all byte-identical to the lone calls" if not bad else "%d DIFFER" % len(bad)), flush=True)
sys.exit(1 if bad else 0)
