#!/usr/bin/env python3
"""One-off parity sweep on BASELINE config C3 (3840x2160, tile 400, 10 x 6 tiles): every second tile of the frame (30 of 60,
all four tile shapes, both tile batches) against the oracle network, +-1 uint8.  ~10 min of CPU oracle time on the GPU box.
    python tests/c3_parity_sweep.py [stride [offset]]     (test infrastructure: it calls the oracle; not collected by pytest)
Round 3 ran stride 2 offset 0 (tiles 0, 2, ... + the corner: 31 tiles), round 4 stride 2 offset 1 (tiles 1, 3, ... 59: the other 30):
between them every tile of the frame has been checked (profiles/r03_slow_parity.txt, profiles/r04_c3_sweep.txt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):  # noqa: E402
    sys.path.insert(0, p)
import oracle_pool  # noqa: E402
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

stride = int(sys.argv[1]) if len(sys.argv) > 1 else 2
offset = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
sr = R.RealSR(0)
sr.load(pp, bp)
sr.tilesize = 400
img = synth.make_image(1236, 3840, 2160)
t = time.time()
out = sr.process(img)
print("C3 frame through rsr_process: %.2f s (first call: plan + workspace)" % (time.time() - t), flush=True)
sr.close()
tiles = [(xi, yi) for yi in range(6) for xi in range(10)][offset::stride]
if (9, 5) not in tiles:
    tiles.append((9, 5))
t = time.time()
n, frac = oracle_pool.check_frame_tiles(out, img, pp, bp, T=400, tiles=tiles)
print("C3: %d of 60 tiles checked against the oracle, all within +-1 uint8; %.2f %% of the bytes differ (%.0f s of oracle time)" % (n, 100 * frac, time.time() - t))
