"""Per-half-stage barrier stamps of workgroup 0 / MFMA wave 0 of conv3x3_flow (needs a -DRSR_EXPERIMENT -DRSR_FLOW_TRACE build:
tools/build_variant.sh trace conv_flow "-DRSR_EXPERIMENT -DRSR_FLOW_TRACE"; RSR_LIB=.../lib/exp/trace.so python tools/flow_trace.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
sr = R.RealSR(0)
sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
sr.tilesize = 200
img = synth.make_image(3, 1920, 1080)
sr.process(img)
for ci, name, nst in [(1, "64->32", 4), (2, "96->32", 6), (4, "160->32", 10), (5, "192->64", 12)]:
    sr.set_option("trace_conv", ci)
    sr.process(img)
    tr = sr.get_trace(1024).astype(np.int64).reshape(-1, 2)
    n = int((tr[:, 0] > 0).sum())
    tr = tr[:n]
    arrive, release = tr[:, 0], tr[:, 1]
    wait = release - arrive
    period = arrive[1:] - arrive[:-1]  # one half-stage of MFMA stream incl. its barrier
    first = (np.arange(n - 1) % nst) == 0
    print("conv %d (%s): %d half-stages traced, total %d ticks = %.0f per half-stage (ideal 1152 per MFMA wave on the SIMD, x2 at NT=2)" % (
        ci, name, n, release[-1] - arrive[0], (release[-1] - arrive[0]) / max(n - 1, 1)))
    print("    barrier wait: mean %.0f p50 %.0f p90 %.0f max %d" % (wait[1:].mean(), np.median(wait[1:]), np.quantile(wait[1:], 0.9), wait[1:].max()))
    print("    period: block-first half-stages mean %.0f p50 %.0f | others mean %.0f p50 %.0f" % (
        period[first].mean(), np.median(period[first]), period[~first].mean(), np.median(period[~first])))
    print("    first 14 (wait, period):", [(int(w), int(p)) for w, p in zip(wait[1:15], period[:14])])
sr.set_option("trace_conv", -1)
sr.close()
