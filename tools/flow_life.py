"""Where does a conv launch spend its time OUTSIDE the steady state?  Per-workgroup s_memtime stamps (100 MHz REFCLK, 10 ns) at
kernel entry, at the first half-stage barrier (weights + first patch in LDS) and at exit, for all 256 workgroups of one launch.
Needs a -DRSR_FLOW_LIFE build:   tools/build_variant.sh life conv_flow "-DRSR_EXPERIMENT -DRSR_FLOW_LIFE"
                                 RSR_LIB=realsr-ncnn-vulkan_amd/lib/exp/life.so python tools/flow_life.py
Reports, per conv class: launch span (first entry -> last exit), entry skew, PROLOGUE (entry -> first half-stage ready) and exit
skew (first exit -> last exit: the tail in which workgroups idle), next to the HIP-event time of the same launch."""
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
print("s_memtime counters are not comparable between workgroups (per-CU offsets): only differences inside one workgroup are used.")
print("ticks/us is calibrated as median workgroup life (entry -> exit) / HIP-event time of the launch: persistent workgroups live for the whole launch")
print("conv                   | event us | ticks/us | prologue us min/med/max (%% of launch) | life us min / med / max")
for ci, name in [(1, "64->32"), (2, "96->32"), (3, "128->32"), (4, "160->32"), (5, "192->64"), (6, "64->32 (bwd order)"), (347, "64->64 @2x"), (349, "64->64 @4x"), (350, "64->3 @4x")]:
    rows = []
    for rep in range(3):
        sr.set_option("trace_conv", ci)
        sr.set_profiling(True)
        sr.get_conv_times(reset=True)
        sr.process(img)
        ev = sr.get_conv_times()[ci] * 1e3
        sr.set_profiling(False)
        tr = sr.get_trace(1024).astype(np.int64).reshape(256, 4)
        g = tr[tr[:, 0] > 0]
        pro, life = (g[:, 1] - g[:, 0]).astype(float), (g[:, 2] - g[:, 0]).astype(float)
        tpu = float(np.median(life)) / ev
        rows.append((ev, tpu, pro.min() / tpu, np.median(pro) / tpu, pro.max() / tpu, 100 * np.median(pro) / np.median(life), life.min() / tpu, np.median(life) / tpu, life.max() / tpu))
    r = min(rows, key=lambda v: v[0])
    print("%3d %-18s | %8.1f | %8.1f | %6.2f %6.2f %6.2f (%4.1f %%) | %7.1f / %7.1f / %7.1f" % ((ci, name) + r), flush=True)
sr.set_option("trace_conv", -1)
sr.close()
