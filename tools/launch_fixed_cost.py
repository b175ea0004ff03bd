"""Per-launch fixed cost vs per-pixel cost of the conv classes: frames of 12 / 48 / 108 / 192 full 220x220 tiles, per-class
kernel time from the engine's HIP events, least-squares line t = t0 + k * padded_px.
    python tools/launch_fixed_cost.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
sr = R.RealSR(0)
sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
sr.tilesize = 200
specs = synth.conv_specs()
rows = {}
for (w, h) in ((800, 600), (1600, 1200), (2400, 1800), (3200, 2400)):
    img = synth.make_image(5, w, h)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
    torch.cuda.synchronize()
    sr.set_profiling(True)
    sr.get_conv_times(reset=True)
    n = 3
    for _ in range(n):
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
    ct = np.array(sr.get_conv_times()) / n
    sr.set_profiling(False)
    px = (w // 200) * (h // 200) * 220 * 220
    for cin in (64, 96, 128, 160, 192):
        idx = [i for i, (ci, co, act) in enumerate(specs) if ci == cin and i <= 346 and co in (32, 64) and not (cin == 64 and co == 64)]
        rows.setdefault(cin, []).append((px, 1e3 * ct[idx].mean()))
    del d_in, d_out
for cin, pts in rows.items():
    x = np.array([p[0] for p in pts], dtype=np.float64)
    y = np.array([p[1] for p in pts])
    k, t0 = np.polyfit(x, y, 1)
    cout = 64 if cin == 192 else 32
    print("%3d->%d: " % (cin, cout) + "  ".join("%.2f Mpx %.1f us" % (a / 1e6, b) for a, b in pts) +
          "  | fit: t0 = %.1f us, slope %.2f us/Mpx = %.0f TFLOP/s marginal" % (t0, k * 1e6, 2 * 9 * cin * cout / (k * 1e6) ))
sr.close()
