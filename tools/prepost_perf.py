"""pre / post kernel time and rate on the C5 frame (1080p, tile 200, TTA) and an RGBA 1080p frame, LDS-staged (dbg 65536) vs the
one-thread-per-pixel kernels (dbg 32768) vs what the launchers pick by default.  Run on the MI355X box:  python tools/prepost_perf.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K_JPEG", 43)
for name, tta, c in (("C5: 1080p RGB, TTA x8", True, 3), ("1080p RGBA (alpha bicubic), no TTA", False, 4), ("1080p RGB, two-kernel path (dbg 8192)", False, 3)):
    sr = R.RealSR(0, tta_mode=tta)
    sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
    sr.tilesize = 200
    w, h = 1920, 1080
    img = synth.make_image(1239, w, h, c)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.empty((h * 4, w * 4, c), dtype=torch.uint8, device="cuda")
    base = 8192 if "8192" in name else 0
    sums = []
    for dbg, label in ((base | 65536, "LDS-staged"), (base | 32768, "per-pixel "), (base, "default   ")):
        sr.set_option("dbg", dbg)
        sr.process_device(d_in.data_ptr(), w, h, c, d_out.data_ptr())
        sr.set_profiling(True)
        sr.get_profile(reset=True)
        n = 2
        for _ in range(n):
            sr.process_device(d_in.data_ptr(), w, h, c, d_out.data_ptr())
        p = sr.get_profile(reset=True)
        sr.set_profiling(False)
        sums.append(int(d_out[::97, ::89].to(torch.int64).sum().item()))
        print("%-40s %s: pre %.3f ms (%.0f GB/s)  post %.3f ms (%.0f GB/s)  frame %.1f ms" % (
            name, label, p["pre_ms"] / n, p["pre_bytes"] / max(p["pre_ms"], 1e-9) / 1e6, p["post_ms"] / n,
            p["post_bytes"] / max(p["post_ms"], 1e-9) / 1e6, p["total_ms"] / n), flush=True)
    assert sums[0] == sums[1] == sums[2], sums
    sr.close()
