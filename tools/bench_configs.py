#!/usr/bin/env python3
"""Secondary BASELINE configs on one GPU (device-resident in/out): C3 = 3840x2160 tile 400, C5 = 1920x1080 tile 200 TTA x8.
   python tools/bench_configs.py [C3] [C5]   (bench.py is the contract line for C2)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

FLOP_PX = 35853696


def padded_px(w, h, T, P=10):
    return sum((min(x0 + T, w) - x0 + 2 * P) * (min(y0 + T, h) - y0 + 2 * P) for y0 in range(0, h, T) for x0 in range(0, w, T))


def run(name, w, h, T, tta, seed, model, wseed, ws_mb=None, n=2):
    d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), model, wseed)
    sr = R.RealSR(0, tta_mode=tta)
    sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
    sr.tilesize = T
    if ws_mb:
        sr.set_option("max_workspace_mb", ws_mb)
    img = synth.make_image(seed, w, h)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
    sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    fl = padded_px(w, h, T) * FLOP_PX * (8 if tta else 1)
    print("%s: %dx%d T=%d tta=%d: %.1f ms/frame = %.1f output Mpix/s, %.1f TFLOP -> %.0f TFLOP/s (%.1f%% of 2.5 PF), checksum %d" % (
        name, w, h, T, tta, dt * 1e3, 16 * w * h / 1e6 / dt, fl / 1e12, fl / dt / 1e12, fl / dt / 2.5e15 * 100,
        int(d_out[::97, ::89].to(torch.int64).sum().item())), flush=True)
    sr.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["C3", "C5"]
    if "C3" in which:
        run("C3", 3840, 2160, 400, False, 1237, "models-DF2K", 42, ws_mb=200000)
    if "C5" in which:
        run("C5", 1920, 1080, 200, True, 1239, "models-DF2K_JPEG", 43, ws_mb=200000)
