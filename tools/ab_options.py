"""Alternating A/B of engine options inside ONE process and ONE context (same board, same clocks, same allocations):
    python tools/ab_options.py --config C2|C3|C5 --variants "fold=1;fold=0" [--rounds 4] [--frames 6]
Every variant is a comma-separated list of rsr_set_option pairs (options a variant does not name keep the value the previous
variant left -- name them all).  Prints ms per frame per round, the mean per variant, and a checksum of the output (variants that
must not change the bytes must agree on it).  Frames are device-resident (rsr_process_device)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

CONFIGS = {  # name: (model, weight seed, image seed, w, h, tile, tta)
    "C2": ("models-DF2K", 42, 1235, 1920, 1080, 200, False),
    "C3": ("models-DF2K", 42, 1236, 3840, 2160, 400, False),
    "C5": ("models-DF2K_JPEG", 43, 1239, 1920, 1080, 200, True),
}
FLOP_PX = 35853696


def padded_px(w, h, T, P=10):
    return sum((min(x0 + T, w) - x0 + 2 * P) * (min(y0 + T, h) - y0 + 2 * P) for y0 in range(0, h, T) for x0 in range(0, w, T))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--variants", default="fold=1;fold=0")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--frames", type=int, default=6)
    a = ap.parse_args()
    model, wseed, iseed, w, h, T, tta = CONFIGS[a.config]
    d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), model, wseed)
    sr = R.RealSR(0, tta_mode=tta)
    sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
    sr.tilesize = T
    img = synth.make_image(iseed, w, h)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
    variants = [v for v in a.variants.split(";") if v]
    flops = padded_px(w, h, T) * FLOP_PX * (8 if tta else 1)
    times = {v: [] for v in variants}
    sums = {}
    for rnd in range(a.rounds + 1):  # round 0 = warm-up (allocations, plans), not reported
        for v in variants:
            for kv in v.split(","):
                k, val = kv.split("=")
                sr.set_option(k, int(val))
            sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(a.frames):
                sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / a.frames * 1e3
            if rnd:
                times[v].append(ms)
            sums[v] = int(d_out[::97, ::89].to(torch.int64).sum().item())
    print("%s %dx%d tile %d tta %d, %d rounds x %d frames, %s" % (a.config, w, h, T, tta, a.rounds, a.frames, os.environ.get("RSR_LIB", "default build")))
    for v in variants:
        m = sum(times[v]) / len(times[v])
        print("  %-28s %s  mean %.2f ms = %.1f %% of 2.5 PF  checksum %d" % (v, " ".join("%.2f" % x for x in times[v]), m, flops / m / 2.5e12 * 100, sums[v]), flush=True)
    sr.close()


if __name__ == "__main__":
    main()
