"""Is a conv launch ROUND-quantised (time ~ ceil(blocks / 256 workgroups)) or smooth in its block count -- and what does a block
cost whose waves are mostly dead?  Not a pytest; run on the MI355X box:

    python tools/round_probe.py            # -> table on stdout (profiles/r04_round_probe.txt)

A strip of k tiles of 220 x 220 padded pixels (image 200k x 200, tile 200) has 98 k blocks of 16 x 32 at the LR level.  The
persistent grid has 256 workgroups: k = 13 is 4.98 rounds, k = 14 is 5.36, ... .  If the per-class time follows ceil(rounds) the
last, partly filled round of a launch costs a whole block time and shaping that round pays; if it follows the block count the
launch is bound by a chip-level rate and only bytes / FLOPs help.  Second sweep: image height 172 / 176 / 180 / 184 / 188 gives
padded tiles of 192 / 196 / 200 / 204 / 208 rows = 12 full block rows + a 13th with 1 / 2 / 3 / 4 live waves of 4.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

MODELS = os.environ.get("RSR_MODELS", "/tmp/rsr_models")


def classes(ct):
    specs = synth.conv_specs()
    out = {}
    for i, (cin, cout, act) in enumerate(specs):
        if i > 346:
            continue
        g = out.setdefault("%d->%d" % (cin, cout), [0, 0.0])
        g[0] += 1
        g[1] += ct[i]
    return {k: v[1] / v[0] * 1e3 for k, v in out.items()}  # us per launch


def run(sr, w, h, reps=3):
    import torch
    img = synth.make_image(3, w, h)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
    sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
    torch.cuda.synchronize()
    sr.set_profiling(True)
    best = None
    for _ in range(reps):
        sr.get_conv_times(reset=True)
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        c = classes(sr.get_conv_times())
        best = c if best is None else {k: min(best[k], c[k]) for k in c}
    sr.set_profiling(False)
    return best


def main():
    d = synth.make_model_dir(MODELS, "models-DF2K", 42)
    sr = R.RealSR(0)
    sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
    sr.tilesize = 200
    keys = ["64->32", "96->32", "128->32", "160->32", "192->64"]
    print("== sweep 1: k tiles of 220x220 (98 blocks each); us per launch, min of 3")
    print("%3s %6s %6s | %s | us per ROUND (ceil) | us per 256 blocks (smooth)" % ("k", "blocks", "rounds", "  ".join("%8s" % k for k in keys)))
    for k in (13, 14, 15, 16, 18, 20, 21, 23, 26, 27, 30, 39, 41, 52, 54):
        c = run(sr, 200 * k, 200)
        nb = 98 * k
        rounds = nb / 256.0
        cr = -(-nb // 256)
        print("%3d %6d %6.2f | %s | %s | %s" % (k, nb, rounds, "  ".join("%8.1f" % c[x] for x in keys),
                                              " ".join("%6.2f" % (c[x] / cr) for x in keys), " ".join("%6.2f" % (c[x] / rounds) for x in keys)), flush=True)
    print("== sweep 2: 26 tiles of 220 x (H+20): 12 full block rows + a 13th with n live waves; us per launch")
    print("%4s %5s %6s | %s" % ("H", "live", "blocks", "  ".join("%8s" % k for k in keys)))
    for hh in (172, 176, 180, 184, 188):
        c = run(sr, 200 * 26, hh)
        rows = hh + 20
        nb = 26 * 7 * (-(-rows // 16))
        print("%4d %5d %6d | %s" % (hh, (rows - 192) // 4, nb, "  ".join("%8.1f" % c[x] for x in keys)), flush=True)
    sr.close()


if __name__ == "__main__":
    main()
