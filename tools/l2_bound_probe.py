"""What would the dense-block convolutions cost if their operands never left the chip?  An UPPER BOUND for any cross-layer fusion
that keeps the RDB working set in the L2 (VERDICT r03 "take the dense block off HBM").  Not a pytest; run on the MI355X box:

    python tools/l2_bound_probe.py            # -> table on stdout (profiles/r04_l2_bound.txt)

One launch of the production kernel over the SAME small tile repeated N times (rsr_set_option "test_repeat", the rsr_conv3x3 /
rsr_conv3x3_res hooks): after the first pass every patch row, every weight and every output line of the launch is a cache hit.
Three footprints (input + output planes of the tile):
    "L2"    64 x 128 px  = 16 blocks:  <= 3.1 MB, inside ONE XCD's 4 MB L2 (every XCD walks the whole tile)
    "MALL"  256 x 512 px = 256 blocks: 25 - 67 MB, outside the L2s, inside the 256 MB Infinity Cache
    "HBM"   the C2 frame through rsr_process_device (per-class HIP-event times), 5236 blocks per launch, 488 - 977 MB
Reported per class: microseconds per 256 blocks (= one block per workgroup) and the TFLOP/s that is, so the three columns compare
directly.  The repeated-tile launches run ~40,000 blocks (160 per workgroup) like 8 frames' worth of one conv.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

MODELS = os.environ.get("RSR_MODELS", "/tmp/rsr_models")
CLASSES = [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]


def one(sr, cin, cout, h, w, repeat, rng):
    x = rng.standard_normal((cin, h, w)).astype(np.float16)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    sr.set_option("test_repeat", repeat)
    best = None
    for _ in range(3):
        if cout == 64:  # the RDB conv5 form: 0.2 * conv + x (identity tap on the matrix pipe)
            sr.conv3x3_res(x, wt, b, 0.2, own_input_residual=True)
        else:
            sr.conv3x3(x, wt, b, lrelu=True)
        us = sr.get_stat("last_test_us")
        best = us if best is None else min(best, us)
    sr.set_option("test_repeat", 1)
    blocks = repeat * (-(-h // 16)) * (-(-w // 32))
    return best, blocks


def frame_classes(sr):
    import torch
    w, h = 1920, 1080
    img = synth.make_image(3, w, h)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
    sr.tilesize = 200
    for _ in range(2):
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
    sr.set_profiling(True)
    best = None
    for _ in range(3):
        sr.get_conv_times(reset=True)
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        ct = sr.get_conv_times()
        cur = {}
        for i, (cin, cout, act) in enumerate(synth.conv_specs()):
            if 1 <= i <= 345:
                cur.setdefault((cin, cout), []).append(ct[i] * 1e3)
        cur = {k: sum(v) / len(v) for k, v in cur.items()}
        best = cur if best is None else {k: min(best[k], cur[k]) for k in cur}
    sr.set_profiling(False)
    return best


def main():
    d = synth.make_model_dir(MODELS, "models-DF2K", 42)
    sr = R.RealSR(0)
    sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
    rng = np.random.default_rng(0)
    fr = frame_classes(sr)
    print("class      | footprint  blocks   us/launch  us/256blk  TFLOP/s (%% of 2.5 PF)")
    for cin, cout in CLASSES:
        flop_blk = 2.0 * 9 * cin * cout * 512
        rows = []
        for name, (h, w, rep) in (("L2", (64, 128, 2560)), ("MALL", (256, 512, 160))):
            us, blocks = one(sr, cin, cout, h, w, rep, rng)
            rows.append((name, (cin + cout) / 16 * 32 * h * w / 1e6, blocks, us))
        rows.append(("HBM", (cin + cout) / 16 * 32 * 2544000 / 1e6, 5236, fr[(cin, cout)]))
        for name, mb, blocks, us in rows:
            per = us / (blocks / 256.0)
            # the frame's blocks are partly empty (220 -> 224 rows / columns): its rate is the algorithmic one, like bench.py's
            tf = (2.0 * 9 * cin * cout * 2544000 if name == "HBM" else flop_blk * blocks) / us / 1e6
            print("%3d -> %-2d | %-4s %7.1f MB %6d %10.1f %9.2f  %7.1f (%4.1f %%)" % (cin, cout, name, mb, blocks, us, per, tf, tf / 25), flush=True)
    sr.close()


if __name__ == "__main__":
    main()
