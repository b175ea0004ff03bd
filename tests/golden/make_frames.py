"""Generates tests/golden/frame_c{2,3,5}.npz: STRIDED oracle samples of every tile of the BASELINE frames C2, C3 and C5.

Why: the live CPU oracle needs ~4 / ~14 / ~33 minutes per frame on the GPU box's host cores, so the default `-m gpu` run could
only afford 60 / 15 / 8 of the 60 tiles (missing #4 / weak #6).  These files hold the oracle's uint8 output of the
WHOLE frame on a lattice -- every 8th output row / column plus every 64th row / column at phase 7 (so that first AND last rows /
columns of the 800- resp. 1600-pixel output tiles are sampled) -- and the GPU tests compare every tile of the engine's frame with
them (+-1 uint8, the same bar as against the live oracle); the live oracle stays on a rotating 6-tile subset per config.

Like cases.npz these are outputs of the CPU *oracle* on the seeded synthetic models ("parity unpinned"): a
fixed target for the HIP path and a drift guard, not reference-held truth.

The oracle follows realsr.cpp:525-838 tile by tile (tests/oracle_pool.py: padded tile -> network (x8 dihedral variants under TTA,
realsr.cpp:617-724) -> crop prepadding*4 -> v*255+0.5 -> truncate -> clamp).  Full frames are cached under tests/golden/_full/
(git-ignored, 100 - 400 MB each; resumable tile by tile).

    python tests/golden/make_frames.py [c2] [c3] [c5]          # ~10 / ~35 / ~75 min on 8 cores
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# name: (model dir name, weight seed, image seed, w, h, tilesize, tta) -- the frames tests/test_gpu_round2.py and bench.py use
FRAMES = {
    "c2": ("models-DF2K", 42, 1235, 1920, 1080, 200, 0),
    "c3": ("models-DF2K", 42, 1236, 3840, 2160, 400, 0),
    "c5": ("models-DF2K_JPEG", 43, 1239, 1920, 1080, 200, 1),
}


def lattice(n):
    """Sampled output rows (or columns) of an axis of n pixels."""
    return np.unique(np.concatenate([np.arange(0, n, 8), np.arange(7, n, 64)])).astype(np.int32)


def main(names):
    import oracle_pool
    from realsr_ncnn_vulkan_amd import synth
    full_dir = os.path.join(HERE, "_full")
    os.makedirs(full_dir, exist_ok=True)
    for name in names:
        mdir, wseed, iseed, w, h, T, tta = FRAMES[name]
        d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), mdir, wseed)
        pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
        img = synth.make_image(iseed, w, h)
        full_path = os.path.join(full_dir, name + ".npy")
        done_path = os.path.join(full_dir, name + ".done.json")
        done = json.load(open(done_path)) if os.path.exists(done_path) and os.path.exists(full_path) else []
        full = np.lib.format.open_memmap(full_path, mode="r+" if done else "w+", dtype=np.uint8, shape=(4 * h, 4 * w, 3))
        xt, yt = (w + T - 1) // T, (h + T - 1) // T
        t0 = time.time()
        for yi in range(yt):
            for xi in range(xt):
                if [xi, yi] in done:
                    continue
                x0, y0 = xi * T, yi * T
                tw, th = min(x0 + T, w) - x0, min(y0 + T, h) - y0
                ref = oracle_pool.ref_tiles(pp, bp, [(oracle_pool.padded_tile(img, x0, y0, tw, th), bool(tta))], workers=1)[0]
                full[4 * y0:4 * (y0 + th), 4 * x0:4 * (x0 + tw)] = ref
                full.flush()
                done.append([xi, yi])
                json.dump(done, open(done_path, "w"))
                print("%s tile (%d,%d) %dx%d done, %d of %d, %.0f s" % (name, xi, yi, tw, th, len(done), xt * yt, time.time() - t0), flush=True)
        rows, cols = lattice(4 * h), lattice(4 * w)
        samp = np.ascontiguousarray(np.asarray(full)[rows][:, cols])
        np.savez_compressed(
            os.path.join(HERE, "frame_%s.npz" % name), rows=rows, cols=cols, samples=samp,
            cfg=np.array([wseed, iseed, w, h, T, tta], np.int32),
            bin_sha256=np.frombuffer(hashlib.sha256(open(bp, "rb").read()).digest(), np.uint8),
            img_sha256=np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8))
        print("%s: %d x %d samples written" % (name, len(rows), len(cols)), flush=True)


if __name__ == "__main__":
    main([a for a in sys.argv[1:] if a in FRAMES] or list(FRAMES))
