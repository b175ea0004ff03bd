"""EVERY pixel of every tile of a BASELINE frame (C2 / C3 / C5) against the oracle's full frame -- the complete form of the strided
check the default `-m gpu` run does (tests/oracle_pool.py check_frame_golden).

The oracle frames are the ones tests/golden/make_frames.py computes and caches under tests/golden/_full/<name>.npy (git-ignored,
100 - 400 MB each, ~10 / ~40 / ~95 minutes of CPU per frame): they travel to the GPU box with the snapshot when `.gpurunignore` lets
them.  Run on the MI355X box:
    python tests/full_frame_sweep.py c5 [c2] [c3] [c2:p ...]  -> one line per tile + a summary; exit code 1 on any |diff| > 1
(a name with the suffix ":p" runs the frame in precise mode, rsr_set_option precise 1)
Logs of such runs are kept under profiles/ (r05_c5_sweep.txt)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(names):
    import make_frames
    import realsr_ncnn_vulkan_amd as R
    from realsr_ncnn_vulkan_amd import synth
    bad = 0
    for name in names:
        precise = name.endswith(":p")
        name = name.split(":")[0]
        mdir, wseed, iseed, w, h, T, tta = make_frames.FRAMES[name]
        full = os.path.join(HERE, "golden", "_full", name + ".npy")
        if not os.path.exists(full):
            print("%s: %s is not here (make it with tests/golden/make_frames.py and let it travel)" % (name, full))
            bad += 1
            continue
        ref = np.load(full, mmap_mode="r")
        d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), mdir, wseed)
        pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
        img = synth.make_image(iseed, w, h)
        z = np.load(os.path.join(HERE, "golden", "frame_%s.npz" % name))  # the committed samples name image and weights
        assert hashlib.sha256(img.tobytes()).digest() == z["img_sha256"].tobytes() and hashlib.sha256(open(bp, "rb").read()).digest() == z["bin_sha256"].tobytes()
        assert (np.asarray(ref)[z["rows"]][:, z["cols"]] == z["samples"]).all(), "the cached full frame is not the one the committed samples were cut from"
        sr = R.RealSR(0, tta_mode=bool(tta))
        sr.load(pp, bp)
        sr.tilesize = T
        if precise:
            sr.set_option("precise", 1)
        out = sr.process(img)
        sr.close()
        xt, yt = (w + T - 1) // T, (h + T - 1) // T
        worst, ndiff, total, bad_tiles = 0, 0, 0, 0
        print("%s: %s, %dx%d, tile %d%s%s: %d x %d tiles, every output pixel against the oracle's frame (computed by tests/golden/make_frames.py)" % (
            name.upper(), mdir, w, h, T, ", TTA x8" if tta else "", ", PRECISE mode" if precise else "", xt, yt))
        for yi in range(yt):
            for xi in range(xt):
                y0, x0 = 4 * yi * T, 4 * xi * T
                y1, x1 = 4 * min((yi + 1) * T, h), 4 * min((xi + 1) * T, w)
                dd = np.abs(out[y0:y1, x0:x1].astype(np.int16) - np.asarray(ref[y0:y1, x0:x1]).astype(np.int16))
                m = int(dd.max())
                nd = int((dd > 0).sum())
                worst = max(worst, m)
                ndiff += nd
                total += dd.size
                print("  tile (%d,%d) %4dx%-4d max |diff| %d, %5.2f %% of the bytes differ%s" % (xi, yi, (x1 - x0) // 4 + 20, (y1 - y0) // 4 + 20, m, 100.0 * nd / dd.size,
                                                                                              "   <-- OUTSIDE +-1" if m > 1 else ""), flush=True)
                if m > 1:
                    bad_tiles += 1
        print("%s: %d of %d tiles within +-1 (max |diff| %d), %.2f %% of all %d bytes differ" % (name.upper(), xt * yt - bad_tiles, xt * yt, worst, 100.0 * ndiff / total, total), flush=True)
        bad += bad_tiles
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main([a for a in sys.argv[1:]] or ["c5"]))
