"""Oracle references for many tiles at once (test infrastructure): the CPU oracle runs 16 OpenMP threads well and no more
(oracle.py), the GPU box has 256 hardware threads -- so whole BASELINE frames are checked tile by tile with a pool of
worker PROCESSES, each holding its own OracleNet, as many as the CPU quota really allows (usable_cpus / 16; one 220x220
tile = 1.7 TFLOP of fp32 CPU work = ~4.5 s on 16 cores).

ref_tiles(pp, bp, [(padded CHW float32 tile, tta)]) -> [uint8 (4*th, 4*tw, 3) of the un-padded rectangle]
following realsr.cpp:525-838: network on the halo'd tile (x8 dihedral variants under TTA, realsr.cpp:617-724, merged
(sum) * 0.125), crop prepadding * 4, v * 255 + 0.5, truncate, clamp."""
import hashlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_NET = None


def _init(pp, bp, threads, child=True):
    global _NET
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    if child:  # a CPU-only worker process never needs torch; the calling process must not inherit the switch (its own
        os.environ["RSR_NO_TORCH"] = "1"  # subprocesses -- bench.py, tools/ -- would load the wrong HIP runtime first)
    import oracle
    _NET = oracle.OracleNet(pp, bp)
    oracle.set_threads(threads)


def _one(job):
    tile, tta, P = job
    if not tta:
        o = _NET.forward(np.ascontiguousarray(tile))
    else:
        acc = None
        for k in range(8):  # the 8 dihedral variants and their inverse maps, as tests/test_oracle.py states them independently
            t = tile
            if k & 4:
                t = t.transpose(0, 2, 1)
            if k & 1:
                t = t[:, ::-1]
            if k & 2:
                t = t[:, :, ::-1]
            r = _NET.forward(np.ascontiguousarray(t))
            if k & 2:
                r = r[:, :, ::-1]
            if k & 1:
                r = r[:, ::-1]
            if k & 4:
                r = r.transpose(0, 2, 1)
            acc = r.astype(np.float32) if acc is None else acc + r
        o = acc * np.float32(0.125)
    o = o[:, 4 * P:o.shape[1] - 4 * P, 4 * P:o.shape[2] - 4 * P]
    return np.clip((o * 255.0 + 0.5).astype(np.int32), 0, 255).astype(np.uint8).transpose(1, 2, 0)


def usable_cpus():
    """CPUs this process may really use: the affinity mask and the cgroup quota, not os.cpu_count() (the GPU box reports 256
    hardware threads and grants ~16 cores: 12 workers x 16 threads there ran no faster than one)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def ref_tiles(pp, bp, jobs, P=10, threads=16, workers=None):
    jobs = [(t, tta, P) for (t, tta) in jobs]
    cpus = usable_cpus()
    if workers is None:
        workers = int(os.environ.get("RSR_ORACLE_WORKERS", "0")) or max(1, min(len(jobs), cpus // threads, 12))
    threads = max(1, min(threads, cpus))
    if workers <= 1:
        _init(pp, bp, threads, child=False)
        return [_one(j) for j in jobs]
    ctx = mp.get_context("spawn")  # never fork a process that holds a HIP context
    with ctx.Pool(workers, initializer=_init, initargs=(pp, bp, threads)) as pool:
        return pool.map(_one, jobs, chunksize=1)


def padded_tile(img, x0, y0, tw, th, P=10):
    """The padded network input of the tile whose un-padded origin is (x0, y0): reflect-101 at the image border
    (realsr.cpp:613 copy_make_border type 2 == numpy 'reflect'), real neighbours elsewhere.  CHW float32 in [0,1]."""
    big = np.pad(img, ((P, P), (P, P), (0, 0)), mode="reflect")
    t = big[y0:y0 + th + 2 * P, x0:x0 + tw + 2 * P, :3]
    return t.astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)


def check_frame_tiles(out, img, pp, bp, T, tiles=None, tta=False, P=10):
    """Compare the tiles `tiles` (list of (xi, yi); None = every tile of the grid) of the engine's output `out` for `img` at
    tile size T with the oracle, +-1 uint8.  Returns (tiles checked, fraction of differing bytes)."""
    h, w = img.shape[:2]
    xt, yt = (w + T - 1) // T, (h + T - 1) // T
    if tiles is None:
        tiles = [(xi, yi) for yi in range(yt) for xi in range(xt)]
    geo = []
    for xi, yi in tiles:
        x0, y0 = xi * T, yi * T
        geo.append((x0, y0, min(x0 + T, w) - x0, min(y0 + T, h) - y0))
    refs = ref_tiles(pp, bp, [(padded_tile(img, *g, P=P), tta) for g in geo], P=P)
    diff = 0
    total = 0
    for (x0, y0, tw, th), ref in zip(geo, refs):
        got = out[4 * y0:4 * (y0 + th), 4 * x0:4 * (x0 + tw), :3].astype(int)
        d = np.abs(got - ref.astype(int))
        assert d.max() <= 1, "tile at (%d,%d) %dx%d: max diff %d" % (x0, y0, tw, th, d.max())
        diff += int((d > 0).sum())
        total += d.size
    return len(geo), diff / max(total, 1)


# ---- whole BASELINE frames: committed strided oracle samples + a rotating live subset ------------------------------------
GOLDEN = os.path.join(ROOT, "tests", "golden")


def check_frame_golden(out, name, img, bin_path, T):
    """Compare EVERY tile of the engine's frame `out` with the committed oracle samples tests/golden/frame_<name>.npz (the oracle's
    uint8 output on a lattice of every 8th row / column + every 64th at phase 7, written by tests/golden/make_frames.py), +-1.
    The file names the image and the weights it was made from; both are checked.  Returns (tiles covered, fraction of differing
    samples)."""
    z = np.load(os.path.join(GOLDEN, "frame_%s.npz" % name))
    assert hashlib.sha256(img.tobytes()).digest() == z["img_sha256"].tobytes(), "golden samples were made from another image"
    assert hashlib.sha256(open(bin_path, "rb").read()).digest() == z["bin_sha256"].tobytes(), "golden samples were made from other weights"
    rows, cols, ref = z["rows"], z["cols"], z["samples"].astype(np.int16)
    got = out[rows][:, cols, :3].astype(np.int16)
    assert got.shape == ref.shape
    d = np.abs(got - ref)
    h, w = img.shape[:2]
    xt, yt = (w + T - 1) // T, (h + T - 1) // T
    covered = 0
    for yi in range(yt):
        for xi in range(xt):
            r = (rows >= 4 * yi * T) & (rows < 4 * min((yi + 1) * T, h))
            c = (cols >= 4 * xi * T) & (cols < 4 * min((xi + 1) * T, w))
            assert r.sum() >= 4 and c.sum() >= 4, "tile (%d,%d) has too few samples" % (xi, yi)
            dt = d[r][:, c]
            assert dt.max() <= 1, "tile (%d,%d): max diff %d vs the golden oracle samples" % (xi, yi, dt.max())
            covered += 1
    return covered, float((d > 0).mean())


def rotating_tiles(xt, yt, k=6):
    """k tiles of an xt x yt grid for the LIVE oracle: the corner, one of the last column, two of the last row (the three edge-tile
    shapes) and interior ones.  The choice is FIXED (seed 6) so that a failure reproduces on the next run and runs are comparable
    (ADVICE r05); RSR_ROTATE=<int> picks another subset, RSR_ROTATE=day the calendar day's (a scheduled job that walks over the frame).
    Every tile is covered by the committed samples in any case (check_frame_golden)."""
    rot = os.environ.get("RSR_ROTATE", "6")
    seed = int(time.time() // 86400) if rot == "day" else int(rot)
    print("live-oracle tile subset: RSR_ROTATE=%d" % seed)
    rng = np.random.default_rng(seed)
    tiles = [(xt - 1, yt - 1)]
    if yt > 1:
        tiles.append((xt - 1, int(rng.integers(0, yt - 1))))
    if xt > 1:
        for xi in rng.choice(xt - 1, size=min(2, xt - 1), replace=False):
            tiles.append((int(xi), yt - 1))
    inner = [(xi, yi) for yi in range(max(1, yt - 1)) for xi in range(max(1, xt - 1))]
    for i in rng.permutation(len(inner)):
        if len(tiles) >= k:
            break
        if inner[i] not in tiles:
            tiles.append(inner[i])
    return tiles[:k]
