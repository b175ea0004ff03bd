"""GPU tests of cross-call merging (engine.h: small images of concurrent rsr_process calls walk the network as ONE tile batch).

The reference runs concurrent process() calls of its proc threads side by side on the device (main.cpp:811-828; README.md:61 "-j 4:4:4 for
many small images"); here they share the launches.  A tile is computed the same way whoever shares its batch, so every image must come
out BYTE-IDENTICAL to the one a lone call produces -- host and device API, RGB (conv_last writes the images itself), RGBA and TTA
(planar blob + postproc), mixed geometries in flight at once, merging on and off."""
import os
import threading

import numpy as np
import pytest

import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def paths(model_dir):
    return os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin")


def run_threads(n, fn):
    errs = []

    def wrap(i):
        try:
            fn(i)
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=wrap, args=(i,)) for i in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


@pytest.mark.parametrize("c,tta,T", [(3, 0, 32), (4, 0, 32), (3, 1, 64)])
def test_merged_images_equal_lone_calls(paths, oracle_net, c, tta, T):
    """12 caller threads x 3 images each of TWO geometries on one context, host API: byte-identical to the serial outputs; batches of
    more than one image did form; one image of the lot is also held against the oracle (+-1)."""
    s = R.RealSR(0, tta_mode=bool(tta))
    try:
        s.load(*paths)
        s.tilesize = T
        geos = [(70, 52), (45, 33)]
        imgs = [synth.make_image(100 + i, *geos[i % 2], c) for i in range(36)]
        s.set_option("merge", 1)
        lone = [s.process(im) for im in imgs]
        assert s.get_stat("merged_batches") == 0
        s.set_option("merge", 16)
        assert s.get_stat("merged_batches") == 0
        outs = [None] * 36

        def work(t):
            for k in range(3):
                i = t * 3 + k
                outs[i] = s.process(imgs[i], push_params=False)
        run_threads(12, work)
        for i in range(36):
            assert np.array_equal(outs[i], lone[i]), i
        nb, ni, widest, mixed = s.get_stat("merged_batches"), s.get_stat("merged_images"), s.get_stat("merged_widest"), s.get_stat("merged_mixed")
        print("c=%d tta=%d: %d images in %d merged batches (%d of them with images of both sizes), widest %d" % (c, tta, ni, nb, mixed, widest))
        assert ni == 36 and nb < 36 and widest >= 2 and mixed >= 1
        ref = oracle_net.process(imgs[5], T, tta=bool(tta))
        assert np.abs(outs[5].astype(int) - ref.astype(int)).max() <= 1
    finally:
        s.close()


def test_merged_batches_of_many_different_sizes(paths):
    """A directory of thumbnails: 40 images of 20 different sizes (1 .. 9 tiles each, some smaller than the halo, one column / one row
    ones) from 10 caller threads.  Batches form across sizes -- their tile and work-item tables are built on the fly (Engine::enqueue_mixed)
    -- and every image equals its lone call; with option merge_mixed = 0 only images of one size share a batch, same bytes."""
    s = R.RealSR(0)
    try:
        s.load(*paths)
        s.tilesize = 32
        rng = np.random.default_rng(7)
        sizes = [(int(rng.integers(3, 100)), int(rng.integers(3, 100))) for _ in range(18)] + [(97, 5), (4, 90)]
        imgs = [synth.make_image(700 + i, *sizes[i % 20]) for i in range(40)]
        s.set_option("merge", 1)
        lone = [s.process(im) for im in imgs]
        for mixed in (1, 0):
            s.set_option("merge", 16)
            s.set_option("merge_mixed", mixed)
            m0, b0 = s.get_stat("merged_mixed"), s.get_stat("merged_batches")
            outs = [None] * 40

            def work(t):
                for i in range(4 * t, 4 * t + 4):
                    outs[i] = s.process(imgs[i], push_params=False)
            run_threads(10, work)
            for i in range(40):
                assert outs[i].shape == lone[i].shape and np.array_equal(outs[i], lone[i]), (mixed, i, sizes[i % 20])
            nm, nb = s.get_stat("merged_mixed") - m0, s.get_stat("merged_batches") - b0
            print("merge_mixed=%d: 40 images of 20 sizes in %d batches, %d of them mixed" % (mixed, nb, nm))
            assert (nm >= 1 and nb < 40) if mixed else nm == 0
    finally:
        s.close()


def test_process_many_from_one_thread(paths, oracle_net):
    """rsr_process_many: 30 images of 6 sizes (RGB and RGBA mixed: they never share a batch) handed over in ONE call by ONE thread --
    merged batches form among the call's helper threads, every output equals the lone rsr_process call's; a bad image in the list fails
    with its index and the others are still done; n = 0 is a no-op."""
    s = R.RealSR(0)
    try:
        s.load(*paths)
        s.tilesize = 32
        sizes = [(40, 30, 3), (33, 21, 4), (64, 64, 3), (20, 70, 3), (50, 50, 4), (90, 16, 3)]
        imgs = [synth.make_image(800 + i, *sizes[i % 6]) for i in range(30)]
        s.set_option("merge", 1)
        lone = [s.process(im) for im in imgs]
        s.set_option("merge", 16)
        b0 = s.get_stat("merged_batches")
        outs = s.process_many(imgs)
        assert len(outs) == 30 and all(np.array_equal(a, b) for a, b in zip(outs, lone))
        nb = s.get_stat("merged_batches") - b0
        print("process_many: 30 images in %d merged batches" % nb)
        assert nb < 30
        assert s.process_many([]) == []
        ref = oracle_net.process(imgs[7], 32)
        assert np.abs(outs[7].astype(int) - ref.astype(int)).max() <= 1
        L = R.lib()
        import ctypes as C
        n = 3
        good = [imgs[0], imgs[2], imgs[3]]
        o = [np.zeros_like(lone[0]), np.zeros_like(lone[2]), np.zeros_like(lone[3])]
        ins_p = (C.c_void_p * n)(good[0].ctypes.data, None, good[2].ctypes.data)  # image 1: null pointer
        outs_p = (C.c_void_p * n)(*[x.ctypes.data for x in o])
        ws, hs, cs = (C.c_int * n)(40, 64, 20), (C.c_int * n)(30, 64, 70), (C.c_int * n)(3, 3, 3)
        rcs = (C.c_int * n)()
        rc = L.rsr_process_many(s._h, n, ins_p, ws, hs, cs, outs_p, rcs)
        assert rc == R.RSR_E_ARG and list(rcs) == [0, R.RSR_E_ARG, 0] and b"image 1" in L.rsr_last_error(s._h)
        assert np.array_equal(o[0], lone[0]) and np.array_equal(o[2], lone[3])
    finally:
        s.close()


def test_merged_device_api_and_precise_mode(paths):
    """The device API (synchronous rsr_process_device calls from 8 threads) merges too; precise mode (lo planes in the slots) likewise."""
    import torch
    s = R.RealSR(0)
    try:
        s.load(*paths)
        s.tilesize = 64
        for precise in (0, 1):
            s.set_option("precise", precise)
            imgs = [synth.make_image(300 + i, 90, 70) for i in range(16)]
            d_in = [torch.from_numpy(im).cuda() for im in imgs]
            d_out = [torch.zeros((280, 360, 3), dtype=torch.uint8, device="cuda") for _ in imgs]
            s.set_option("merge", 1)
            lone = [s.process(im) for im in imgs]
            s.set_option("merge", 16)
            b0 = s.get_stat("merged_batches")

            def work(t):
                for i in (2 * t, 2 * t + 1):
                    s.process_device(d_in[i].data_ptr(), 90, 70, 3, d_out[i].data_ptr())
            run_threads(8, work)
            torch.cuda.synchronize()
            for i in range(16):
                assert np.array_equal(d_out[i].cpu().numpy(), lone[i]), (precise, i)
            assert s.get_stat("merged_batches") - b0 < 16
    finally:
        s.close()


def test_large_images_and_tile_ranges_are_not_merged(paths):
    """An image that fills the chip by itself (more than a quarter of the work items a merged batch aims at) and calls over tile ranges
    keep the single-call path (split tail, early download): no merged batch is counted, bytes as ever."""
    s = R.RealSR(0)
    try:
        s.load(*paths)
        s.tilesize = 200
        img = synth.make_image(9, 1000, 700)  # 20 tiles of 220 x 220 = 1,960 work items
        a = s.process(img)
        assert s.get_stat("merged_batches") == 0
        s.tilesize = 32
        small = synth.make_image(10, 64, 64)
        out = np.zeros((256, 256, 3), np.uint8)
        s.process_rows(small, out, 0, 1)
        assert s.get_stat("merged_batches") == 0
        b = s.process(small)
        assert s.get_stat("merged_batches") == 1 and np.array_equal(out[:128], b[:128])
        s.set_option("merge_target_items", 8)  # nothing is small any more
        assert np.array_equal(s.process(small), b) and s.get_stat("merged_batches") == 1
        s.tilesize = 200
        assert np.array_equal(s.process(img), a)
    finally:
        s.close()


def test_merged_batches_under_a_small_workspace_budget(paths):
    """max_workspace_mb far below what a merged batch wants: the cached plan of a geometry is cut into several tile batches (a narrower
    merged batch launches a prefix of each), a batch of mixed sizes that does not fit falls back to one image at a time -- same bytes."""
    s = R.RealSR(0)
    try:
        s.load(*paths)
        s.tilesize = 64
        imgs = [synth.make_image(900 + i, *((150, 120) if i % 3 else (100, 70))) for i in range(18)]
        s.set_option("merge", 1)
        lone = [s.process(im) for im in imgs]
        s.set_option("merge", 16)
        s.set_option("max_workspace_mb", 200)  # one slot of 84 x 84 px x 6,048 B = 43 MB: four or so tiles per batch
        for mixed in (0, 1):
            s.set_option("merge_mixed", mixed)
            outs = [None] * 18

            def work(t):
                for i in range(3 * t, 3 * t + 3):
                    outs[i] = s.process(imgs[i], push_params=False)
            run_threads(6, work)
            for i in range(18):
                assert np.array_equal(outs[i], lone[i]), (mixed, i)
        assert s.get_stat("merged_batches") > 0 and s.get_stat("workspace_mb") <= 260
    finally:
        s.close()


def test_a_failing_merged_batch_reports_to_every_caller(paths):
    """A batch whose workspace cannot be had (test hook ws_fail_above_mb) fails EVERY call that was merged into it with RSR_E_NOMEM -- nobody
    hangs, nobody gets a stale image -- and the context works again once the cause is gone."""
    s = R.RealSR(0)
    try:
        s.load(*paths)
        s.tilesize = 32
        imgs = [synth.make_image(400 + i, 60, 60) for i in range(8)]
        good = [s.process(im) for im in imgs]
        s.set_option("ws_fail_above_mb", 0)
        codes = [None] * 8

        def work(i):
            try:
                s.process(imgs[i], push_params=False)
                codes[i] = 0
            except R.RealSRError as e:
                codes[i] = e.code
        run_threads(8, work)
        assert codes == [R.RSR_E_NOMEM] * 8, codes
        s.set_option("ws_fail_above_mb", -1)
        s.set_option("ws_clamp_mb", -1)
        outs = [None] * 8

        def work2(i):
            outs[i] = s.process(imgs[i], push_params=False)
        run_threads(8, work2)
        for i in range(8):
            assert np.array_equal(outs[i], good[i]), i
    finally:
        s.close()
