"""GPU parity tests (run with -m gpu on the MI355X box).  Everything goes through the C-ABI
(librealsr_hip.so via ctypes); the oracle is only the checker.

Tolerances (BASELINE.json north_star): final uint8 within +-1 per channel of the CPU path at the same
tile size; pre-quantise fp error reported and bounded (stated per test)."""
import os

import numpy as np
import pytest

import oracle
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def paths(model_dir):
    return os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin")


@pytest.fixture(scope="module")
def sr(paths):
    s = R.RealSR(0)
    s.load(*paths)
    yield s
    s.close()


@pytest.fixture(scope="module")
def sr_tta(paths):
    s = R.RealSR(0, tta_mode=True)
    s.load(*paths)
    yield s
    s.close()


def test_native_library_is_the_one_loaded(sr):
    maps = open("/proc/self/maps").read()
    assert "librealsr_hip.so" in maps


# ---- layer level ---------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,h,w,ups", [(64, 32, 20, 40, False), (96, 32, 17, 33, False), (128, 32, 16, 32, False),
                                              (160, 32, 33, 65, False), (192, 64, 16, 32, False), (3, 64, 9, 70, False),
                                              (64, 3, 33, 31, False), (64, 64, 10, 21, True), (64, 64, 1, 1, False)])
@pytest.mark.parametrize("lrelu", [False, True])
def test_conv3x3_layer_matches_oracle(sr, cin, cout, h, w, ups, lrelu):
    """fp16 inputs/weights, fp32 accumulate vs the oracle's fp32 conv on the same rounded operands:
    only accumulation order + the final fp16 rounding differ -> |d| <= 2^-10 * |ref| + 1e-3."""
    rng = np.random.default_rng(cin * 1000 + cout + h)
    x = rng.standard_normal((cin, h, w)).astype(np.float16)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    xr = x.astype(np.float32)
    if ups:
        xr = xr.repeat(2, axis=1).repeat(2, axis=2)
    ref = oracle.conv3x3(xr, wt, b, 2 if lrelu else 0, 0.2)
    try:
        # conv3x3_flow: 8 x 32 / 4 x 64 MFMA waves, with / without deferred epilogue, weights LDS-resident / streamed (flags 4),
        # rows below the tile skipped / computed
        for flags, dbg in ((0, 0), (3, 0), (4, 0), (7, 0), (0, 32), (3, 32)):
            sr.set_option("flow_flags", flags)
            sr.set_option("dbg", dbg)
            got = sr.conv3x3(x, wt, b, lrelu=lrelu, upsample2x=ups).astype(np.float32)
            assert got.shape == ref.shape
            assert (np.abs(got - ref) <= np.abs(ref) * 2.0 ** -10 + 1e-3).all(), "flags=%d dbg=%d max err %g" % (
                flags, dbg, np.abs(got - ref).max())
    finally:
        sr.set_option("flow_flags", 0)
        sr.set_option("dbg", 0)


@pytest.mark.parametrize("cin,cout,h,w,ups", [(64, 32, 100, 70, False), (160, 32, 52, 40, False), (192, 64, 36, 33, False),
                                              (64, 64, 22, 40, True), (64, 3, 41, 31, False)])
def test_rows_below_the_tile_are_skipped_not_miscomputed(sr, cin, cout, h, w, ups):
    """Tile heights that end in the first / second / third 4-row group of a 16-row block: MFMA waves whose rows lie wholly
    below the tile skip the block (conv_flow.hip: wave_is_dead).  Few workgroups (num_cu 8) give every wave live and dead
    blocks in a row -- deferred drain of a live block from inside a dead one, dead first block, dead last block.  The skipping
    build must give the SAME BYTES as the non-skipping one (dbg 32)."""
    rng = np.random.default_rng(cin + 7 * h)
    x = rng.standard_normal((cin, h, w)).astype(np.float16)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    try:
        for flags in (0, 3, 4):
            for ncu in (256, 8):
                sr.set_option("flow_flags", flags)
                sr.set_option("num_cu", ncu)
                sr.set_option("dbg", 32)
                ref = sr.conv3x3(x, wt, b, lrelu=True, upsample2x=ups)
                sr.set_option("dbg", 0)
                got = sr.conv3x3(x, wt, b, lrelu=True, upsample2x=ups)
                assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), (flags, ncu)
    finally:
        for k, v in (("flow_flags", 0), ("dbg", 0), ("num_cu", 256)):
            sr.set_option(k, v)


@pytest.mark.parametrize("cin,cout,h,w,ups", [(64, 32, 40, 44, False), (160, 32, 52, 36, False), (192, 64, 36, 46, False), (96, 32, 70, 10, False),
                                              (64, 64, 22, 19, True), (64, 3, 41, 34, False), (128, 32, 17, 33, False)])
def test_folded_blocks_equal_plain_blocks(sr, cin, cout, h, w, ups):
    """A tile's last block column of 1..14 pixels runs as FOLDED work items (kernels.h kFoldBit: two block rows of the narrow column
    in one 16 x 32 block; 420 = 13 x 32 + 4, 260 = 8 x 32 + 4, 140 = 4 x 32 + 12 are the padded tile widths of BASELINE C3 / C2,
    realsr.cpp:170-181,246-249).  Only the loaders' gather and the epilogue's scatter differ, every output value is accumulated in the
    same order: option "fold" 0 (one plain block column more) must give the SAME BYTES.  Heights whose second strip is partly /
    wholly below the tile, an unpaired last block row, a tile narrower than one block, the nearest-x2 gather, conv_last in both
    forms, every wave layout, few workgroups (a workgroup then meets folded and plain blocks back to back)."""
    rng = np.random.default_rng(cin + 11 * w)
    x = rng.standard_normal((cin, h, w)).astype(np.float16)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    try:
        for flags in ((0, 8) if cout == 3 else (0, 1, 2, 3, 4)):
            for ncu in (256, 8):
                sr.set_option("flow_flags", flags)
                sr.set_option("num_cu", ncu)
                sr.set_option("fold", 0)
                ref = sr.conv3x3(x, wt, b, lrelu=(cout != 3), upsample2x=ups)
                sr.set_option("fold", 1)
                got = sr.conv3x3(x, wt, b, lrelu=(cout != 3), upsample2x=ups)
                assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), (flags, ncu)
                if cout == 64 and not ups:  # the residual epilogues fetch / add through the same lane -> pixel map
                    res = rng.standard_normal((cout, h, w)).astype(np.float16)
                    for s1, own, rr, s2 in ((0.2, True, None, 1.0), (0.2, True, res, 0.2), (1.0, False, res, 1.0)):
                        sr.set_option("fold", 0)
                        ref = sr.conv3x3_res(x, wt, b, s1, own_input_residual=own, res=rr, s2=s2)
                        sr.set_option("fold", 1)
                        got = sr.conv3x3_res(x, wt, b, s1, own_input_residual=own, res=rr, s2=s2)
                        assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), (flags, ncu, s1, own, s2)
    finally:
        for k, v in (("flow_flags", 0), ("fold", 1), ("num_cu", 256)):
            sr.set_option(k, v)


@pytest.mark.parametrize("cin,cout,h,w", [(192, 64, 20, 40), (64, 64, 33, 50), (192, 64, 70, 90)])
def test_residual_epilogues_match_numpy(sr, cin, cout, h, w):
    """The Eltwise / BinaryOp layers behind a Convolution in x4.param, fused into its epilogue: RDB conv5
    (v = 0.2*conv + x, x4.param:17-18; x rides in the accumulator as an identity tap), every third
    RDB (v = 0.2*v + rrdb_in, x4.param:47) and trunk_conv + global skip (x4.param:994-995).  One extra fp16 rounding of
    the intermediate -> |d| <= 2^-9 |ref| + 2e-3.  num_cu = 8 gives every workgroup several blocks (ring wrap-around)."""
    rng = np.random.default_rng(cin + h)
    x = rng.standard_normal((cin, h, w)).astype(np.float16)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = rng.standard_normal((cout, h, w)).astype(np.float16)
    conv = oracle.conv3x3(x.astype(np.float32), wt, b, 0, 0.2)
    x0, r = x[:cout].astype(np.float32), res.astype(np.float32)
    forms = {"conv5": (0.2, True, None, 1.0, 0.2 * conv + x0), "conv5+rrdb": (0.2, True, res, 0.2, 0.2 * (0.2 * conv + x0) + r),
             "trunk": (1.0, False, res, 1.0, conv + r)}
    try:
        for flags, dbg, ncu in ((0, 0, 256), (1, 0, 256), (0, 0, 8), (1, 0, 8), (0, 32, 8), (4, 0, 8), (5, 0, 256)):
            for k, v in (("flow_flags", flags), ("dbg", dbg), ("num_cu", ncu)):
                sr.set_option(k, v)
            for name, (s1, own, rr, s2, ref) in forms.items():
                got = sr.conv3x3_res(x, wt, b, s1, own_input_residual=own, res=rr, s2=s2).astype(np.float32)
                assert (np.abs(got - ref) <= np.abs(ref) * 2.0 ** -9 + 2e-3).all(), (name, flags, dbg, ncu, np.abs(got - ref).max())
    finally:
        for k, v in (("flow_flags", 0), ("dbg", 0), ("num_cu", 256)):
            sr.set_option(k, v)


# ---- the four shaders: bit exact --------------------------------------------------------------------
@pytest.mark.parametrize("w,h,c", [(50, 43, 3), (50, 43, 4), (7, 9, 3), (200, 37, 3), (33, 64, 3)])
def test_preproc_kernels_bit_exact(sr, w, h, c):
    rng = np.random.default_rng(w * h + c)
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    T, P = 32, 10
    for yi in range((h + T - 1) // T):
        y0, y1 = max(yi * T - P, 0), min((yi + 1) * T + P, h)
        band = np.ascontiguousarray(img[y0:y1])
        for xi in range((w + T - 1) // T):
            twn = min((xi + 1) * T, w) - xi * T
            thn = min((yi + 1) * T, h) - yi * T
            tw, th = twn + 2 * P, thn + 2 * P
            a = (P, P, xi * T, min(yi * T, P))  # pad_top, pad_left, crop_x, crop_y: realsr.cpp:401-404
            if c == 4:
                r1, al1 = oracle.preproc(band, tw, th, *a, alphaw=twn, alphah=thn)
                r2, al2 = sr.preproc(band, tw, th, *a, alphaw=twn, alphah=thn)
                assert (al1.view(np.uint16) == al2.view(np.uint16)).all()
            else:
                r1 = oracle.preproc(band, tw, th, *a)
                r2 = sr.preproc(band, tw, th, *a)
                t1 = oracle.preproc_tta(band, tw, th, *a)
                t2 = sr.preproc_tta(band, tw, th, *a)
                for k in range(8):
                    assert (t1[k].view(np.uint16) == t2[k].view(np.uint16)).all(), "tta blob %d" % k
            assert (r1.view(np.uint16) == r2.view(np.uint16)).all()


@pytest.mark.parametrize("c", [3, 4])
def test_postproc_kernel_bit_exact(sr, c):
    rng = np.random.default_rng(c)
    tw, th, twn, thn = 30, 26, 10, 6
    bot = (rng.standard_normal((3, th * 4, tw * 4)) * 0.6 + 0.5).astype(np.float16)  # over/undershoots [0,1]
    alpha = (rng.random((thn * 4, twn * 4)) * 300 - 20).astype(np.float16) if c == 4 else None
    o1 = rng.integers(0, 256, (thn * 4, 100, c), dtype=np.uint8)
    o2 = o1.copy()
    oracle.postproc(bot, o1, 12, twn * 4, 40, 40, alpha=alpha)
    sr.postproc(bot, o2, 12, twn * 4, 40, 40, alpha=alpha)
    assert (o1 == o2).all()
    assert (o1[:, :12] == o2[:, :12]).all() and (o1[:, 52:] == o2[:, 52:]).all()  # untouched columns kept


def test_postproc_tta_kernel(sr):
    rng = np.random.default_rng(8)
    bots = [(rng.standard_normal((3, 104, 120) if k < 4 else (3, 120, 104)) * 0.6 + 0.5).astype(np.float16) for k in range(8)]
    o1 = rng.integers(0, 256, (24, 100, 3), dtype=np.uint8)
    o2 = o1.copy()
    oracle.postproc_tta(bots, o1, 12, 40, 40, 40)
    sr.postproc_tta(bots, o2, 12, 40, 40, 40)
    assert (o1 == o2).all()  # same fp32 sum order as the shader (v0+...+v7)*0.125


# ---- network on one tile ------------------------------------------------------------------------------
def test_network_tile_prequantise_error(sr, oracle_net):
    """Pre-quantise error of the `output` blob in [0,1] units, fp16 storage (the default).  Stated tolerance: max <= 3.0e-3 -- below one
    uint8 step, 1/255 = 3.92e-3, with margin, so that it IMPLIES the +-1 bar -- and p99.9 <= 1.5e-3 (fp16 storage / fp32 accumulate --
    the reference Vulkan path's arithmetic, realsr.cpp:44-46 -- vs fp32 everywhere, 351 convs; measured 1.6 - 2.1e-3 / 0.9 - 1.1e-3 on the
    stand-ins).  SURVEY 8(c)'s target of 2e-3 / 5e-4 is NOT met by fp16 storage (the trunk is rounded 92 times on its way through the
    RRDBs); precise mode meets it: tests/test_gpu_precise.py.  uint8 +-1."""
    img = synth.make_image(5, 44, 36)
    x = (img.astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)).astype(np.float16)
    ref = oracle_net.forward(x.astype(np.float32))
    q = lambda v: np.clip(np.floor(v * 255.0 + 0.5), 0, 255)
    try:
        for flags in (0, 8):  # conv_last with (dy, cout) in the MFMA's M dimension / through the generic 32-cout path
            sr.set_option("flow_flags", flags)
            got = sr.net_forward(x).astype(np.float32)
            d = np.abs(got - ref)
            print("flow_flags=%d max %.3e p99.9 %.3e mean %.3e" % (flags, d.max(), np.quantile(d, 0.999), d.mean()))
            assert d.max() <= 3.0e-3 and np.quantile(d, 0.999) <= 1.5e-3
            assert np.abs(q(got) - q(ref)).max() <= 1
    finally:
        sr.set_option("flow_flags", 0)


@pytest.mark.parametrize("w,h", [(28, 24), (64, 32), (45, 50)])
def test_kernel_paths_agree(sr, w, h):
    """The wave layouts / epilogue forms of conv3x3_flow (8x32 / 4x64 MFMA waves, deferred / inline epilogue, weights resident /
    streamed, dead rows skipped / computed, few / many workgroups) are restatements of the same arithmetic: identical accumulation order per output
    value, so the whole network must agree BIT FOR BIT between them."""
    img = synth.make_image(17, w, h)
    x = (img.astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)).astype(np.float16)
    try:
        ref = sr.net_forward(x)
        assert np.isfinite(ref.astype(np.float32)).all()
        for dbg, flags, ncu in [(0, 1, 256), (0, 2, 256), (0, 3, 256), (0, 4, 256), (32, 0, 256), (0, 0, 16), (32, 3, 16), (0, 7, 16)]:
            sr.set_option("dbg", dbg)
            sr.set_option("flow_flags", flags)
            sr.set_option("num_cu", ncu)
            got = sr.net_forward(x)
            assert np.array_equal(got.view(np.uint16), ref.view(np.uint16)), (dbg, flags, ncu, np.abs(got.astype(np.float32) - ref.astype(np.float32)).max())
    finally:
        sr.set_option("dbg", 0)
        sr.set_option("flow_flags", 0)
        sr.set_option("num_cu", 256)


def test_oversized_tile_is_refused(sr):
    """32-bit plane offsets: a padded tile above 2^21 pixels is an argument error, not a wrong image."""
    sr.tilesize = 2000
    try:
        with pytest.raises(R.RealSRError) as e:
            sr.process(np.zeros((1500, 1500, 3), np.uint8))
        assert e.value.code == R.RSR_E_ARG
    finally:
        sr.tilesize = 200


# ---- end to end through rsr_process ---------------------------------------------------------------------
E2E = [(50, 43, 3, 32, False), (70, 20, 3, 64, False), (24, 20, 3, 200, False), (5, 3, 3, 32, False),
       (33, 64, 3, 32, False), (37, 20, 4, 32, False), (40, 33, 3, 32, True), (21, 38, 4, 16, True)]


@pytest.mark.parametrize("w,h,c,T,tta", E2E)
def test_process_matches_oracle_within_one(sr, sr_tta, oracle_net, w, h, c, T, tta):
    eng = sr_tta if tta else sr
    eng.tilesize = T
    img = synth.make_image(9 + w, w, h, c)
    ref = oracle_net.process(img, T, tta=tta)
    got = eng.process(img)
    assert got.shape == ref.shape == (4 * h, 4 * w, c)
    d = np.abs(got.astype(int) - ref.astype(int))
    assert d.max() <= 1, "max diff %d at %s" % (d.max(), np.argwhere(d > 1)[:4])
    assert (d > 0).mean() < 0.15  # +-1 is the bar; ~3-6 % of the bytes sit on a rounding boundary


def test_process_matches_oracle_on_seeded_random_geometries(sr, sr_tta, oracle_net):
    """A seeded sweep over image sizes, channel counts, tile sizes that divide nothing and TTA: ragged last tiles in both
    directions, single-row / single-column images, tiles larger than the image -- same +-1 bar as above."""
    rng = np.random.default_rng(20240923)
    cases = [(1, 1, 3, 32, False), (1, 47, 3, 33, False), (61, 1, 4, 35, True)]
    for _ in range(9):
        tta = bool(rng.integers(0, 4) == 0)
        lim = 40 if tta else 90
        cases.append((int(rng.integers(2, lim)), int(rng.integers(2, lim)), int(rng.choice([3, 4])), int(rng.integers(32, 71)), tta))
    for w, h, c, T, tta in cases:
        eng = sr_tta if tta else sr
        eng.tilesize = T
        img = synth.make_image(1000 + w * 97 + h, w, h, c)
        ref = oracle_net.process(img, T, tta=tta)
        got = eng.process(img)
        assert got.shape == ref.shape == (4 * h, 4 * w, c)
        d = np.abs(got.astype(int) - ref.astype(int))
        assert d.max() <= 1, ((w, h, c, T, tta), int(d.max()), np.argwhere(d > 1)[:4])


def test_golden_fixtures(sr, sr_tta):
    g = np.load(os.path.join(HERE, "golden", "cases.npz"))
    names = sorted(k[:-4] for k in g.files if k.endswith("_cfg"))
    assert len(names) >= 6
    for n in names:
        seed, w, h, c, T, tta = [int(v) for v in g[n + "_cfg"]]
        eng = sr_tta if tta else sr
        eng.tilesize = T
        got = eng.process(g[n + "_in"])
        d = np.abs(got.astype(int) - g[n + "_out"].astype(int))
        assert d.max() <= 1, n


def test_device_api_equals_host_api(sr):
    import torch
    sr.tilesize = 32
    img = synth.make_image(77, 61, 47)
    host = sr.process(img)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.zeros((47 * 4, 61 * 4, 3), dtype=torch.uint8, device="cuda")
    sr.process_device(d_in.data_ptr(), 61, 47, 3, d_out.data_ptr())
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy() == host).all()
    # On a caller's (non-default) stream, asynchronously.  The first call finds the context idle and runs ON that stream (no hop through
    # the context's compute stream: stat device_direct); the second, issued right behind it, finds work pending -- the compute stream
    # waits for the first -- and takes the ordered path through the compute stream.  Same bytes both ways; work the caller enqueues on
    # its stream afterwards (the copy below) sees the finished image.
    st = torch.cuda.Stream()
    outs = [torch.zeros_like(d_out) for _ in range(3)]
    n0 = sr.get_stat("device_direct")
    with torch.cuda.stream(st):
        for o in outs:
            sr.process_device(d_in.data_ptr(), 61, 47, 3, o.data_ptr(), stream=st.cuda_stream)
        copies = [o.clone() for o in outs]
    st.synchronize()
    assert sr.get_stat("device_direct") - n0 >= 1
    for o in copies:
        assert (o.cpu().numpy() == host).all()
    torch.cuda.synchronize()
    sr.process_device(d_in.data_ptr(), 61, 47, 3, outs[0].data_ptr(), stream=st.cuda_stream)  # idle again: direct
    st.synchronize()
    assert sr.get_stat("device_direct") - n0 >= 2 and (outs[0].cpu().numpy() == host).all()


def test_packed_blob_load_equals_file_load(paths, sr):
    import torch
    blob = R.model_pack(*paths)
    img = synth.make_image(78, 40, 30)
    sr.tilesize = 32
    want = sr.process(img)
    s2 = R.RealSR(0)
    s2.load_packed(blob)
    s2.tilesize = 32
    assert (s2.process(img) == want).all()
    s3 = R.RealSR(0)
    dblob = torch.from_numpy(blob).cuda()
    s3.load_packed(dblob.numel(), device_ptr=dblob.data_ptr())
    s3.tilesize = 32
    assert (s3.process(img) == want).all()
    s2.close()
    s3.close()


def test_small_workspace_batches_give_identical_output(paths):
    """Tile batches are an implementation detail: forcing 1-2 tiles per batch must not change a byte."""
    img = synth.make_image(79, 90, 70)
    a = R.RealSR(0)
    a.load(*paths)
    a.tilesize = 32
    want = a.process(img)
    a.set_option("max_workspace_mb", 40)
    got = a.process(img)
    a.close()
    assert (got == want).all()


def test_process_before_load_is_an_error():
    s = R.RealSR(0)
    with pytest.raises(R.RealSRError) as e:
        s.process(np.zeros((8, 8, 3), np.uint8))
    assert e.value.code == R.RSR_E_STATE
    s.close()


# ---- full BASELINE size: properties that need no oracle over the whole frame ---------------------------------
def test_full_hd_properties(sr, oracle_net):
    """C2 (1920x1080, T=200): determinism, tile locality, and one full-size tile against the oracle."""
    sr.tilesize = 200
    img = synth.make_image(3, 1920, 1080)
    a = sr.process(img)
    b = sr.process(img)
    assert a.shape == (4320, 7680, 3)
    assert (a == b).all()
    # locality: a crop cut on tile boundaries reproduces the interior tiles bit for bit
    sub = np.ascontiguousarray(img[200:800, 400:1000])
    s = sr.process(sub)
    assert (s[800:1600, 800:1600] == a[1600:2400, 2400:3200]).all()
    # one full-size interior tile (padded 220x220) against the oracle network
    P = 10
    tile = img[400 - P:600 + P, 600 - P:800 + P].astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
    ref = oracle_net.forward(tile)[:, 40:-40, 40:-40]
    ref8 = np.clip((ref * 255.0 + 0.5).astype(np.int32), 0, 255).transpose(1, 2, 0)
    d = np.abs(a[1600:2400, 2400:3200].astype(int) - ref8)
    assert d.max() <= 1


def test_baseline_config_c1_against_oracle(oracle_net):
    """BASELINE config C1 at full size: 256x256 RGB, tile 128 (2x2 tiles of 148x148 padded), the JPEG-model weights
    (seed 43).  The whole frame against the CPU restatement: +-1 uint8 everywhere."""
    import tempfile
    d = synth.make_model_dir(tempfile.mkdtemp(prefix="rsr_c1_"), "models-DF2K_JPEG", 43)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    net = oracle.OracleNet(pp, bp)
    img = synth.make_image(1235, 256, 256)
    ref = net.process(img, 128)
    s = R.RealSR(0)
    s.load(pp, bp)
    s.tilesize = 128
    got = s.process(img)
    s.close()
    assert got.shape == ref.shape == (1024, 1024, 3)
    dd = np.abs(got.astype(int) - ref.astype(int))
    assert dd.max() <= 1
    assert (dd > 0).mean() < 0.15
