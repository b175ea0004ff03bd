"""GPU tests of the PRECISE residual stream (rsr_set_option "precise" = 1; conv_flow.hip EPI 4 / 5 / 6 / 7).

The reference's Vulkan path stores every feature map in fp16 (realsr.cpp:44-46); the parity bar is its fp32 CPU path
(realsr.cpp:525-838).  In precise mode the 64-channel trunk is kept as hi + lo / 2048 (an fp16 + one bf8 byte per element) and
conv_last's fp32 result goes to the uint8 conversion unrounded: profiles/r06_storage_emulation.txt (CPU emulation) says this
halves the pre-quantise error, and these tests hold the engine to it -- layer by layer against numpy, the network against the
fp32 oracle with the tolerance SURVEY.md 8(c) set (max <= 2e-3, p99.9 <= 5e-4), end to end +-1 uint8 incl. the wide-swing
channel-spread stand-in that the fp16-storage default misses by one byte in three million."""
import os
import sys

import numpy as np
import pytest

import oracle
import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

pytestmark = pytest.mark.gpu

LO = np.float32(2048.0)


def bf8(b):
    """bf8 / e5m2 bytes -> float32: the upper byte of an IEEE fp16."""
    return (b.astype(np.uint16) << 8).view(np.float16).astype(np.float32)


def rand_hi_lo(rng, shape):
    """A random trunk tensor as the engine stores it: hi = fp16, lo = a residue byte that is consistent with hi (|lo / 2048| at most half
    an ulp of hi: exponent(lo) <= exponent(hi) + 0; a random sign and 2-bit mantissa)."""
    hi = rng.standard_normal(shape).astype(np.float16)
    e_hi = (hi.view(np.uint16) >> 10) & 31                      # biased exponent of hi
    e_lo = np.clip(e_hi.astype(np.int32) - rng.integers(1, 6, shape), 1, 30)  # |residue| * 2048 <= ulp(hi) / 2 * 2048 = 2^(e_hi - 15): exponent field below e_hi
    lo = ((rng.integers(0, 2, shape) << 7) | (e_lo << 2) | rng.integers(0, 4, shape)).astype(np.uint8)
    return hi, lo


def join(hi, lo):
    return hi.astype(np.float32) + bf8(lo) / LO


@pytest.fixture(scope="module")
def paths(model_dir):
    return os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin")


@pytest.fixture(scope="module")
def srp(paths):
    s = R.RealSR(0)
    s.load(*paths)
    s.set_option("precise", 1)
    yield s
    s.close()


@pytest.mark.parametrize("cin,h,w", [(192, 20, 40), (64, 33, 50), (192, 70, 90), (3, 9, 70)])
def test_precise_residual_epilogues_match_numpy(srp, cin, h, w):
    """EPI 4 / 5 on one convolution: v = s1*(conv + b) + (x_hi + x_lo/2048) [, v = s2*v + (r_hi + r_lo/2048)] in fp32, ONE rounding:
    hi = fp16(v), lo = bf8((v - hi)*2048).  Against numpy on the same operands only the accumulation order of the conv differs: hi must be
    the fp16 rounding of the reference (up to that noise at a rounding boundary), and hi + lo/2048 must agree with it to 2^-14 relative
    (the residue, <= 2^-11 |v|, kept to the 2 mantissa bits of a bf8) -- 2^-9 (two fp16 roundings) is what the default epilogue is
    allowed (test_residual_epilogues_match_numpy).  num_cu = 8: every workgroup walks several blocks (prefetch registers, ring wrap)."""
    rng = np.random.default_rng(cin + h)
    cout = 64
    x_hi, x_lo = rand_hi_lo(rng, (max(cin, cout), h, w))
    x = x_hi[:cin]
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    r_hi, r_lo = rand_hi_lo(rng, (cout, h, w))
    conv = oracle.conv3x3(x.astype(np.float32), wt, b, 0, 0.2)
    xr = join(x_hi[:cout], x_lo[:cout])
    forms = {"trunk": (1.0, False, None, r_hi, r_lo, 1.0, conv + join(r_hi, r_lo)),
             "trunk, no lo": (1.0, False, None, r_hi, None, 1.0, conv + r_hi.astype(np.float32)),
             "conv_first": (1.0, False, None, None, None, 1.0, conv)}
    if cin >= cout:
        forms.update({"conv5": (0.2, True, x_lo[:cout], None, None, 1.0, 0.2 * conv + xr),
                      "conv5, no lo": (0.2, True, None, None, None, 1.0, 0.2 * conv + x_hi[:cout].astype(np.float32)),
                      "conv5+rrdb": (0.2, True, x_lo[:cout], r_hi, r_lo, 0.2, 0.2 * (0.2 * conv + xr) + join(r_hi, r_lo))})
    try:
        for flags, dbg, ncu in ((0, 0, 256), (0, 0, 8), (0, 32, 8), (4, 0, 8), (1, 0, 256)):
            for k, v in (("flow_flags", flags), ("dbg", dbg), ("num_cu", ncu)):
                srp.set_option(k, v)
            for name, (s1, own, xl, rh, rl, s2, ref) in forms.items():
                if name == "conv_first" and cin != 3:
                    continue
                hi, lo = srp.conv3x3_res_precise(x, wt, b, s1, own_input_residual=own, x_lo=xl, res=rh, res_lo=rl, s2=s2)
                got = join(hi, lo)
                tol = np.abs(ref) * 2.0 ** -14 + 3e-5  # the bf8 residue + the conv's summation order (|conv| ~ 1, 1728 terms)
                assert (np.abs(got - ref) <= tol).all(), (name, flags, dbg, ncu, np.abs(got - ref).max())
                assert (np.abs(hi.astype(np.float32) - ref) <= np.abs(ref) * 2.0 ** -11 + 3e-5).all(), name  # one fp16 rounding, not two
                assert np.abs(got - ref).mean() < 0.2 * np.abs(hi.astype(np.float32) - ref).mean(), name  # and lo does carry the residue
                # hi is the fp16 rounding of the value whose residue lo carries: the residue is at most half an ulp of hi
                assert (np.abs(got - hi.astype(np.float32)) <= np.abs(got) * 2.0 ** -11 + 1e-7).all(), name
                hi2, none = srp.conv3x3_res_precise(x, wt, b, s1, own_input_residual=own, x_lo=xl, res=rh, res_lo=rl, s2=s2, want_lo=False)
                assert none is None and np.array_equal(hi2.view(np.uint16), hi.view(np.uint16)), name
    finally:
        for k, v in (("flow_flags", 0), ("dbg", 0), ("num_cu", 256)):
            srp.set_option(k, v)


def test_precise_network_prequantise_error(srp, oracle_net, weights):
    """The network output before quantisation, precise mode vs the fp32 oracle: max <= 2e-3, p99.9 <= 5e-4 in [0,1] units -- the
    target SURVEY.md 8(c) set and fp16 storage misses (9e-4 .. 1.1e-3 p99.9).  Also: the engine agrees with the PyTorch-CPU emulation of
    ITS storage (tests/torch_ref.py trunk='split' = fp16 + a bf8 residue, fea16=False, out32) as closely as that agrees with the oracle."""
    import torch_ref
    img = synth.make_image(5, 44, 36)
    x = (img.astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)).astype(np.float16)
    ref = oracle_net.forward(x.astype(np.float32))
    emu = torch_ref.net_forward_storage_np(weights, x.astype(np.float32), trunk="split", fea16=False, out32=True)
    q = lambda v: np.clip(np.floor(v * 255.0 + 0.5), 0, 255)  # noqa: E731
    try:
        outs = {}
        for flags in (0, 8):  # conv_last with (dy, cout) in M (EPI 6) / through the generic path (EPI 7)
            srp.set_option("flow_flags", flags)
            got = srp.net_forward_f32(x)
            d = np.abs(got - ref)
            de = np.abs(got - emu)
            print("precise, flow_flags=%d: vs oracle max %.3e p99.9 %.3e mean %.3e | vs emulation max %.3e mean %.3e | emulation vs oracle mean %.3e" % (
                flags, d.max(), np.quantile(d, 0.999), d.mean(), de.max(), de.mean(), np.abs(emu - ref).mean()))
            assert d.max() <= 2e-3 and np.quantile(d, 0.999) <= 5e-4
            assert np.abs(q(got) - q(ref)).max() <= 1
            assert de.mean() <= 1.2 * np.abs(emu - ref).mean()
            outs[flags] = got
            # the fp16 view of the same blob (rsr_net_forward in precise mode) is its rounding
            assert np.array_equal(srp.net_forward(x).view(np.uint16), got.astype(np.float16).view(np.uint16))
    finally:
        srp.set_option("flow_flags", 0)


def test_precise_kernel_paths_agree(srp):
    """Layouts / schedules that must not change a bit in precise mode either (weights resident / streamed, dead rows skipped / computed,
    few / many workgroups, the 4 x 64 layout for the plain 64-channel convs -- the precise residual convs always run 8 x 32)."""
    img = synth.make_image(17, 45, 50)
    x = (img.astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)).astype(np.float16)
    try:
        ref = srp.net_forward_f32(x)
        assert np.isfinite(ref).all()
        for dbg, flags, ncu in [(0, 1, 256), (0, 2, 256), (0, 4, 256), (32, 0, 256), (0, 0, 16), (32, 3, 16), (0, 7, 16)]:
            for k, v in (("dbg", dbg), ("flow_flags", flags), ("num_cu", ncu)):
                srp.set_option(k, v)
            got = srp.net_forward_f32(x)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (dbg, flags, ncu, np.abs(got - ref).max())
    finally:
        for k, v in (("dbg", 0), ("flow_flags", 0), ("num_cu", 256)):
            srp.set_option(k, v)


@pytest.mark.parametrize("w,h,c,T,tta", [(50, 43, 3, 32, 0), (50, 43, 4, 32, 0), (70, 40, 3, 32, 1), (130, 90, 3, 64, 0), (33, 64, 4, 200, 1)])
def test_precise_process_matches_oracle_within_one(paths, oracle_net, w, h, c, T, tta):
    """End to end in precise mode: RGB (conv_last writes the image itself, EPI 6), RGBA and TTA (planar fp32 blob -> postproc_tiles<float>),
    +-1 uint8 against the oracle; fewer bytes differ than with fp16 storage; the fused and the two-kernel RGB paths agree byte for byte."""
    img = synth.make_image(3, w, h, c)
    ref = oracle_net.process(img, T, tta=bool(tta))
    s = R.RealSR(0, tta_mode=bool(tta))
    try:
        s.load(*paths)
        s.tilesize = T
        base = s.process(img)
        s.set_option("precise", 1)
        got = s.process(img)
        d = np.abs(got.astype(int) - ref.astype(int))
        d0 = np.abs(base.astype(int) - ref.astype(int))
        print("%dx%dx%d T=%d tta=%d: precise != on %.2f %% (max %d), fp16 storage != on %.2f %% (max %d)" % (w, h, c, T, tta, 100 * (d > 0).mean(), d.max(),
                                                                                                       100 * (d0 > 0).mean(), d0.max()))
        assert d.max() <= 1 and d0.max() <= 1
        assert (d[..., :3] > 0).mean() < (d0[..., :3] > 0).mean()
        if c == 3 and not tta:
            s.set_option("dbg", 8192)  # conv_last never writes the image itself: planar fp32 + postproc
            assert np.array_equal(s.process(img), got)
            s.set_option("dbg", 0)
        s.set_option("precise", 0)
        assert np.array_equal(s.process(img), base)  # and back: the workspace layouts do not leak into each other
    finally:
        s.close()


def test_precise_mode_holds_the_bar_on_the_wide_swing_model(tmp_path_factory):
    """The stand-in fp16 storage misses (profiles/r05_fp16_storage.txt): log-normal per-channel gains, output swing -0.56 .. 1.79
    (synth.make_weights(45, chan_sigma=1, last_gain=0.2)).  With fp16 storage -- the engine's default AND the emulation of the reference's
    GPU path -- the pre-quantise error reaches 4.3e-3 > 1/255 and one byte of C1 is off by 2; precise mode keeps C1 within +-1 with a
    headroom of > 1.8 on the tile (emulation: 2.3)."""
    d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "m45_wide", 45, chan_sigma=1.0, last_gain=0.2)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    net = oracle.OracleNet(pp, bp)
    img = synth.make_image(1234, 256, 256)
    big = np.pad(img, ((10, 10), (10, 10), (0, 0)), mode="reflect")
    t = np.ascontiguousarray(big[:148, :148, :3].astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0))
    ref = net.forward(t)
    s = R.RealSR(0)
    try:
        s.load(pp, bp)
        e16 = np.abs(s.net_forward(t.astype(np.float16)).astype(np.float32) - ref).max()
        s.set_option("precise", 1)
        e32 = np.abs(s.net_forward_f32(t.astype(np.float16)) - ref).max()
        print("wide-swing model, 148x148 tile: max pre-quantise error fp16 storage %.3e (headroom %.2f), precise %.3e (headroom %.2f)" % (
            e16, (1 / 255) / e16, e32, (1 / 255) / e32))
        assert e32 <= (1 / 255) / 1.8 and e32 < 0.6 * e16
        s.tilesize = 128
        got = s.process(img)
        refimg = net.process(img, 128)
        dd = np.abs(got.astype(int) - refimg.astype(int))
        print("C1 on the wide-swing model, precise: max |d| = %d, %.2f %% of the bytes differ" % (dd.max(), 100 * (dd > 0).mean()))
        assert dd.max() <= 1
    finally:
        s.close()


@pytest.mark.parametrize("name", ["c2", "c3", "c5"])
def test_precise_baseline_frames_against_golden_samples(name):
    """The BASELINE frames C2 (1080p, tile 200), C3 (4K, tile 400) and C5 (models-DF2K_JPEG stand-in, 1080p, TTA x8) in PRECISE mode:
    all 60 tiles of each within +-1 of the committed strided oracle samples (tests/golden/frame_c*.npz, written from the fp32 oracle by
    tests/golden/make_frames.py -- the same files the default-mode tests of tests/test_gpu_round2.py use), and FEWER samples differ than
    with fp16 storage (5 - 6 % there).  C2 / C3: conv_last writes the image from its fp32 result (EPI 6); C5: eight planar fp32 blobs per
    tile merged by postproc_tiles_lds<float>."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_frames
    import oracle_pool
    mdir, wseed, iseed, w, h, T, tta = make_frames.FRAMES[name]
    d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), mdir, wseed)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    img = synth.make_image(iseed, w, h)
    s = R.RealSR(0, tta_mode=bool(tta))
    try:
        s.load(pp, bp)
        s.tilesize = T
        s.set_option("precise", 1)
        out = s.process(img)
    finally:
        s.close()
    n, frac = oracle_pool.check_frame_golden(out, name, img, bp, T)
    print("%s, precise mode: %d of 60 tiles within +-1 of the golden oracle samples, %.2f %% of the samples differ" % (name.upper(), n, 100 * frac))
    assert n == 60 and frac < 0.035


def test_precise_tile_ranges_and_group_equal_the_whole_image(paths):
    """Tile ranges (rsr_process_rows / rsr_process_group: what several GPUs do with one image) in precise mode: the pieces put together
    equal the whole-image call, RGB (fused conv_last, output rows relative to the range's first) and RGBA (planar fp32 blob)."""
    for c in (3, 4):
        img = synth.make_image(21, 150, 130, c)
        a, b = R.RealSR(0), R.RealSR(0)
        try:
            for s in (a, b):
                s.load(*paths)
                s.tilesize = 48
                s.set_option("precise", 1)
            whole = a.process(img)
            out = np.zeros_like(whole)
            a.process_rows(img, out, 0, 1)
            b.process_rows(img, out, 1, 3)
            assert np.array_equal(out, whole), c
            assert np.array_equal(R.process_group([a, b, a], img), whole), c
        finally:
            a.close()
            b.close()
