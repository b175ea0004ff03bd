"""CLI (realsr-ncnn-vulkan_amd/bin/realsr-hip): flag surface and validation of the reference tool
(main.cpp:101-115, :484-603), codecs (image_io.h), and -- on the GPU box -- an end-to-end directory run."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "realsr-ncnn-vulkan_amd", "bin", "realsr-hip")
CSRC = os.path.join(ROOT, "realsr-ncnn-vulkan_amd", "csrc")


def write_png(path, img, filt=0):
    """Independent PNG writer (numpy + zlib): 8-bit RGB/RGBA/gray, given filter type per row (0 or 2)."""
    h, w = img.shape[:2]
    c = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    rows = img.reshape(h, w * c).astype(np.uint8)
    if filt == 2:
        up = np.vstack([np.zeros((1, w * c), np.uint8), rows[:-1]])
        rows = (rows.astype(np.int16) - up).astype(np.uint8)
    raw = b"".join(bytes([filt]) + r.tobytes() for r in rows)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png(path):
    """Independent PNG reader for what save_png emits (8-bit RGB/RGBA, any filter)."""
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat = 8, b""
    while pos < len(d):
        n, t = struct.unpack(">I4s", d[pos:pos + 8])
        body = d[pos + 8:pos + 8 + n]
        if t == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
        elif t == b"IDAT":
            idat += body
        pos += 12 + n
    c = {2: 3, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * c + 1)
    out = np.zeros((h, w * c), np.uint8)
    for y in range(h):
        ft, row = raw[y, 0], raw[y, 1:].astype(np.int32)
        if ft == 0:
            out[y] = row
        elif ft == 1:
            for i in range(c):
                out[y, i::c] = np.cumsum(row[i::c]) & 255
        elif ft == 2:
            out[y] = (row + (out[y - 1] if y else 0)) & 255
        else:
            raise AssertionError("unexpected filter %d" % ft)
    return out.reshape(h, w, c)


@pytest.fixture(scope="module")
def io_harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("io")
    src = d / "t_io.cpp"
    src.write_text('#include "image_io.h"\nint main(int c, char** v){Image im; std::string e = imgio::load_image(v[1], im);'
                   'if(!e.empty()){fprintf(stderr,"%s\\n",e.c_str());return 1;} e = imgio::save_image(v[2], im);'
                   'if(!e.empty()){fprintf(stderr,"%s\\n",e.c_str());return 2;} printf("%d %d %d\\n",im.w,im.h,im.elempack);return 0;}\n')
    exe = d / "t_io"
    lib = os.path.join(ROOT, "realsr-ncnn-vulkan_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", CSRC, "-o", str(exe), str(src), "-L", lib, "-lrealsr_hip",
                           "-Wl,-rpath," + lib, "-lz", "-ldl"])
    return str(exe)


@pytest.mark.parametrize("shape,filt", [((13, 17, 3), 0), ((9, 20, 4), 2), ((11, 7), 0), ((6, 5, 2), 2)])
def test_png_codec_roundtrip(io_harness, tmp_path, shape, filt):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    write_png(tmp_path / "in.png", img, filt)
    out = subprocess.check_output([io_harness, str(tmp_path / "in.png"), str(tmp_path / "out.png")]).decode().split()
    got = read_png(tmp_path / "out.png")
    if img.ndim == 2:  # gray -> RGB (main.cpp:247-252)
        want = np.repeat(img[:, :, None], 3, 2)
    elif img.shape[2] == 2:  # gray+alpha -> RGBA (main.cpp:253-260)
        want = np.concatenate([np.repeat(img[:, :, :1], 3, 2), img[:, :, 1:]], 2)
    else:
        want = img
    assert [int(v) for v in out] == [want.shape[1], want.shape[0], want.shape[2]]
    assert (got == want).all()


def test_pnm_and_unsupported_formats(io_harness, tmp_path):
    img = np.arange(5 * 4 * 3, dtype=np.uint8).reshape(5, 4, 3)
    with open(tmp_path / "a.ppm", "wb") as f:
        f.write(b"P6\n# c\n4 5\n255\n" + img.tobytes())
    subprocess.check_call([io_harness, str(tmp_path / "a.ppm"), str(tmp_path / "o.png")], stdout=subprocess.DEVNULL)
    assert (read_png(tmp_path / "o.png") == img).all()
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 20)  # a truncated jpg: decode failure (main.cpp:292-299)
    r = subprocess.run([io_harness, str(tmp_path / "x.jpg"), str(tmp_path / "o2.png")], capture_output=True)
    assert r.returncode == 1
    (tmp_path / "x.webp").write_bytes(b"RIFF\x10\0\0\0WEBPVP8 " + b"\0" * 16)
    r = subprocess.run([io_harness, str(tmp_path / "x.webp"), str(tmp_path / "o3.png")], capture_output=True)
    assert r.returncode == 1 and b"webp" in r.stderr


def test_webp_decode_and_lossless_encode(io_harness, tmp_path):
    """webp in (lossy, lossless, with alpha) and lossless webp out go through libwebp like the reference's webp_image.h;
    Pillow (its own bundled libwebp) is the independent decoder / encoder on the other side."""
    Image = pytest.importorskip("PIL.Image")
    from PIL import features
    if not features.check("webp"):
        pytest.skip("Pillow without webp")
    import ctypes.util
    if not (ctypes.util.find_library("webp") or os.environ.get("RSR_LIBWEBP")):
        pytest.skip("no libwebp at run time")
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:37, 0:51]
    rgb = np.stack([(xx * 5) % 256, (yy * 7) % 256, (xx * yy) % 256], 2).astype(np.uint8)
    rgb[10:20, 10:30] = rng.integers(0, 256, (10, 20, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, ((xx + yy) * 3 % 256).astype(np.uint8)[:, :, None]], 2)
    cases = {"lossy": (rgb, dict(quality=80)), "lossless": (rgb, dict(lossless=True)),
             "lossy_alpha": (rgba, dict(quality=70)), "lossless_alpha": (rgba, dict(lossless=True, exact=True))}
    for name, (arr, kw) in cases.items():
        src = tmp_path / (name + ".webp")
        Image.fromarray(arr).save(src, **kw)
        want = np.asarray(Image.open(src))
        out = subprocess.check_output([io_harness, str(src), str(tmp_path / (name + ".png"))]).decode().split()
        assert [int(v) for v in out] == [want.shape[1], want.shape[0], want.shape[2]], name
        assert (read_png(tmp_path / (name + ".png")) == want).all(), name
        if "lossless" in name:
            assert (want == arr).all()
    for arr in (rgb, rgba):  # png -> lossless webp (WebPEncodeLosslessRGB / RGBA, webp_image.h:66-85)
        write_png(tmp_path / "e.png", arr)
        subprocess.check_call([io_harness, str(tmp_path / "e.png"), str(tmp_path / "e.webp")], stdout=subprocess.DEVNULL)
        assert (np.asarray(Image.open(tmp_path / "e.webp")) == arr).all()


def test_jpg_roundtrip_and_png_features(io_harness, tmp_path):
    """stb codecs as in the reference (main.cpp:208-267, 339-416): jpg encode at quality 100 and decode again; a PNG with a
    tRNS colour key gets its alpha channel (stb semantics); a 16-bit PNG is reduced to 8 bit."""
    yy, xx = np.mgrid[0:48, 0:64]
    img = np.stack([(xx * 4) % 256, (yy * 5) % 256, ((xx + yy) * 2) % 256], 2).astype(np.uint8)  # smooth: jpg keeps it close
    write_png(tmp_path / "g.png", img)
    subprocess.check_call([io_harness, str(tmp_path / "g.png"), str(tmp_path / "g.jpg")], stdout=subprocess.DEVNULL)
    assert open(tmp_path / "g.jpg", "rb").read(3) == b"\xff\xd8\xff"
    out = subprocess.check_output([io_harness, str(tmp_path / "g.jpg"), str(tmp_path / "g2.png")]).decode().split()
    assert [int(v) for v in out] == [64, 48, 3]
    back = read_png(tmp_path / "g2.png").astype(int)
    assert np.abs(back - img.astype(int)).mean() < 2.0
    # RGB png + tRNS colour key (0,0,0) -> RGBA with alpha 0 exactly on the keyed pixels
    key = img.copy()
    key[:4, :4] = 0
    h, w = key.shape[:2]
    raw = b"".join(b"\0" + r.tobytes() for r in key.reshape(h, w * 3))

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(tmp_path / "k.png", "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"tRNS", b"\0" * 6)
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    out = subprocess.check_output([io_harness, str(tmp_path / "k.png"), str(tmp_path / "k2.png")]).decode().split()
    assert int(out[2]) == 4
    k2 = read_png(tmp_path / "k2.png")
    assert (k2[:4, :4, 3] == 0).all() and (k2[8:, 8:, 3] == 255).all() and (k2[..., :3] == key).all()
    # 16-bit RGB -> 8 bit (high byte)
    raw16 = b"".join(b"\0" + np.stack([r, r], -1).astype(np.uint8).tobytes() for r in img.reshape(h, w * 3))
    with open(tmp_path / "s.png", "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw16, 6))
                + chunk(b"IEND", b""))
    subprocess.check_call([io_harness, str(tmp_path / "s.png"), str(tmp_path / "s2.png")], stdout=subprocess.DEVNULL)
    assert (read_png(tmp_path / "s2.png") == img).all()


def run_cli(*args):
    return subprocess.run([CLI] + list(args), capture_output=True, text=True)


def test_cli_flag_validation(tmp_path):
    assert os.path.exists(CLI), "run __graft_entry__.build() first"
    r = run_cli()
    assert r.returncode != 0 and "Usage:" in r.stderr and "-j load:proc:save" in r.stderr
    png = tmp_path / "a.png"
    write_png(png, np.zeros((4, 4, 3), np.uint8))
    out = str(tmp_path / "o.png")
    assert "invalid scale argument" in run_cli("-i", str(png), "-o", out, "-s", "2").stderr
    assert "invalid tilesize argument" in run_cli("-i", str(png), "-o", out, "-t", "16").stderr
    assert "invalid tilesize argument" in run_cli("-i", str(png), "-o", out, "-t", "64,64").stderr
    assert "invalid jobs_proc thread count argument" in run_cli("-i", str(png), "-o", out, "-g", "0,1", "-j", "1:2,2,2:2").stderr
    assert "invalid thread count argument" in run_cli("-i", str(png), "-o", out, "-j", "0:2:2").stderr
    assert "invalid outputpath extension type" in run_cli("-i", str(png), "-o", str(tmp_path / "o.bmp")).stderr
    assert "unknown model dir type" in run_cli("-i", str(png), "-o", out, "-m", "models-foo").stderr
    assert "either file or directory" in run_cli("-i", str(png), "-o", str(tmp_path)).stderr
    assert "no CPU fallback" in run_cli("-i", str(png), "-o", out, "-g", "-1").stderr
    # webp output needs libwebp at run time; pointing the binding at nothing must fail loudly, before any GPU work
    r = subprocess.run([CLI, "-i", str(png), "-o", str(tmp_path / "o.webp")], capture_output=True, text=True,
                       env=dict(os.environ, RSR_LIBWEBP="/nonexistent/libwebp.so"))
    assert r.returncode != 0 and "needs libwebp.so at run time" in r.stderr


@pytest.mark.gpu
def test_cli_directory_run_matches_library(tmp_path, model_dir):
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    outd.mkdir()
    imgs = {"a": synth.make_image(31, 40, 30), "b": synth.make_image(32, 33, 21, 4), "a_dup": synth.make_image(33, 24, 24)}
    write_png(ind / "a.png", imgs["a"])
    write_png(ind / "b.png", imgs["b"])
    with open(ind / "a.ppm", "wb") as f:  # same stem as a.png -> the collision rule renames its output (main.cpp:625-637)
        f.write(b"P6\n24 24\n255\n" + imgs["a_dup"].tobytes())
    r = run_cli("-i", str(ind), "-o", str(outd), "-m", model_dir, "-t", "32", "-j", "2:2:2", "-v")
    assert r.returncode == 0, r.stderr
    assert "both a.ppm and a.png output a.png" in r.stderr or "both a.png and a.ppm" in r.stderr
    assert sorted(os.listdir(outd)) == ["a.png", "a.ppm.png", "b.png"]
    # the reference's progress lines, one per tile: "%.2f%%" of (yi * xtiles + xi) / (ytiles * xtiles) (realsr.cpp:481) -- a.png is
    # 40x30 at tile 32 = 2 x 1 tiles -> 0.00% and 50.00%; b.png 33x21 = 2 x 1; a.ppm 24x24 = 1 tile
    lines = r.stderr.splitlines()
    assert lines.count("0.00%") == 3 and lines.count("50.00%") == 2, r.stderr
    sr = R.RealSR(0)
    sr.load(os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin"))
    sr.tilesize = 32
    names = sorted(os.listdir(ind))  # a.png, a.ppm, b.png: the first of a collision keeps the plain name
    first, second = names[0], names[1]
    key = {"a.png": "a", "a.ppm": "a_dup"}
    assert (read_png(outd / (first.split(".")[0] + ".png")) == sr.process(imgs[key[first]])).all()
    assert (read_png(outd / (second + ".png")) == sr.process(imgs[key[second]])).all()
    assert (read_png(outd / "b.png") == sr.process(imgs["b"])).all()
    sr.close()


@pytest.mark.gpu
def test_cli_many_small_images_and_precise_mode(tmp_path, model_dir):
    """The reference's small-image mode ("-j 4:4:4 for many small images", README.md:61): 24 same-size pngs through -j 2:8:2 -- eight proc
    threads call the one context concurrently, their images are merged into shared tile batches (engine.h) -- every output byte equal
    to a lone library call's; progress lines still one per tile and image (realsr.cpp:481).  The same directory with RSR_PRECISE=1 in the
    environment equals the library with rsr_set_option("precise", 1)."""
    ind, outd, outp = tmp_path / "in", tmp_path / "out", tmp_path / "outp"
    for d in (ind, outd, outp):
        d.mkdir()
    imgs = {"s%02d.png" % i: synth.make_image(500 + i, 48, 40) for i in range(24)}
    for name, im in imgs.items():
        write_png(ind / name, im)
    sr = R.RealSR(0)
    sr.load(os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin"))
    sr.tilesize = 32
    sr.set_option("merge", 1)
    want = {n: sr.process(im) for n, im in imgs.items()}
    sr.set_option("precise", 1)
    wantp = {n: sr.process(im) for n, im in imgs.items()}
    sr.close()
    r = run_cli("-i", str(ind), "-o", str(outd), "-m", model_dir, "-t", "32", "-j", "2:8:2")
    assert r.returncode == 0, r.stderr
    lines = r.stderr.splitlines()
    assert lines.count("0.00%") == 24 and lines.count("25.00%") == 24 and lines.count("75.00%") == 24, r.stderr[-2000:]  # 48x40 at tile 32 = 2 x 2 tiles
    for n in imgs:
        assert (read_png(outd / n) == want[n]).all(), n
    rp = subprocess.run([CLI, "-i", str(ind), "-o", str(outp), "-m", model_dir, "-t", "32", "-j", "2:8:2", "-v"], capture_output=True, text=True,
                        env=dict(os.environ, RSR_PRECISE="1"))
    assert rp.returncode == 0 and "precise residual trunk" in rp.stderr, rp.stderr[-2000:]
    differs = 0
    for n in imgs:
        got = read_png(outp / n)
        assert (got == wantp[n]).all(), n
        differs += int((got != want[n]).any())
    assert differs > 0  # (and precise mode is not the default mode by another name)


@pytest.mark.gpu
def test_cli_jpg_directory_and_threads(tmp_path, model_dir):
    """-j 2:4:2 over a directory of jpg + RGBA png inputs, jpg output: 4 proc threads share one context (main.cpp:811-828);
    the RGBA image cannot be a jpg and is written as <name>.jpg.png (main.cpp:278-288).  The png outputs equal the library
    byte for byte; the jpg outputs decode to within the codec's error of it."""
    ind, outd, tmp = tmp_path / "in", tmp_path / "out", tmp_path / "tmp"
    for d in (ind, outd, tmp):
        d.mkdir()
    io_src = tmp / "t_io.cpp"
    io_src.write_text('#include "image_io.h"\nint main(int c, char** v){Image im; std::string e = imgio::load_image(v[1], im);'
                      'if(!e.empty()) return 1; e = imgio::save_image(v[2], im); return e.empty() ? 0 : 2;}\n')
    lib = os.path.join(ROOT, "realsr-ncnn-vulkan_amd", "lib")
    conv = str(tmp / "t_io")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", CSRC, "-o", conv, str(io_src), "-L", lib, "-lrealsr_hip", "-Wl,-rpath," + lib, "-lz", "-ldl"])
    sr = R.RealSR(0)
    sr.load(os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin"))
    sr.tilesize = 32
    want = {}
    for i, (w, h) in enumerate([(40, 30), (33, 21), (52, 47), (24, 64), (45, 45), (70, 20)]):
        img = synth.make_image(60 + i, w, h)
        write_png(tmp / "p.png", img)
        subprocess.check_call([conv, str(tmp / "p.png"), str(ind / ("j%d.jpg" % i))])
        subprocess.check_call([conv, str(ind / ("j%d.jpg" % i)), str(tmp / "back.png")])  # what the CLI will decode
        want["j%d.jpg" % i] = sr.process(read_png(tmp / "back.png"))
    rgba = synth.make_image(70, 37, 29, 4)
    write_png(ind / "r.png", rgba)
    want["r.jpg.png"] = sr.process(rgba)
    sr.close()
    r = run_cli("-i", str(ind), "-o", str(outd), "-m", model_dir, "-t", "32", "-j", "2:4:2", "-f", "jpg", "-v")
    assert r.returncode == 0, r.stderr
    assert "r.png has alpha channel" in r.stderr
    assert sorted(os.listdir(outd)) == sorted(want)
    assert (read_png(outd / "r.jpg.png") == want["r.jpg.png"]).all()
    for name in want:
        if name.endswith(".jpg"):
            subprocess.check_call([conv, str(outd / name), str(tmp / "o.png")])
            d = np.abs(read_png(tmp / "o.png").astype(int) - want[name].astype(int))
            assert d.mean() < 2.5, (name, d.mean())


@pytest.mark.gpu
def test_cli_webp_in_and_out(tmp_path, model_dir):
    """webp inputs (lossy RGB, lossless RGBA) and -f webp output (lossless, webp_image.h:66-85): the CLI's output decodes
    -- with Pillow's own libwebp -- to exactly what the library returns for the pixels Pillow decodes from the inputs."""
    Image = pytest.importorskip("PIL.Image")
    from PIL import features
    if not features.check("webp"):
        pytest.skip("Pillow without webp")
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    outd.mkdir()
    Image.fromarray(synth.make_image(81, 44, 31)).save(ind / "a.webp", quality=85)
    Image.fromarray(synth.make_image(82, 29, 40, 4)).save(ind / "b.webp", lossless=True, exact=True)
    sr = R.RealSR(0)
    sr.load(os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin"))
    sr.tilesize = 32
    want = {n: sr.process(np.asarray(Image.open(ind / (n + ".webp")))) for n in ("a", "b")}
    sr.close()
    r = run_cli("-i", str(ind), "-o", str(outd), "-m", model_dir, "-t", "32", "-f", "webp")
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(outd)) == ["a.webp", "b.webp"]
    for n in ("a", "b"):
        got = np.asarray(Image.open(outd / (n + ".webp")))
        assert got.shape == want[n].shape, n
        if got.shape[2] == 4:  # WebPEncodeLosslessRGBA is not "exact": colour under alpha == 0 is the encoder's to choose
            vis = want[n][:, :, 3] > 0
            assert (got[:, :, 3] == want[n][:, :, 3]).all() and (got[vis] == want[n][vis]).all(), n
        else:
            assert (got == want[n]).all(), n
