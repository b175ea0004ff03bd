"""CPU tests of the host side of librealsr_hip.so: it loads, exports every symbol include/realsr_hip.h
declares, parses/validates/packs the model; nothing here launches a kernel."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import realsr_ncnn_vulkan_amd as R
from realsr_ncnn_vulkan_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "realsr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsr_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    L = R.lib()
    names = header_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), "librealsr_hip.so does not export %s" % n
    assert sorted(R.EXPORTS) == names
    assert b"gfx950" in L.rsr_version()


def test_library_does_not_link_the_oracle():
    import subprocess
    out = subprocess.check_output(["ldd", R.LIB_PATH]).decode()
    assert "oracle" not in out
    syms = subprocess.check_output(["nm", "-D", R.LIB_PATH]).decode()
    assert "orc_" not in syms


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(R.RealSRError) as e:
        R.RealSR(0)
    assert e.value.code == R.RSR_E_DEVICE
    assert "no CPU fallback" in str(e.value) or "device" in str(e.value)


def test_model_info_both_encodings(model_dir, tmp_path, weights):
    info = R.model_info(os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin"))
    assert info == dict(n_layers=999, n_convs=351, n_weights=16684416, n_biases=13571, bin_encoding=1)
    d = tmp_path / "models-DF2K_JPEG"
    d.mkdir()
    synth.write_param(str(d / "x4.param"))
    synth.write_bin(str(d / "x4.bin"), weights, "fp32")
    assert R.model_info(str(d / "x4.param"), str(d / "x4.bin"))["bin_encoding"] == 0


def test_error_codes(model_dir, tmp_path):
    pp, bp = os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin")
    with pytest.raises(R.RealSRError) as e:
        R.model_info(str(tmp_path / "missing.param"), bp)
    assert e.value.code == R.RSR_E_IO
    bad = tmp_path / "bad.param"
    bad.write_text(open(pp).read().replace("7767517", "7767518", 1))
    with pytest.raises(R.RealSRError) as e:
        R.model_info(str(bad), bp)
    assert e.value.code == R.RSR_E_FORMAT
    # a graph that parses but is not the RRDBNet the fused schedule implements
    txt = open(pp).read()
    bad.write_text(txt.replace("-23310=1,2.000000e-01", "-23310=1,1.000000e-01", 1))
    with pytest.raises(R.RealSRError) as e:
        R.model_info(str(bad), bp)
    assert e.value.code == R.RSR_E_GRAPH
    lines = txt.split("\n")
    # swap the operands of the first Eltwise: 0.2*x + 1.0*x5 is a different network
    i = next(k for k, l in enumerate(lines) if l.startswith("Eltwise"))
    f = lines[i].split()
    f[4], f[5] = f[5], f[4]
    bad.write_text("\n".join(lines[:i] + [" ".join(f)] + lines[i + 1:]))
    with pytest.raises(R.RealSRError) as e:
        R.model_info(str(bad), bp)
    assert e.value.code == R.RSR_E_GRAPH
    tb = tmp_path / "trunc.bin"
    tb.write_bytes(open(bp, "rb").read()[:-100])
    with pytest.raises(R.RealSRError) as e:
        R.model_info(pp, str(tb))
    assert e.value.code == R.RSR_E_IO


def test_packed_blob_layout(model_dir, weights):
    """Un-swizzle the packed LDS images (16-channel-plane images of conv3x3_flow) and compare with the OIHW weights; the blob
    is what rsr_load uploads and what the multi-GPU broadcast carries."""
    pp, bp = os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin")
    bl = R.model_pack(pp, bp)
    assert 33e6 < bl.size < 34.5e6  # 33.5 MB: what one RCCL broadcast moves
    rec = np.dtype([("cin", "<u4"), ("cout", "<u4"), ("act", "<u4"), ("nplanes", "<u4"), ("nt", "<u4"),
                    ("slope", "<f4"), ("b_off", "<u8"), ("w16_off", "<u8"), ("aux_off", "<u8")])
    assert rec.itemsize == 48
    specs = synth.conv_specs()
    magic, version, nconv, flags = np.frombuffer(bl[:16], np.uint32)
    assert magic == 0x50525352 and version == 6 and nconv == 351 and flags == 0
    assert int(np.frombuffer(bl[16:24], np.uint64)[0]) == bl.size
    table = np.frombuffer(bl[24:24 + 351 * 48], rec)
    for i in (0, 1, 4, 5, 346, 349, 350):
        t = table[i]
        cin, cout, act = specs[i]
        assert (t["cin"], t["cout"], t["act"]) == (cin, cout, act)
        np_, nt = (cin + 31) // 32, (cout + 31) // 32
        assert (t["nplanes"], t["nt"]) == (np_, nt)
        rows = 9 * nt * 32
        W, b = weights[i]
        Wp = np.zeros((nt * 32, np_ * 32, 3, 3), np.float32)
        Wp[:cout, :cin] = W
        # 16-channel-plane images: [plane][tap][row][2 slots of 8], slots swapped when (row >> 3) & 1; row n of a 32-row tile carries
        # output channel row_cout(n) (model.h: the order in which the MFMA result registers are consecutive channels)
        def row_cout(n):
            i = n & 31
            return (n & ~31) + ((i >> 4) & 1) * 16 + ((i >> 2) & 1) * 8 + ((i >> 3) & 1) * 4 + (i & 3)
        assert sorted(row_cout(n) for n in range(64)) == list(range(64)) and [row_cout(n) for n in (0, 3, 4, 8, 12, 20, 24, 31)] == [0, 3, 8, 4, 12, 24, 20, 31]
        img16 = np.frombuffer(bl[int(t["w16_off"]):int(t["w16_off"]) + 2 * np_ * rows * 32], np.float16).reshape(2 * np_, rows, 2, 8)
        for pl in range(2 * np_):
            for row in (0, 5, 9, 21, 31, 40, rows - 1):
                tap, n = divmod(row, nt * 32)
                swz = (n >> 3) & 1
                for slot in range(2):
                    got = img16[pl, row, slot ^ swz].astype(np.float32)
                    want = Wp[row_cout(n), pl * 16 + slot * 8: pl * 16 + slot * 8 + 8, tap // 3, tap % 3]
                    assert (got == want).all(), (i, pl, row, slot)
        bias = np.frombuffer(bl[int(t["b_off"]):int(t["b_off"]) + nt * 32 * 4], np.float32)
        bp_ = np.zeros(nt * 32, np.float32)
        bp_[:cout] = b
        assert (bias == bp_[[row_cout(n) for n in range(nt * 32)]]).all()
        # conv_last's aux image, (dy, cout) in the MFMA's M dimension: [plane][dx][row = dy*8 + c][2 slots of 8], other rows zero
        if cout <= 4:
            aux = np.frombuffer(bl[int(t["aux_off"]):int(t["aux_off"]) + 2 * np_ * 3 * 32 * 32], np.float16).reshape(2 * np_, 3, 32, 2, 8)
            for pl in range(2 * np_):
                for dx in range(3):
                    for row in range(32):
                        dy, c = divmod(row, 8)
                        swz = (row >> 3) & 1
                        for slot in range(2):
                            got = aux[pl, dx, row, slot ^ swz].astype(np.float32)
                            want = Wp[c, pl * 16 + slot * 8: pl * 16 + slot * 8 + 8, dy, dx] if (dy < 3 and c < cout) else np.zeros(8, np.float32)
                            assert (got == want).all(), (i, pl, dx, row, slot)
        else:
            assert t["aux_off"] == 0


def test_tile_partition_balances_one_image_over_the_gpus():
    """rsr_process_group's split of ONE image: contiguous row-major tile ranges, every share non-empty, loads
    (padded pixels) within one tile of each other.  C2 = 1920x1080 at tile 200: 60 tiles of four shapes over 8 GPUs -- split by
    tile ROWS (6 of them, the last 80 px high) it would use 6 GPUs and wait for the slowest."""
    def areas(w, h, T, P):
        xt, yt = -(-w // T), -(-h // T)
        return [(min((t % xt + 1) * T, w) - (t % xt) * T + 2 * P) * (min((t // xt + 1) * T, h) - (t // xt) * T + 2 * P) for t in range(xt * yt)]
    for (w, h, T, parts) in [(1920, 1080, 200, 8), (1920, 1080, 200, 2), (1920, 1080, 200, 4), (3840, 2160, 400, 8), (70, 150, 32, 5), (70, 150, 32, 20), (50, 40, 64, 3)]:
        a = areas(w, h, T, 10)
        b = R.tile_partition(w, h, T, 10, parts)
        used = len(b) - 1
        assert used == min(parts, len(a)) and b[0] == 0 and b[-1] == len(a)
        assert all(b[i] < b[i + 1] for i in range(used)), b
        loads = [sum(a[b[i]:b[i + 1]]) for i in range(used)]
        assert max(loads) - min(loads) <= 2 * max(a), (w, h, T, parts, loads)      # within a tile of the ideal share on either side
        assert max(loads) <= sum(a) / used + max(a), (w, h, T, parts, loads)
    b = R.tile_partition(1920, 1080, 200, 10, 8)  # 6-7 full tiles per share; the last share takes the 10 low tiles of the last row
    assert b == [0, 7, 14, 20, 27, 34, 41, 47, 60]
    with pytest.raises(R.RealSRError):
        R.tile_partition(0, 10, 32, 10, 2)


def test_rccl_probe_resolves_the_collective_without_a_gpu():
    """rsr_rccl_probe (host-only): the RCCL branch of rsr_create_group starts with dlopen(librccl) + six entry points
    (group.cpp).  The ROCm image carries librccl, so the probe must succeed in the GPU-less container too -- and creating the
    group itself must still fail loudly there (no device), whatever RSR_GROUP_FORCE_RCCL says."""
    assert R.rccl_probe() is None
    import torch
    if not torch.cuda.is_available():
        env_before = os.environ.get("RSR_GROUP_FORCE_RCCL")
        os.environ["RSR_GROUP_FORCE_RCCL"] = "1"
        try:
            d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
            with pytest.raises(R.RealSRError) as e:
                R.create_group([0], os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
            assert e.value.code == R.RSR_E_DEVICE
        finally:
            if env_before is None:
                del os.environ["RSR_GROUP_FORCE_RCCL"]
            else:
                os.environ["RSR_GROUP_FORCE_RCCL"] = env_before


def test_shard_frames_partitions_exactly():
    for n, ws in [(64, 8), (7, 4), (1, 2), (0, 3)]:
        seen = sorted(i for r in range(ws) for i in R.shard_frames(n, ws, r))
        assert seen == list(range(n))


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """bench.py --gpus 8 inside a 1-rank environment must not print a 1-GPU line labelled as anything else."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_real_model_harness_skips_without_a_blob_and_runs_host_checks(model_dir, tmp_path):
    """tools/check_real_model.py: with no x4.bin anywhere it reports SKIP and exits 0; on a model directory it identifies the
    encoding by size (33,424,520 B fp16-tagged / 66,793,352 B raw fp32, SURVEY a-7), packs, and walks the graph in fp32 for the
    fp16 activation-range guard -- the walk must agree with the oracle's own .param interpreter.  (GPU legs: -m gpu.)"""
    import json
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_real_model.py")
    env = dict(os.environ)
    env.pop("RSR_REAL_MODELS", None)
    r = subprocess.run([sys.executable, tool, str(tmp_path)], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("SKIP"), r.stdout + r.stderr
    out = tmp_path / "rep.json"
    r = subprocess.run([sys.executable, tool, model_dir, "--no-gpu", "--json", str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RESULT: ok" in r.stdout, r.stdout + r.stderr
    rep = json.loads(out.read_text())
    assert rep["encoding_by_size"] == rep["encoding_by_parser"] == "fp16-tagged" and rep["bin_bytes"] == 33424520
    assert rep["walk_vs_oracle_max"] < 1e-4 and len(rep["activation_peaks"]) == 1 + 69 + 4
    assert 1.0 < max(v for _, v in rep["activation_peaks"]) < 65504


def test_header_is_plain_c_and_a_c_host_links_against_the_abi(tmp_path):
    """The drop-in boundary is a C ABI (extern "C", plain pointers and sizes, no C++ / torch types in the signatures): include/realsr_hip.h
    must compile as C99 with -pedantic, and a host written in C must link against librealsr_hip.so and reach the host-only entry points
    without a GPU -- rsr_version, rsr_tile_partition (C2's 60 tiles over 8 shares), rsr_model_info on a generated model directory, and
    rsr_create failing LOUDLY with an error code and a message when there is no gfx950 device (no CPU fallback: realsr.cpp:147-151's
    -g -1 path exists only as the test oracle)."""
    import subprocess
    import sys
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "realsr_hip.h"
int main(void)
{
    int b[9], n, i, rc;
    rsr_ctx* ctx = NULL;
    printf("version %s\n", rsr_version());
    n = rsr_tile_partition(1920, 1080, 200, 10, 8, b);
    printf("shares %d:", n);
    for (i = 0; i <= n; i++) printf(" %d", b[i]);
    printf("\n");
    rc = rsr_create(&ctx, 0, 0, 1);
    printf("create rc %d ctx %s msg [%s]\n", rc, ctx ? "set" : "null", rc ? rsr_last_error(NULL) : "");
    if (rc == 0) rsr_destroy(ctx);
    return (n == 8 && b[0] == 0 && b[8] == 60) ? 0 : 1;
}
''')
    lib = os.path.join(ROOT, "realsr-ncnn-vulkan_amd", "lib")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, "realsr_hip.h")])
    exe = str(tmp_path / "host")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-o", exe, str(src), "-L", lib, "-lrealsr_hip", "-Wl,-rpath," + lib])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "version " in r.stdout and "shares 8: 0 " in r.stdout
    import torch
    if not torch.cuda.is_available():  # the GPU-less container: the C host sees a loud failure, not a fallback
        assert "create rc -" in r.stdout and "ctx null" in r.stdout and "msg []" not in r.stdout, r.stdout


def test_real_model_harness_refuses_weights_that_overflow_fp16(tmp_path):
    """Weight statistics nobody has seen (the real x4.bin blobs are absent): the engine -- like the reference's Vulkan path,
    realsr.cpp:44-46 -- STORES every feature map as fp16.  The network is positively homogeneous up to its biases, so fp16's
    relative precision does not care about the scale of the activations (the hot = 32 model of tests/test_gpu_round2.py holds +-1
    with trunk peaks of ~6e3) -- until a value passes 65,504.  A model whose trunk does (synth hot = 512: 194 x 512 = 9.9e4) must
    be refused LOUDLY by tools/check_real_model.py's range guard, which is the one thing a range can tell; precision itself is
    measured directly by the harness's GPU legs (pre-quantise error, C1 +-1)."""
    import subprocess
    import sys
    from realsr_ncnn_vulkan_amd import synth
    d = synth.make_model_dir(str(tmp_path), "models-hot512", 44, hot=512.0, last_gain=0.15)
    tool = os.path.join(ROOT, "tools", "check_real_model.py")
    r = subprocess.run([sys.executable, tool, d, "--no-gpu", "--range-tile", "48"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1, r.stdout + r.stderr
    assert "OVERFLOWS fp16 storage" in r.stdout and "RESULT: FAILED" in r.stdout, r.stdout


def test_flow_hook_generator_schedules_every_piece_once(tmp_path):
    """tools/gen_flow_hooks.py (run by the csrc Makefile): the deferred drain of one block = 4 rows x (8 pair conversions + 2 plane
    stores) = 40 pieces, each behind exactly one MFMA cell, a row's stores behind its conversions, every third cell by default;
    steps without a piece get an empty hook and the plain step schedule (RSR_PIN_S<step> 1)."""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("RSR_HOOK_STRIDE", None)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_flow_hooks.py"), str(tmp_path)], env=env, stdout=subprocess.DEVNULL)
    inc = (tmp_path / "conv_flow_hooks.inc").read_text().splitlines()
    cells = {}
    for ln in inc:
        m = re.match(r"#define RSR_HK_S(\d+)_(\d+) (.*)", ln)
        if m and "pk[" in m.group(3):
            cells[12 * int(m.group(1)) + int(m.group(2))] = m.group(3)
    order = sorted(cells)
    assert len(order) == 40 and order[0] == 1 and all(b - a == 3 for a, b in zip(order, order[1:])) and order[-1] < 120
    kinds = ["ST" if "row_store1" in cells[c] else "P" for c in order]
    assert kinds == (["P"] * 8 + ["ST"] * 2) * 4
    stores = [re.search(r"row_store1\(o_, od, (\d), (\d)\)", cells[c]).groups() for c in order if "row_store1" in cells[c]]
    assert stores == [(str(r), str(p)) for r in range(4) for p in range(2)]
    pins = dict(re.match(r"#define RSR_PIN_S(\d+) (\d)", ln).groups() for ln in inc if ln.startswith("#define RSR_PIN_S"))
    assert [pins[str(s)] for s in range(12)] == ["0"] * 10 + ["1"] * 2
    undef = (tmp_path / "conv_flow_hooks_undef.inc").read_text()
    assert undef.count("#undef RSR_PIN_S") == 12
