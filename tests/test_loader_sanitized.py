"""The host-only model loader (csrc/model.cpp: ncnn .param parser, .bin reader, graph validation, weight packer, packed-blob check --
what replaces ncnn::Net::load_param / load_model, realsr.cpp:75-76) under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU
build (GPU sanitizers are not available on this pool): a small g++ harness loads the valid synthetic model, then several hundred
MUTATED .param / .bin files and packed blobs (truncations, flipped bytes, spliced lines, wild counts and offsets).  Every input must
end in RSR_OK or a clean negative RSR_E_* -- never a sanitizer report, a crash or a hang."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "realsr-ncnn-vulkan_amd", "csrc")

HARNESS = r'''
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "model.h"
using namespace rsr;
// usage: harness load <param> <bin> | blob <file>      prints "rc <code>"
int main(int argc, char** argv)
{
    if (argc >= 4 && !std::strcmp(argv[1], "load"))
    {
        Model m;
        std::string e;
        int rc = load_model(argv[2], argv[3], m, e);
        if (rc == 0)
        {
            std::vector<unsigned char> buf(packed_size(m));
            rc = pack_model(m, buf.data(), buf.size(), e);
            if (rc == 0) rc = check_packed(buf.data(), packed_table_bytes(), buf.size(), e);
            if (rc == 0 && argc >= 5)
            {
                std::ofstream f(argv[4], std::ios::binary);
                f.write(reinterpret_cast<const char*>(buf.data()), std::streamsize(packed_table_bytes()));
                std::printf("total %zu\n", buf.size());
            }
        }
        std::printf("rc %d\n", rc);
        return 0;
    }
    if (argc >= 4 && !std::strcmp(argv[1], "blob"))
    {
        std::ifstream f(argv[2], std::ios::binary);
        std::vector<char> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        std::string e;
        const int rc = b.size() >= packed_table_bytes() ? check_packed(b.data(), packed_table_bytes(), size_t(std::atoll(argv[3])), e) : -3;
        std::printf("rc %d\n", rc);
        return 0;
    }
    return 2;
}
'''


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    src = d / "harness.cpp"
    src.write_text(HARNESS)
    exe = str(d / "harness")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", CSRC,
                           "-o", exe, str(src), os.path.join(CSRC, "model.cpp")])
    return exe


def run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (args, r.returncode, r.stderr[-1500:])
    return int(r.stdout.strip().splitlines()[-1].split()[1]), r.stdout


def test_loader_survives_mutated_models_under_asan_ubsan(harness, model_dir, tmp_path):
    pp, bp = os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin")
    head = tmp_path / "head.bin"
    rc, out = run(harness, "load", pp, bp, head)
    assert rc == 0
    total = int(out.split("total")[1].split()[0])
    rng = np.random.default_rng(2026)
    param = open(pp, "rb").read()
    lines = param.split(b"\n")
    binb = np.fromfile(bp, dtype=np.uint8)
    codes = {}

    def note(rc):
        assert rc <= 0
        codes[rc] = codes.get(rc, 0) + 1

    # ---- .param mutations: truncation, dropped / duplicated / shuffled lines, byte flips, wild numbers ----
    for i in range(120):
        kind = i % 6
        ls = list(lines)
        if kind == 0:
            data = param[:int(rng.integers(0, len(param)))]
        elif kind == 1:
            del ls[int(rng.integers(0, len(ls)))]
            data = b"\n".join(ls)
        elif kind == 2:
            k = int(rng.integers(2, len(ls) - 1))
            ls.insert(k, ls[int(rng.integers(2, len(ls) - 1))])
            data = b"\n".join(ls)
        elif kind == 3:
            a = bytearray(param)
            for _ in range(int(rng.integers(1, 8))):
                a[int(rng.integers(0, len(a)))] = int(rng.integers(0, 256))
            data = bytes(a)
        elif kind == 4:
            k = int(rng.integers(2, len(ls) - 1))
            ls[k] = ls[k].replace(b"0=32", b"0=%d" % int(rng.choice([-1, 0, 2 ** 31 - 1, 10 ** 12]))).replace(b"6=", b"6=9" if rng.random() < 0.5 else b"6=-")
            data = b"\n".join(ls)
        else:
            ls[1] = b"%d %d" % (int(rng.choice([0, -5, 10 ** 9, 999])), int(rng.choice([0, -1, 10 ** 9, 1275])))
            data = b"\n".join(ls)
        p = tmp_path / "m.param"
        p.write_bytes(data)
        note(run(harness, "load", p, bp)[0])
    # ---- .bin mutations: truncation, wrong tags, extension ----
    for i in range(40):
        kind = i % 4
        if kind == 0:
            b = binb[:int(rng.integers(0, binb.size))]
        elif kind == 1:
            b = binb.copy()
            b[:4] = rng.integers(0, 256, 4)  # the first weight tag
        elif kind == 2:
            b = binb.copy()
            off = int(rng.integers(0, binb.size - 4)) & ~3
            b[off:off + 4] = rng.integers(0, 256, 4)
        else:
            b = np.concatenate([binb, rng.integers(0, 256, int(rng.integers(1, 4096)), dtype=np.uint8)])
        q = tmp_path / "m.bin"
        b.tofile(q)
        note(run(harness, "load", pp, q)[0])
    # ---- packed blob headers from "outside" (a broadcast): flipped bytes, wild offsets, wrong totals ----
    hb = np.fromfile(head, dtype=np.uint8)
    assert run(harness, "blob", head, total)[0] == 0
    for i in range(120):
        h = hb.copy()
        kind = i % 3
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                h[int(rng.integers(0, h.size))] = int(rng.integers(0, 256))
            tot = total
        elif kind == 1:
            off = (int(rng.integers(0, h.size - 8)) // 8) * 8
            wild = [0, 2 ** 63, 2 ** 64 - 1, total, total - 1, 2 ** 32][int(rng.integers(0, 6))]
            h[off:off + 8] = np.frombuffer(int(wild).to_bytes(8, "little"), dtype=np.uint8)
            tot = total
        else:
            tot = int(rng.choice([0, 1, h.size, total - 1, total // 2, 2 ** 40]))
        q = tmp_path / "h.bin"
        h.tofile(q)
        note(run(harness, "blob", q, tot)[0])
    print("return codes over 280 mutated inputs:", dict(sorted(codes.items())))
    assert sum(v for k, v in codes.items() if k < 0) >= 200  # the mutations do bite; the few that leave the model valid return 0
