"""GPU bring-up / A-B sweep for conv3x3_flow -- not a pytest.  Run on the MI355X box:

    python tools/flow_diag.py [conv] [net] [e2e] [perf]

conv: single layers vs the oracle conv for every flow_flags value, one workgroup-per-CU and few-workgroup grids
      (several blocks per workgroup: ring wrap-around, deferred epilogue across blocks)
net:  whole network on one tile vs the oracle (pre-quantise)          e2e: rsr_process vs oracle process
perf: C2 frame time per kernel / flag variant with the per-layer-class table
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

MODELS = os.environ.get("RSR_MODELS", "/tmp/rsr_models")


def stats(name, got, ref):
    d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    tol = np.abs(ref) * 2.0 ** -10 + 1e-3
    bad = int((d > tol).sum())
    print("  %-58s max|d|=%.3e p99.9=%.3e bad=%d/%d" % (name, d.max(), np.quantile(d, 0.999), bad, d.size), flush=True)
    return bad, d


def sec_conv(sr):
    print("== conv layers, kernel 4 vs oracle conv")
    rng = np.random.default_rng(0)
    cases = [(64, 32, 20, 40, False), (96, 32, 17, 33, False), (128, 32, 16, 32, False), (160, 32, 48, 64, False),
             (192, 64, 16, 32, False), (3, 64, 9, 70, False), (64, 3, 33, 31, False), (64, 64, 10, 21, True),
             (64, 64, 1, 1, False), (64, 32, 100, 130, False), (192, 64, 70, 90, False), (64, 64, 40, 50, True)]
    total_bad = 0
    for flags in (0, 3):
        for ncu in (256, 8):
            sr.set_option("flow_flags", flags)
            sr.set_option("num_cu", ncu)
            for cin, cout, h, w, ups in cases:
                if ncu == 8 and h * w < 2000:
                    continue
                x = rng.standard_normal((cin, h, w)).astype(np.float16)
                wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
                b = rng.standard_normal(cout).astype(np.float32)
                for lrelu in (False, True):
                    xr = x.astype(np.float32)
                    if ups:
                        xr = xr.repeat(2, axis=1).repeat(2, axis=2)
                    ref = oracle.conv3x3(xr, wt, b, 2 if lrelu else 0, 0.2)
                    got = sr.conv3x3(x, wt, b, lrelu=lrelu, upsample2x=ups).astype(np.float32)
                    bad, d = stats("flags=%d ncu=%d %d->%d %dx%d ups=%d lrelu=%d" % (flags, ncu, cin, cout, h, w, ups, lrelu), got, ref)
                    total_bad += bad
                    if bad:
                        print("    per-channel bad (first 16):", (d > 1e-2).reshape(cout, -1).sum(1)[:16])
                        print("    per-row bad (first 24):", (d > 1e-2).sum(axis=(0, 2))[:24])
                        print("    per-col bad (first 40):", (d > 1e-2).sum(axis=(0, 1))[:40])
    sr.set_option("num_cu", 256)
    sr.set_option("flow_flags", 0)
    print("conv section: total bad =", total_bad)


def sec_res(sr):
    print("== residual epilogues (RDB conv5 / every-third / trunk forms) vs numpy")
    rng = np.random.default_rng(3)
    total_bad = 0
    for (cin, cout, h, w) in [(192, 64, 20, 40), (64, 64, 33, 50), (192, 64, 70, 90)]:
        x = rng.standard_normal((cin, h, w)).astype(np.float16)
        wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        res = rng.standard_normal((cout, h, w)).astype(np.float16)
        conv = oracle.conv3x3(x.astype(np.float32), wt, b, 0, 0.2)
        forms = {"conv5": (0.2, True, None, 1.0, 0.2 * conv + x[:cout].astype(np.float32)),
                 "conv5+rrdb": (0.2, True, res, 0.2, 0.2 * (0.2 * conv + x[:cout].astype(np.float32)) + res.astype(np.float32)),
                 "trunk": (1.0, False, res, 1.0, conv + res.astype(np.float32))}
        for kern, flags, dbg, ncu in ((4, 0, 0, 256), (4, 1, 0, 256), (4, 0, 0, 8), (4, 1, 0, 8), (4, 0, 32, 8)):
            sr.set_option("flow_flags", flags)
            sr.set_option("dbg", dbg)
            sr.set_option("num_cu", ncu)
            for name, (s1, own, r, s2, ref) in forms.items():
                got = sr.conv3x3_res(x, wt, b, s1, own_input_residual=own, res=r, s2=s2).astype(np.float32)
                d = np.abs(got - ref)
                bad = int((d > np.abs(ref) * 2.0 ** -9 + 2e-3).sum())
                total_bad += bad
                print("  %d->%d %dx%d kernel=%d flags=%d dbg=%d ncu=%d %-11s max|d|=%.3e bad=%d/%d%s" % (
                    cin, cout, h, w, kern, flags, dbg, ncu, name, np.nanmax(d), bad, d.size, "  NaN!" if np.isnan(got).any() else ""), flush=True)
                if bad:
                    print("    per-channel bad:", (d > 1e-2).reshape(cout, -1).sum(1))
    sr.set_option("flow_flags", 0)
    sr.set_option("dbg", 0)
    sr.set_option("num_cu", 256)
    print("res section: total bad =", total_bad)


def sec_net(sr, net):
    print("== whole network on one tile vs oracle (pre-quantise, [0,1] units)")
    q = lambda v: np.clip(np.floor(v * 255.0 + 0.5), 0, 255)
    for (w, h) in [(28, 24), (52, 52), (75, 40)]:
        img = synth.make_image(5, w, h)
        x = (img.astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.)).astype(np.float16)
        ref = net.forward(x.astype(np.float32))
        for kern, flags in ((4, 0), (4, 1), (4, 2)):
            sr.set_option("flow_flags", flags)
            got = sr.net_forward(x).astype(np.float32)
            d = np.abs(got - ref)
            du = np.abs(q(got) - q(ref))
            print("  tile %dx%d kernel=%d flags=%d: max %.3e p99.9 %.3e mean %.3e | u8 max %d frac!=0 %.4f" % (
                w, h, kern, flags, d.max(), np.quantile(d, 0.999), d.mean(), du.max(), (du > 0).mean()), flush=True)
    sr.set_option("flow_flags", 0)


def sec_e2e(net):
    print("== rsr_process vs oracle process (uint8, same tile size)")
    pp, bp = os.path.join(MODELS, "models-DF2K", "x4.param"), os.path.join(MODELS, "models-DF2K", "x4.bin")
    for (w, h, c, T, tta) in [(50, 43, 3, 32, False), (40, 33, 3, 32, True), (37, 41, 4, 32, False), (5, 3, 3, 32, False)]:
        sr = R.RealSR(0, tta_mode=tta)
        sr.load(pp, bp)
        sr.tilesize = T
        img = synth.make_image(9, w, h, c)
        ref = net.process(img, T, tta=tta)
        got = sr.process(img)
        d = np.abs(got.astype(int) - ref.astype(int))
        print("  %dx%dx%d T=%d tta=%d: max |d|=%d  frac!=0 %.4f  frac>1 %.6f" % (w, h, c, T, tta, d.max(), (d > 0).mean(), (d > 1).mean()), flush=True)
        sr.close()


def layer_table(ct, npx):
    specs = synth.conv_specs()
    groups = {}
    for i, (cin, cout, act) in enumerate(specs):
        lvl = 0 if i <= 346 else (1 if i == 347 else 2)
        name = "%3d->%-2d @%dx" % (cin, cout, 1 << lvl)
        g = groups.setdefault(name, [0, 0.0, 0.0])
        g[0] += 1
        g[1] += ct[i]
        g[2] += 2.0 * 9 * cin * cout * npx * (4 ** lvl)
    for name, (n, ms, fl) in groups.items():
        print("      %s x%-3d %8.3f ms  %7.1f TFLOP/s (%4.1f%%)  avg %7.1f us" % (name, n, ms, fl / ms / 1e9, fl / ms / 1e9 / 25, ms / n * 1e3))
    # conv5 of RDB j of every RRDB (j = 2 also carries the RRDB residual)
    c5 = [[ct[1 + (3 * b + j) * 5 + 4] for b in range(23)] for j in range(3)]
    print("      conv5 by RDB position: " + "  ".join("j=%d %.1f us" % (j, 1e3 * sum(v) / len(v)) for j, v in enumerate(c5)))


def sec_perf():
    print("== timing: 1920x1080 T=200 (C2), device API")
    import torch
    pp, bp = os.path.join(MODELS, "models-DF2K", "x4.param"), os.path.join(MODELS, "models-DF2K", "x4.bin")
    sr = R.RealSR(0)
    sr.load(pp, bp)
    sr.tilesize = 200
    w, h = 1920, 1080
    npx = 2544000
    img = synth.make_image(3, w, h)
    d_in = torch.from_numpy(img).cuda()
    d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
    variants = os.environ.get("RSR_PERF_VARIANTS", "flow_flags=0;dbg=32,trim=0;dbg=0,trim=1;flow_flags=1;flow_flags=2;flow_flags=0").split(";")
    sums = {}
    for var in variants:
        opts = dict(kv.split("=") for kv in var.split(",") if kv)
        for k, v in opts.items():
            sr.set_option({"ws": "max_workspace_mb"}.get(k, k), int(v))
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())  # warmup (allocs)
        torch.cuda.synchronize()
        t = time.time()
        n = 4
        for _ in range(n):
            sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.time() - t) / n
        sums[var] = int(d_out[::97, ::89].to(torch.int64).sum().item())
        sr.set_profiling(True)
        sr.get_conv_times(reset=True)
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        p = sr.get_profile()
        ct = sr.get_conv_times()
        sr.set_profiling(False)
        print("  %s: %.1f ms/frame = %.1f Mpix/s out; conv %.1f ms -> %.1f TFLOP/s (%.1f%% of 2.5 PF); pre %.3f post %.3f ms; checksum %d" % (
            var, dt * 1e3, 33.1776 / dt, p["conv_ms"], p["conv_flops"] / p["conv_ms"] / 1e9,
            p["conv_flops"] / p["conv_ms"] / 1e9 / 2500 * 100, p["pre_ms"], p["post_ms"], sums[var]), flush=True)
        layer_table(ct, npx)
    out = R.PinnedArray((h * 4, w * 4, 3))
    pin = R.PinnedArray((h, w, 3))
    pin.array[:] = img
    page_out = np.empty((h * 4, w * 4, 3), dtype=np.uint8)
    for name, src, dst in (("pageable", img, page_out), ("pinned", pin.array, out.array)):
        sr.process(src, out=dst)
        t = time.time()
        for _ in range(3):
            sr.process(src, out=dst, push_params=False)
        dt = (time.time() - t) / 3
        print("  host API rsr_process, %s buffers: %.1f ms/frame = %.1f Mpix/s" % (name, dt * 1e3, 33.1776 / dt), flush=True)
    sr.close()


def main():
    secs = sys.argv[1:] or ["conv", "net", "e2e", "perf"]
    d = synth.make_model_dir(MODELS, "models-DF2K", 42)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    net = oracle.OracleNet(pp, bp)
    sr = R.RealSR(0)
    sr.load(pp, bp)
    for s in secs:
        try:
            {"conv": lambda: sec_conv(sr), "res": lambda: sec_res(sr), "net": lambda: sec_net(sr, net), "e2e": lambda: sec_e2e(net), "perf": sec_perf}[s]()
        except Exception:
            traceback.print_exc()
            print("SECTION FAILED:", s, flush=True)
    sr.close()


if __name__ == "__main__":
    main()
