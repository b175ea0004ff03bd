"""GPU diagnostic sweep (not a pytest): run on the MI355X box, prints a detailed parity / timing log.

    python tests/gpu_diag.py [section ...]       sections: conv shader net e2e perf
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle  # noqa: E402
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

MODELS = os.environ.get("RSR_MODELS", "/tmp/rsr_models")


def stats(name, got, ref):
    got = got.astype(np.float64)
    ref = ref.astype(np.float64)
    d = np.abs(got - ref)
    print("  %-44s max|d|=%.3e  p99.9=%.3e  mean=%.3e  ref_rms=%.3e" % (
        name, d.max(), np.quantile(d, 0.999), d.mean(), np.sqrt((ref ** 2).mean())), flush=True)
    return d.max()


def sec_conv(sr):
    print("== single conv layers vs oracle conv (fp32 on the same fp16-rounded inputs/weights)")
    rng = np.random.default_rng(0)
    cases = [(64, 32, 20, 40, False), (96, 32, 17, 33, False), (192, 64, 16, 32, False), (3, 64, 9, 70, False),
             (64, 3, 33, 31, False), (64, 64, 10, 21, True), (160, 32, 48, 64, False)]
    for dma in (3, 2):
        sr.set_option("kernel", dma)
        sr.set_option("use_dma", 1)
        for cin, cout, h, w, ups in cases:
            x = rng.standard_normal((cin, h, w)).astype(np.float16)
            wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float16).astype(np.float32)
            b = rng.standard_normal(cout).astype(np.float32)
            for lrelu in (False, True):
                xr = x.astype(np.float32)
                if ups:
                    xr = xr.repeat(2, axis=1).repeat(2, axis=2)
                ref = oracle.conv3x3(xr, wt, b, 2 if lrelu else 0, 0.2)
                got = sr.conv3x3(x, wt, b, lrelu=lrelu, upsample2x=ups).astype(np.float32)
                m = stats("kern=%d %d->%d %dx%d ups=%d lrelu=%d" % (dma, cin, cout, h, w, ups, lrelu), got, ref)
                if m > 0.05:
                    # help localise a layout bug
                    print("    first rows got:", got[0, 0, :6], " ref:", ref[0, 0, :6])
                    print("    per-channel max err (first 8):", np.abs(got - ref).reshape(cout, -1).max(1)[:8])
                    print("    per-row max err (first 8):", np.abs(got - ref).max(axis=(0, 2))[:8])
                    print("    per-col max err (first 8):", np.abs(got - ref).max(axis=(0, 1))[:8])
    sr.set_option("kernel", 3)


def sec_shader(sr):
    print("== pre/post shader kernels vs scalar restatement (bit exact)")
    rng = np.random.default_rng(1)
    T, P = 32, 10
    for (w, h, c) in [(50, 43, 3), (200, 37, 3), (7, 9, 3)]:
        img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        xt, yt = (w + T - 1) // T, (h + T - 1) // T
        for yi in range(yt):
            y0, y1 = max(yi * T - P, 0), min((yi + 1) * T + P, h)
            band = np.ascontiguousarray(img[y0:y1])
            for xi in range(xt):
                twn = min((xi + 1) * T, w) - xi * T
                thn = min((yi + 1) * T, h) - yi * T
                tw, th = twn + 2 * P, thn + 2 * P
                a = (P, P, xi * T, min(yi * T, P))
                r1 = oracle.preproc(band, tw, th, *a)
                r2 = sr.preproc(band, tw, th, *a)
                t1 = oracle.preproc_tta(band, tw, th, *a)
                t2 = sr.preproc_tta(band, tw, th, *a)
                bad = [int((r1.view(np.uint16) != r2.view(np.uint16)).sum())] + [
                    int((t1[k].view(np.uint16) != t2[k].view(np.uint16)).sum()) for k in range(8)]
                if any(bad):
                    print("  %dx%d tile(%d,%d) tw=%d th=%d band=%s mismatches plain+8tta: %s" % (w, h, yi, xi, tw, th, band.shape, bad))
                    k = next(i for i, b in enumerate(bad) if b)
                    A, B = (r1, r2) if k == 0 else (t1[k - 1], t2[k - 1])
                    idx = np.argwhere(A.view(np.uint16) != B.view(np.uint16))
                    print("    blob %d first mismatches (c,y,x):" % k, idx[:6].tolist(), "oracle", [float(A[tuple(i)]) for i in idx[:6]], "hip", [float(B[tuple(i)]) for i in idx[:6]])
        print("  preproc %dx%dx%d done" % (w, h, c), flush=True)


def sec_net(sr, net):
    print("== whole network on one tile vs oracle (pre-quantise, [0,1] units)")
    for (w, h) in [(28, 24), (52, 52)]:
        img = synth.make_image(5, w, h)
        x = (img.astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.)).astype(np.float16)
        ref = net.forward(x.astype(np.float32))
        for tf in (1, 0):
            for dma in (1, 0):
                sr.set_option("trunk_fp32", tf)
                sr.set_option("use_dma", dma)
                t = time.time()
                got = sr.net_forward(x).astype(np.float32)
                dt = time.time() - t
                stats("tile %dx%d trunk_fp32=%d dma=%d (%.0f ms)" % (w, h, tf, dma, dt * 1e3), got, ref)
                q = lambda v: np.clip(np.floor(v * 255.0 + 0.5), 0, 255)
                du = np.abs(q(got) - q(ref))
                print("      u8: max diff %d, frac!=0 %.4f, frac>1 %.6f" % (du.max(), (du > 0).mean(), (du > 1).mean()), flush=True)
    sr.set_option("trunk_fp32", 0)
    sr.set_option("use_dma", 1)


def sec_e2e(net):
    print("== rsr_process vs oracle process (uint8, same tile size)")
    pp, bp = os.path.join(MODELS, "models-DF2K", "x4.param"), os.path.join(MODELS, "models-DF2K", "x4.bin")
    for (w, h, c, T, tta) in [(40, 33, 3, 32, True), (37, 41, 4, 32, False), (5, 3, 3, 32, False)]:
        sr = R.RealSR(0, tta_mode=tta)
        sr.load(pp, bp)
        sr.tilesize = T
        img = synth.make_image(9, w, h, c)
        t = time.time()
        ref, ref32 = net.process(img, T, tta=tta, want_f32=True)
        t_or = time.time() - t
        t = time.time()
        got = sr.process(img)
        t_gpu = time.time() - t
        d = np.abs(got.astype(int) - ref.astype(int))
        print("  %dx%dx%d T=%d tta=%d: max |d|=%d  frac!=0 %.4f  frac>1 %.6f   (oracle %.1fs, gpu %.2fs)" % (
            w, h, c, T, tta, d.max(), (d > 0).mean(), (d > 1).mean(), t_or, t_gpu), flush=True)
        if d.max() > 1:
            ys, xs, cs = np.nonzero(d > 1)
            print("    worst at", list(zip(ys[:5], xs[:5], cs[:5])), "per-channel max", d.reshape(-1, c).max(0))
        sr.close()


def layer_table(ct, npx):
    """ct: per-conv ms (351) for ONE frame; npx: padded LR px per frame."""
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
    variants = os.environ.get("RSR_PERF_VARIANTS", "kernel=2,trunk_fp32=1;kernel=2,trunk_fp32=0;kernel=1,trunk_fp32=0").split(";")
    for var in variants:
        opts = dict(kv.split("=") for kv in var.split(",") if kv)
        for k, v in opts.items():
            sr.set_option({"dma": "use_dma", "ws": "max_workspace_mb"}.get(k, k), int(v))
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())  # warmup (allocs)
        torch.cuda.synchronize()
        t = time.time()
        n = 3
        for _ in range(n):
            sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.time() - t) / n
        sr.set_profiling(True)
        sr.get_conv_times(reset=True)
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        p = sr.get_profile()
        ct = sr.get_conv_times()
        sr.set_profiling(False)
        print("  %s: %.1f ms/frame = %.1f Mpix/s out; conv %.1f ms, %.1f TFLOP -> %.1f TFLOP/s (%.1f%% of 2.5 PF); pre %.3f ms post %.3f ms" % (
            var, dt * 1e3, 33.1776 / dt, p["conv_ms"], p["conv_flops"] / 1e12, p["conv_flops"] / p["conv_ms"] / 1e9,
            p["conv_flops"] / p["conv_ms"] / 1e9 / 2500 * 100, p["pre_ms"], p["post_ms"]), flush=True)
        layer_table(ct, npx)
    sr.process(img)
    t = time.time()
    for _ in range(3):
        out = sr.process(img)
    dt = (time.time() - t) / 3
    print("  host API rsr_process (pageable H2D 6.2 MB + D2H 99.5 MB incl.): %.1f ms/frame = %.1f Mpix/s" % (dt * 1e3, 33.1776 / dt))
    sr.close()


def main():
    secs = sys.argv[1:] or ["conv", "shader", "net", "e2e", "perf"]
    d = synth.make_model_dir(MODELS, "models-DF2K", 42)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    net = oracle.OracleNet(pp, bp)
    print("oracle threads:", oracle.max_threads())
    sr = R.RealSR(0)
    sr.load(pp, bp)
    for s in secs:
        try:
            if s == "conv":
                sec_conv(sr)
            elif s == "shader":
                sec_shader(sr)
            elif s == "net":
                sec_net(sr, net)
            elif s == "e2e":
                sec_e2e(net)
            elif s == "perf":
                sec_perf()
            elif s == "trace":
                sec_trace()
        except Exception:
            traceback.print_exc()
            print("SECTION FAILED:", s, flush=True)




def sec_trace():
    """per-stage barrier stamps of workgroup 0 / MFMA wave 0 for a few conv launches (kernel 3, NT=1)"""
    import torch  # noqa: F401
    pp, bp = os.path.join(MODELS, "models-DF2K", "x4.param"), os.path.join(MODELS, "models-DF2K", "x4.bin")
    sr = R.RealSR(0)
    sr.load(pp, bp)
    sr.tilesize = 200
    img = synth.make_image(3, 1920, 1080)
    sr.process(img)
    if os.environ.get("RSR_TRACE_DBG"):
        sr.set_option("dbg", int(os.environ["RSR_TRACE_DBG"]))
    for ci, name in [(1, "64->32"), (4, "160->32"), (5, "192->64")]:
        sr.set_option("trace_conv", ci)
        sr.process(img)
        tr = sr.get_trace(1024).astype(np.int64).reshape(-1, 2)
        n = int((tr[:, 0] > 0).sum())
        tr = tr[:n]
        arrive, release = tr[:, 0], tr[:, 1]
        wait = release - arrive
        busy = arrive[1:] - release[:-1]
        nplanes = {1: 2, 4: 5, 5: 6}[ci]
        print("  conv %d (%s): %d stages traced, total %d ticks" % (ci, name, n, release[-1] - arrive[0]))
        print("    barrier wait ticks: mean %.0f  p50 %.0f  p90 %.0f  max %d" % (wait.mean(), np.median(wait), np.quantile(wait, 0.9), wait.max()))
        last = (np.arange(n - 1) % nplanes) == nplanes - 1  # segments that contain an epilogue
        print("    busy (release->next arrival): plain stages mean %.0f p50 %.0f | stages with epilogue mean %.0f p50 %.0f" % (
            busy[~last].mean(), np.median(busy[~last]), busy[last].mean(), np.median(busy[last])))
        print("    first 12 (wait,busy):", [(int(w), int(b)) for w, b in zip(wait[:12], busy[:12])])
        if os.environ.get("RSR_OVLTRACE") and ci == 5:
            full = sr.get_trace(8192).astype(np.int64)
            ws = full[1024:1024 + 8 * n].reshape(n, 8)[:, :6]
            rows = np.array([[ws[s_, 0] - release[s_]] + [ws[s_, k + 1] - ws[s_, k] for k in range(5)] +
                             [(arrive[s_ + 1] - ws[s_, 5]) if s_ + 1 < n else 0] for s_ in range(n) if ws[s_, 0] > 0])
            print("    NT=2 epilogue, ticks (median over %d): stage-MFMA %d | handshake %d | row0 %d | row1 %d | row2 %d | row3 %d | to-barrier %d" % (
                (len(rows),) + tuple(int(np.median(rows[:, k])) for k in range(7))))
            print("      p90:", [int(np.quantile(rows[:, k], 0.9)) for k in range(7)])
        if os.environ.get("RSR_OVLTRACE") and ci != 5:
            full = sr.get_trace(8192).astype(np.int64)
            ws = full[1024:1024 + 8 * n].reshape(n, 8)[:, :6]
            rows = []
            for s_ in range(n):
                if ws[s_, 0] > 0:
                    rows.append([ws[s_, 0] - release[s_]] + [ws[s_, k + 1] - ws[s_, k] for k in range(5)] +
                                [(arrive[s_ + 1] - ws[s_, 5]) if s_ + 1 < n else 0])
            rows = np.array(rows)
            print("    overlapped-epilogue stage, ticks (median over %d): main48 %d | row0 %d | row1||E0 %d | row2||E1 %d | row3||E2 %d | tail %d | to-barrier %d" % (
                (len(rows),) + tuple(int(np.median(rows[:, k])) for k in range(7))))
            print("      p90:", [int(np.quantile(rows[:, k], 0.9)) for k in range(7)])
    sr.set_option("trace_conv", -1)
    sr.close()


if __name__ == "__main__":
    main()
