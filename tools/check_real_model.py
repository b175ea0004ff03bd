#!/usr/bin/env python3
"""Real-weights harness: everything this repository measures runs on synthetic weights (synth.make_weights) because the reference
checkout ships x4.param without x4.bin (/root/reference/.MISSING_LARGE_BLOBS).  The day a real blob is at hand:

    python tools/check_real_model.py <dir with x4.param + x4.bin>  [--no-gpu] [--no-bench] [--json out.json]
    python tools/check_real_model.py                # looks in $RSR_REAL_MODELS, ./models/models-DF2K, ./models/models-DF2K_JPEG

It (1) identifies the .bin encoding by its size -- 33,424,520 B = fp16-tagged (flag 0x01306B47 per conv), 66,793,352 B = raw
fp32 (SURVEY a-7; what ncnn::Net::load_model reads, realsr.cpp:75-76) -- and cross-checks with the parser; (2) host-only:
parses, validates the graph, packs the blob, reports weight statistics (max |w|, how many weights are not fp16-representable);
(3) walks the network in fp32 on the CPU oracle over a real-image-like tile and reports the ACTIVATION RANGE per RDB -- the
engine stores every feature map as fp16 (max 65504): the overflow guard for weights nobody has run through it before; (4) on
the GPU: BASELINE C1 (256x256, tile 128) whole frame against the oracle, +-1 uint8, the pre-quantise error (max / p99.9, the
tolerances of tests/test_gpu_parity.py: 3.0e-3 / 1.5e-3 with fp16 storage; when the headroom of the +-1 bar is below 1.5 the harness
switches to precise mode -- rsr_set_option precise 1 -- and says so), and (5) the C2 bench leg (1080p, tile 200, device-resident, 5 steps).

Exit status: 0 = every check that could run passed, or nothing to check (no blob: prints SKIP); 1 = a check failed.
The oracle is the checker here, as in tests/ -- this is test infrastructure, not a product path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SIZE_FP16_TAGGED, SIZE_RAW_FP32 = 33424520, 66793352  # 351 tags + fp16 weights + fp32 biases | fp32 everything
FP16_MAX = 65504.0


def find_model_dir(arg):
    cands = [arg] if arg else [os.environ.get("RSR_REAL_MODELS"), os.path.join(ROOT, "models", "models-DF2K"),
                               os.path.join(ROOT, "models", "models-DF2K_JPEG"), "/root/reference/models/models-DF2K",
                               "/root/reference/models/models-DF2K_JPEG"]
    for d in cands:
        if d and os.path.isfile(os.path.join(d, "x4.param")) and os.path.isfile(os.path.join(d, "x4.bin")):
            return d
    return None


def encoding_by_size(n):
    if n == SIZE_FP16_TAGGED:
        return "fp16-tagged"
    if n == SIZE_RAW_FP32:
        return "raw-fp32"
    return None


def host_checks(pp, bp, rep):
    import oracle
    import realsr_ncnn_vulkan_amd as R
    ok = True
    size = os.path.getsize(bp)
    rep["bin_bytes"] = size
    rep["encoding_by_size"] = encoding_by_size(size)
    info = R.model_info(pp, bp)  # parses + validates the DAG against RRDBNet(3,3,64,23,32); raises on anything else
    rep["model_info"] = info
    enc_names = {1: "fp16-tagged", 0: "raw-fp32", 2: "mixed/table"}
    rep["encoding_by_parser"] = enc_names.get(info["bin_encoding"], str(info["bin_encoding"]))
    if rep["encoding_by_size"] is None:
        print("  NOTE: %d bytes is neither %d (fp16-tagged) nor %d (raw fp32): per-conv mixed encoding; the parser says %s" % (
            size, SIZE_FP16_TAGGED, SIZE_RAW_FP32, rep["encoding_by_parser"]))
    elif rep["encoding_by_size"] != rep["encoding_by_parser"]:
        print("  FAIL: size says %s, parser says %s" % (rep["encoding_by_size"], rep["encoding_by_parser"]))
        ok = False
    blob = R.model_pack(pp, bp)
    rep["packed_bytes"] = int(blob.size)
    net = oracle.OracleNet(pp, bp)
    wmax, n_unrep, n_tot, n_inf = 0.0, 0, 0, 0
    for i in range(net.num_convs):
        w = net.conv(i)["weight"]
        wmax = max(wmax, float(np.abs(w).max()))
        h = w.astype(np.float16)
        n_inf += int(np.isinf(h).sum())
        n_unrep += int((h.astype(np.float32) != w).sum())
        n_tot += w.size
    rep["weights"] = {"count": n_tot, "max_abs": wmax, "not_fp16_representable": n_unrep, "overflow_to_inf_in_fp16": n_inf}
    print("  encoding %s (%d B); %d convs, %d weights, max |w| = %.4g; %d (%.3f %%) not fp16-representable (the packer rounds them, "
          "as ncnn's fp16-storage path does); packed blob %.1f MB" % (rep["encoding_by_parser"], size, net.num_convs, n_tot, wmax, n_unrep,
                                                                      100.0 * n_unrep / max(n_tot, 1), blob.size / 1e6))
    if n_inf:
        print("  FAIL: %d weights overflow fp16" % n_inf)
        ok = False
    return ok, net


def activation_ranges(net, tile):
    """The graph of x4.param in fp32 on the CPU (oracle.conv3x3 layer by layer, Concat / residuals in numpy): max |value| of what
    the engine stores as fp16 -- conv_first, per RDB the dense features x1..x4 and the block output, the trunk, the 4x tail."""
    import oracle
    ci = [0]

    def conv(x, act):
        c = net.conv(ci[0])
        ci[0] += 1
        return oracle.conv3x3(np.ascontiguousarray(x, dtype=np.float32), c["weight"], c["bias"], 2 if act else 0, 0.2)

    peaks = []
    fea = conv(tile, False)
    peaks.append(("conv_first", float(np.abs(fea).max())))
    x = fea
    for b in range(23):
        rrdb_in = x
        for j in range(3):
            feats = [x]
            for k in range(4):
                feats.append(conv(np.concatenate(feats, 0), True))
            x5 = conv(np.concatenate(feats, 0), False)
            pk = max(float(np.abs(f).max()) for f in feats[1:] + [x5])
            x = x5 * np.float32(0.2) + x
            peaks.append(("RRDB %d RDB %d" % (b + 1, j + 1), max(pk, float(np.abs(x).max()))))
        x = x * np.float32(0.2) + rrdb_in
    trunk = conv(x, False) + fea
    peaks.append(("trunk_conv + skip", float(np.abs(trunk).max())))
    up = conv(trunk.repeat(2, 1).repeat(2, 2), True)
    up = conv(up.repeat(2, 1).repeat(2, 2), True)
    peaks.append(("upconv1/2", float(np.abs(up).max())))
    hr = conv(up, True)
    out = conv(hr, False)
    peaks.append(("HRconv", float(np.abs(hr).max())))
    peaks.append(("conv_last (output)", float(np.abs(out).max())))
    assert ci[0] == net.num_convs
    return peaks, out


def gpu_checks(pp, bp, net, rep, bench):
    import realsr_ncnn_vulkan_amd as R
    from realsr_ncnn_vulkan_amd import synth
    ok = True
    sr = R.RealSR(0)
    sr.load(pp, bp)
    # pre-quantise, one 148 x 148 padded tile of the C1 frame.  Tolerances (one uint8 step is 1/255 = 3.92e-3):
    #   fp16 storage (the engine's default = the reference GPU path's own storage, realsr.cpp:44-46):  max <= 3.0e-3, p99.9 <= 1.5e-3
    #   precise mode (rsr_set_option precise 1):  max <= 2.6e-3 (a headroom of 1.5 on the +-1 bar); on the stand-ins it meets SURVEY 8(c)'s
    #   2e-3 / 5e-4 (tests/test_gpu_precise.py)
    # The harness measures fp16 storage first; when its headroom (one step / max error) is below 1.5 -- or a tolerance is missed -- it
    # switches the engine to precise mode, measures again, and runs the C1 and bench legs in the mode it RECOMMENDS for these weights.
    img = synth.make_image(1234, 256, 256)
    big = np.pad(img, ((10, 10), (10, 10), (0, 0)), mode="reflect")
    x = (big[:148, :148, :3].astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)).astype(np.float16)
    got = sr.net_forward(x).astype(np.float32)
    ref = net.forward(x.astype(np.float32))
    e = np.abs(got - ref)
    step = 1.0 / 255.0
    head16 = step / max(float(e.max()), 1e-12)
    rep["pre_quantise"] = {"max": float(e.max()), "p99_9": float(np.quantile(e, 0.999)), "nan_or_inf": bool(~np.isfinite(got).all()), "headroom": head16}
    print("  network output before quantisation, fp16 storage (default), engine vs oracle on one 148x148 tile ([0,1] units): max %.3e  p99.9 %.3e  "
          "(tolerance 3.0e-3 / 1.5e-3); headroom of the +-1 uint8 bar = one step (3.92e-3) / max = %.2f" % (e.max(), np.quantile(e, 0.999), head16))
    ok16 = bool(np.isfinite(got).all() and e.max() <= 3.0e-3 and np.quantile(e, 0.999) <= 1.5e-3)
    # Is that error the engine's arithmetic or the fp16 STORAGE format (which the reference's Vulkan path shares, realsr.cpp:44-46)?
    # A PyTorch-CPU emulation of fp16 storage / fp32 arithmetic on the same tile must deviate from the fp32 oracle by the same amount.
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch_ref
        wl = [(c["weight"], c["bias"]) for c in (net.conv(i) for i in range(net.num_convs))]
        emu = torch_ref.net_forward_fp16_storage_np(wl, x.astype(np.float32))
        ee, ex = np.abs(emu - ref), np.abs(got - emu)
        rep["fp16_storage_emulation"] = {"emulation_vs_oracle_mean": float(ee.mean()), "emulation_vs_oracle_p99_9": float(np.quantile(ee, 0.999)), "emulation_vs_oracle_max": float(ee.max()),
                                         "engine_vs_oracle_mean": float(e.mean()), "engine_vs_emulation_mean": float(ex.mean())}
        print("  fp16-storage emulation (PyTorch CPU) vs oracle on that tile: max %.3e  p99.9 %.3e  mean %.3e;  engine vs oracle mean %.3e;  engine vs emulation mean %.3e" % (
            ee.max(), np.quantile(ee, 0.999), ee.mean(), e.mean(), ex.mean()))
        if abs(e.mean() / max(ee.mean(), 1e-12) - 1) > 0.15:
            print("  FAIL: the engine's deviation from the oracle is not what fp16 storage alone explains (mean error differs from the emulation's by > 15 %)")
            ok = False
    except Exception as ex_:  # noqa: BLE001 -- a diagnostic, never the reason the harness dies
        print("  (fp16-storage emulation skipped: %r)" % (ex_,))
    rep["recommended_mode"] = "default"
    if not ok16 or head16 < 1.5:
        print("  fp16 storage leaves a headroom below 1.5 on these weights%s: switching to PRECISE mode (rsr_set_option precise 1; CLI: RSR_PRECISE=1)" % (
            "" if ok16 else " and misses the stated tolerance"))
        sr.set_option("precise", 1)
        got = sr.net_forward_f32(x)
        e = np.abs(got - ref)
        headp = step / max(float(e.max()), 1e-12)
        rep["pre_quantise_precise"] = {"max": float(e.max()), "p99_9": float(np.quantile(e, 0.999)), "nan_or_inf": bool(~np.isfinite(got).all()), "headroom": headp}
        print("  network output before quantisation, precise mode: max %.3e  p99.9 %.3e  (tolerance max 2.6e-3); headroom %.2f" % (e.max(), np.quantile(e, 0.999), headp))
        if not np.isfinite(got).all() or e.max() > 2.6e-3:
            print("  FAIL: pre-quantise error outside the stated tolerance in precise mode too -- fp16 operands cannot hold +-1 against the fp32 CPU path on these weights")
            ok = False
        else:
            rep["recommended_mode"] = "precise"
    print("  recommended mode for these weights: %s" % rep["recommended_mode"])
    # C1: whole frame, tile 128, +-1 uint8
    sr.tilesize = 128
    t = time.time()
    out = sr.process(img)
    t_gpu = time.time() - t
    t = time.time()
    want = net.process(img, 128)
    t_cpu = time.time() - t
    d = np.abs(out.astype(int) - want.astype(int))
    rep["c1"] = {"max_diff": int(d.max()), "frac_differ": float((d > 0).mean()), "frac_gt1": float((d > 1).mean()), "gpu_s_first_call": t_gpu, "oracle_s": t_cpu}
    print("  C1 (256x256, tile 128, 4 tiles) engine (%s mode) vs oracle: max |d| = %d, %.2f %% of the bytes differ, %.4f %% by more than 1   (oracle %.1f s)" % (
        rep["recommended_mode"], d.max(), 100 * (d > 0).mean(), 100 * (d > 1).mean(), t_cpu))
    if d.max() > 1:
        print("  FAIL: C1 outside +-1")
        ok = False
    if bench:
        import torch
        sr.tilesize = 200
        w, h = 1920, 1080
        frame = synth.make_image(1235, w, h)
        d_in = torch.from_numpy(frame).cuda()
        d_out = torch.empty((h * 4, w * 4, 3), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 5
        for _ in range(n):
            sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
        rep["c2_bench"] = {"ms_per_frame": dt * 1e3, "mpix_per_s": 33.1776 / dt, "frac_of_peak": 2544000 * 35853696 / dt / 2.5e15}
        print("  C2 (1920x1080, tile 200, device-resident, %d steps): %.2f ms/frame = %.1f Mpix/s = %.1f %% of 2.5 PFLOP/s" % (
            n, dt * 1e3, 33.1776 / dt, 100 * 2544000 * 35853696 / dt / 2.5e15))
    sr.close()
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir", nargs="?", default=None)
    ap.add_argument("--no-gpu", action="store_true", help="host-only checks (encoding, graph, pack, activation ranges)")
    ap.add_argument("--no-bench", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--range-tile", type=int, default=148, help="edge of the padded tile the fp32 activation-range walk runs on (148 = C1's tiles)")
    args = ap.parse_args()
    if not args.no_gpu:
        # PyTorch's bundled HIP runtime has to be the first one in the process (realsr-ncnn-vulkan_amd/__init__.py: lib()); an
        # inherited RSR_NO_TORCH=1 (the oracle worker pool sets it for its CPU-only children) would load /opt/rocm's copy first
        os.environ.pop("RSR_NO_TORCH", None)
        import torch  # noqa: F401
    d = find_model_dir(args.dir)
    if d is None:
        print("SKIP: no x4.param + x4.bin found%s (the reference checkout ships x4.param only: .MISSING_LARGE_BLOBS)" % (
            " in " + args.dir if args.dir else "; set RSR_REAL_MODELS or pass the directory"))
        return 0
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    print("model directory: %s" % d)
    rep = {"dir": d}
    ok, net = host_checks(pp, bp, rep)
    from realsr_ncnn_vulkan_amd import synth
    img = synth.make_image(1234, 256, 256)
    big = np.pad(img, ((10, 10), (10, 10), (0, 0)), mode="reflect")
    rt = max(16, min(int(args.range_tile), 276))
    tile = big[:rt, :rt, :3].astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
    t = time.time()
    peaks, walked = activation_ranges(net, tile)
    graph_err = float(np.abs(walked - net.forward(np.ascontiguousarray(tile))).max())  # the walk above against the oracle's own DAG interpreter
    rep["walk_vs_oracle_max"] = graph_err
    if graph_err > 1e-4:
        print("  FAIL: the layer-by-layer walk differs from the oracle's .param interpreter by %.3e" % graph_err)
        ok = False
    worst = max(peaks, key=lambda kv: kv[1])
    rep["activation_peaks"] = peaks
    print("  fp32 activation range over a %dx%d tile (%.1f s): largest |value| %.4g at %s; fp16 max is %.0f" % (rt, rt, time.time() - t, worst[1], worst[0], FP16_MAX))
    for name, v in peaks:
        if v > FP16_MAX / 16:
            print("    %-22s %.4g%s" % (name, v, "   <-- OVERFLOWS fp16 storage" if v > FP16_MAX else "   (within 16x of the fp16 limit)"))
    if worst[1] > FP16_MAX:
        print("  FAIL: an activation exceeds the fp16 range the engine (and the reference's Vulkan fp16-storage path) stores it in")
        ok = False
    if not args.no_gpu:
        import torch
        if not torch.cuda.is_available():
            print("  no GPU visible: GPU checks skipped (run on the MI355X box, or pass --no-gpu)")
        else:
            ok = gpu_checks(pp, bp, net, rep, not args.no_bench) and ok
    rep["ok"] = bool(ok)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rep, f, indent=1)
    print("RESULT: %s" % ("ok" if ok else "FAILED"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
