#!/usr/bin/env python3
"""bench.py -- RealSR x4 tiled inference on MI355X: BASELINE.json's metric on its config C2.

    python bench.py --gpus N --steps K --warmup W
        N = 1: runs in this process.  N > 1 without a torch.distributed environment: this script launches the N ranks
        itself (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...) and refuses to
        report anything but n_gpus == N.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          # the driver's own launch: WORLD_SIZE must equal N
    python bench.py --mode group --gpus N --frames F    # the PRODUCT's multi-GPU path in ONE process (config C4):
        rsr_create_group (weights by one RCCL broadcast) + jobs_proc threads per GPU on one shared frame queue
    python bench.py --mode group --gpus 1 --frames 64 --frame-size 256x256 --tile 128 --jobs-proc 16
        the reference's small-image mode ("-j 4:4:4 for many small images", README.md:61): concurrent calls are merged into one tile batch

A "step" = one pass of the hot path (preproc -> 351 fused convs -> postproc over all 60 tiles) over one synthetic
1920x1080 RGB frame per GPU.  One process per GPU; the only collective is the broadcast of the packed weights (RCCL) at
load.  Weak scaling: every rank upsamples its own frame each step; value = total output Mpix / max-rank time.

`value` is the HBM-resident rate (input and output images in device memory when the timed region starts -- the driver
contract: a PCIe-inclusive rate is never `value`).  SURVEY.md 8(d) defines the end-user metric as host memory -> host
memory: that run (rsr_process, the reference's RealSR::process boundary, H2D + network + D2H) is timed right behind it and
reported in `host_to_host` (pinned buffers = what the CLI uses; pageable = a caller that hands over malloc'd memory).

Prints ONE JSON line (rank 0).  The CPU oracle is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

W_IN, H_IN, TILE, PREPAD, SCALE = 1920, 1080, 200, 10, 4
FLOP_PER_PADDED_LR_PX = 35853696  # 2 x 17,926,848 MAC
PEAK_F16_TFLOPS = 2500.0  # gfx950 dense f16 MFMA peak, MI355X_MICROARCH.md
_LR = 2 * 9 * (3 * 64 + 69 * ((64 + 96 + 128 + 160) * 32 + 192 * 64) + 64 * 64)
_UP = 2 * 9 * (4 * 64 * 64 + 16 * 64 * 64 + 16 * 64 * 64 + 16 * 64 * 3)
assert _LR + _UP == FLOP_PER_PADDED_LR_PX
HBM_PEAK_TBPS = 8.0  # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured stream)
RIDGE_FLOP_PER_BYTE = PEAK_F16_TFLOPS / HBM_PEAK_TBPS  # 312.5


def executed_lr_px(tw, th):
    """Pixels of matrix work the LR-level convs execute for a padded tw x th tile: 32-pixel MFMA rows, 4-row wave granularity
    (rows below the tile are skipped), the last block column folded over two block rows when it is 1..14 pixels wide
    (kernels.h kFoldBit) -- a model of engine.cpp append_block_items + conv_flow.hip wave_is_dead; the PMC count of
    SQ_INSTS_VALU_MFMA_MOPS_F16 (profiles/r05_pmc_counters.txt) is the measurement."""
    rows = -(-th // 4) * 4
    rem = tw % 32
    if 1 <= rem <= 14:
        pairs = -(-th // 32)  # one folded block (two strips of 16 rows) per pair of block rows
        return (tw // 32) * 32 * rows + pairs * 16 * 32
    return -(-tw // 32) * 32 * rows


def executed_flop(w, h, T, P, tta=False):
    """FLOPs the kernels execute for a frame (model): LR-level convs on executed_lr_px, 2x / 4x-level convs x (1 - 0.139) (blocks that
    only feed cropped halo pixels are left out: 13.9 % at the 4x level, round 3)."""
    lr = 0
    for y0 in range(0, h, T):
        for x0 in range(0, w, T):
            tw, th = min(x0 + T, w) - x0 + 2 * P, min(y0 + T, h) - y0 + 2 * P
            lr += 4 * (executed_lr_px(tw, th) + executed_lr_px(th, tw)) if tta else executed_lr_px(tw, th)
    return lr * _LR + padded_px(w, h, T, P) * (8 if tta else 1) * _UP * (1 - 0.139)


DOMINANT_KERNEL = "conv3x3_flow<1, 1, false, 1, true, true>"


def padded_px(w, h, T, P=10):
    n = 0
    for y0 in range(0, h, T):
        for x0 in range(0, w, T):
            n += (min(x0 + T, w) - x0 + 2 * P) * (min(y0 + T, h) - y0 + 2 * P)
    return n


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the newest tracked rocprofv3 PMC summary (tools/gpu_round.sh writes
    it: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md).  None when no such file is there."""
    for name in ("r06_pmc_traffic.txt", "r05_pmc_traffic.txt", "r04_pmc_traffic.txt", "r03_pmc_traffic.txt", "r02_pmc_traffic.txt"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            txt = open(path).read()
        except OSError:
            continue
        m = re.search(r"^dominant\s+\S.*?bytes_per_launch=([0-9.eE+]+)", txt, re.M)
        if m:
            return float(m.group(1)), "profiles/" + name
    return None, None


def cpu_baseline(pp, bp, gpu_c1=None, gpu_c1_precise=None):
    """CPU restatement (NOT ncnn: the reference's -g -1 path cannot be built here) on bounded samples of the workload:
    (1) the oracle, 16 OpenMP threads, one padded 220x220 tile of the C2 frame; (2) BASELINE config C1 AS STATED -- the whole
    256x256 frame at tile 128 through the oracle with one thread (~37 s); (3) PyTorch-CPU
    (oneDNN) on the 220x220 tile, an independent construction of the same graph."""
    import numpy as np
    import torch
    import oracle
    from realsr_ncnn_vulkan_amd import synth
    net = oracle.OracleNet(pp, bp)
    img = synth.make_image(1234, 200, 200)
    net.process(synth.make_image(1, 24, 24), 200)  # warm the thread pool
    t = time.time()
    out = net.process(img, 200)
    dt = time.time() - t
    threads = oracle.max_threads()
    c2_px = padded_px(W_IN, H_IN, TILE, PREPAD)
    res = {"value": round(out.shape[0] * out.shape[1] / 1e6 / dt, 5), "unit": "Mpix/s", "cores": threads,
           "kind": "port", "host_cpus": os.cpu_count(),
           # the C2 frame on these cores, EXTRAPOLATED from the one tile by padded pixels (60 tiles = 52.56 x the sample): not measured
           "c2_extrapolated_ms": round(dt * 1e3 * c2_px / (220.0 * 220.0), 0), "c2_extrapolation": "tile seconds x 2,544,000 / 48,400 padded px",
           "sample": "oracle/realsr_oracle.c (CPU restatement of RealSR::process_cpu, not ncnn), one 200x200 "
                     "image at tile=200 = one padded 220x220 tile of the C2 frame, %.1f s, %.1f GFLOP/s" % (
                         dt, 220 * 220 * FLOP_PER_PADDED_LR_PX / dt / 1e9)}
    try:
        # BASELINE config C1 as stated: models-DF2K_JPEG (the seed-43 stand-in), 256x256, tile 128, CPU path, ONE thread
        # (realsr.cpp:525-838 with num_threads = 1, main.cpp:782) -- the actual frame (four 148x148 padded tiles, 3.141 TFLOP), not a sample
        dj = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K_JPEG", 43)
        netj = oracle.OracleNet(os.path.join(dj, "x4.param"), os.path.join(dj, "x4.bin"))
        oracle.set_threads(1)
        c1 = synth.make_image(1234, 256, 256)
        t = time.time()
        o1 = netj.process(c1, 128)
        d1 = time.time() - t
        res["c1_single_thread"] = {
            "value": round(o1.shape[0] * o1.shape[1] / 1e6 / d1, 6), "unit": "Mpix/s", "cores": 1, "seconds": round(d1, 2),
            "checksum": int(o1[::97, ::89].astype("int64").sum()),
            "gpu_c1_max_abs_diff_uint8": (int(np.abs(o1.astype(np.int16) - gpu_c1.astype(np.int16)).max()) if gpu_c1 is not None and gpu_c1.shape == o1.shape else None),
            "gpu_c1_bytes_differing_pct": (round(100.0 * float((o1 != gpu_c1).mean()), 3) if gpu_c1 is not None and gpu_c1.shape == o1.shape else None),
            # the same frame with rsr_set_option("precise", 1): the residual trunk keeps a byte of rounding residue per element
            "gpu_c1_precise_max_abs_diff_uint8": (int(np.abs(o1.astype(np.int16) - gpu_c1_precise.astype(np.int16)).max())
                                                  if gpu_c1_precise is not None and gpu_c1_precise.shape == o1.shape else None),
            "gpu_c1_precise_bytes_differing_pct": (round(100.0 * float((o1 != gpu_c1_precise).mean()), 3)
                                                   if gpu_c1_precise is not None and gpu_c1_precise.shape == o1.shape else None),
            "sample": "measured: the whole C1 frame (models-DF2K_JPEG stand-in, 256x256 -> 1024x1024, tile 128 = four 148x148 padded tiles, "
                      "3.141 TFLOP) through the oracle with 1 thread, %.1f s, %.1f GFLOP/s (CPU restatement, not ncnn)" % (
                          d1, 4 * 148 * 148 * FLOP_PER_PADDED_LR_PX / d1 / 1e9)}
    finally:
        oracle.set_threads(max(1, min(16, os.cpu_count() or 1)))
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch_ref
        weights = [(c["weight"], c["bias"]) for c in (net.conv(i) for i in range(net.num_convs))]
        x = np.random.default_rng(0).random((3, 220, 220), dtype=np.float32)
        torch_ref.net_forward_np(weights, x[:, :32, :32])  # warm up oneDNN
        t = time.time()
        torch_ref.net_forward_np(weights, x)
        d2 = time.time() - t
        res["torch_cpu"] = {"value": round(0.64 / d2, 5), "unit": "Mpix/s", "cores": torch.get_num_threads(),
                            "sample": "PyTorch %s CPU (oneDNN), fp32, one padded 220x220 tile (network only), %.2f s, %.1f GFLOP/s" % (
                                torch.__version__, d2, 220 * 220 * FLOP_PER_PADDED_LR_PX / d2 / 1e9)}
    except Exception as e:  # noqa: BLE001
        res["torch_cpu"] = {"value": None, "sample": "failed: %r" % (e,)}
    return res


def board_gemm_ceiling(dev):
    """The vendor fp16 GEMM (torch.mm -> hipBLASLt / rocBLAS) on N(0,1) and on all-zero operands, same board, same moment:
    what the 1,400 W cap leaves of the 2.5 PFLOP/s peak for full-entropy data -- the practical MFMA ceiling next to
    which roofline.frac should be read.  ~0.5 s of GPU time, outside the timed region."""
    import torch
    n, iters, out = 8192, 20, {}
    a = torch.empty((n, n), dtype=torch.float16, device=dev)
    b = torch.empty((n, n), dtype=torch.float16, device=dev)
    for fill in ("zeros", "randn"):
        if fill == "randn":
            a.normal_()
            b.normal_()
        else:
            a.zero_()
            b.zero_()
        for _ in range(3):
            torch.mm(a, b)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(iters):
            torch.mm(a, b)
        torch.cuda.synchronize()
        out[fill] = 2.0 * n ** 3 * iters / (time.time() - t) / 1e12
    return out


def other_config(name, dev, w, h, T, tta, img_seed, model, wseed, steps=3, keep=None, precise=False):
    """A further single-GPU BASELINE config, device-resident like `value`: C1 (256x256, tile 128: the GPU side of the config the
    reference runs on its CPU path), C3 (3840x2160, tile 400), C5 (1080p, -x TTA).  keep: dict that receives the output frame."""
    import torch
    import realsr_ncnn_vulkan_amd as R
    from realsr_ncnn_vulkan_amd import synth
    d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), model, wseed)
    sr = R.RealSR(dev.index or 0, tta_mode=tta)
    try:
        sr.load(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
        sr.tilesize, sr.prepadding, sr.scale = T, PREPAD, SCALE
        if precise:
            sr.set_option("precise", 1)
        img = synth.make_image(img_seed, w, h)
        d_in = torch.from_numpy(img).to(dev)
        d_out = torch.empty((h * SCALE, w * SCALE, 3), dtype=torch.uint8, device=dev)
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())  # warm-up: plan, workspace
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        fl = padded_px(w, h, T, PREPAD) * FLOP_PER_PADDED_LR_PX * (8 if tta else 1)
        res = {"config": "%s, %dx%d, tile %d%s%s, 1 GPU, device-resident" % (model, w, h, T, ", TTA x8 (-x)" if tta else "", ", precise mode" if precise else ""),
               "ms_per_frame": round(dt * 1e3, 2), "value": round(16.0 * w * h / 1e6 / dt, 2), "unit": "Mpix/s", "steps": steps,
               "frame_tflop": round(fl / 1e12, 3), "frac_of_peak": round(fl / dt / 1e12 / PEAK_F16_TFLOPS, 4),
               "frac_of_peak_executed": round(executed_flop(w, h, T, PREPAD, tta) / dt / 1e12 / PEAK_F16_TFLOPS, 4),
               "tile_batches": int(sr.get_stat("plan_batches")),
               # the engine's own count of 16 x 32 blocks per level (x 512 px x the level's FLOP per px): an upper bound of the executed
               # matrix work (MFMA waves still skip rows below the tile); the model above is held against it
               "frac_of_peak_blocks": round((sr.get_stat("plan_items_lr") * _LR + sr.get_stat("plan_items_2x") * 2 * 9 * 64 * 64 +
                                             sr.get_stat("plan_items_4x") * 2 * 9 * (2 * 64 * 64 + 64 * 3)) * 512 / dt / 1e12 / PEAK_F16_TFLOPS, 4),
               "checksum": int(d_out[::97, ::89].to(torch.int64).sum().item())}
        # one more, profiled, frame (outside the timed steps): the HBM-bound pre / post kernels against their algorithmic bytes
        # (pre 3 B in + 64 B out per padded px; post 6 B x slots in + 3 B out per output px).  Under TTA the
        # postproc kernel is the 8-way gather + merge (realsr_postproc_tta.comp); non-TTA RGB has none (fused into conv_last).
        sr.set_profiling(True)
        sr.get_profile(reset=True)
        sr.process_device(d_in.data_ptr(), w, h, 3, d_out.data_ptr())
        p = sr.get_profile(reset=True)
        sr.set_profiling(False)
        res["pre_ms"] = round(p["pre_ms"], 4)
        res["pre_GBps"] = round(p["pre_bytes"] / max(p["pre_ms"], 1e-9) / 1e6, 1)
        res["post_ms"] = round(p["post_ms"], 4)
        res["post_GBps"] = round(p["post_bytes"] / max(p["post_ms"], 1e-9) / 1e6, 1) if p["post_ms"] > 0 else None
        if keep is not None:
            keep[name] = d_out.cpu().numpy()
        return res
    finally:
        sr.close()


# ---- the product's own multi-GPU path (config C4): one process, rsr_create_group, shared frame queue ----------------------
def group_mode(args):
    import numpy as np
    import torch
    import realsr_ncnn_vulkan_amd as R
    from realsr_ncnn_vulkan_amd import synth
    n = args.gpus
    same = os.environ.get("RSR_BENCH_SAME_GPU") == "1"  # 1-GPU test hook: every "GPU" of the group is a context on device 0
    have = torch.cuda.device_count()
    if not same and have < n:
        raise SystemExit("bench.py --mode group --gpus %d: only %d device(s) visible" % (n, have))
    d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    t0 = time.perf_counter()
    if same:
        srs, transport = [], "host (RSR_BENCH_SAME_GPU: %d contexts on device 0)" % n
        for _ in range(n):
            s = R.RealSR(0)
            s.load(pp, bp)
            s.set_option("max_workspace_mb", 24 * 1024)
            srs.append(s)
    else:
        srs, transport = R.create_group(list(range(n)), pp, bp)
    load_s = time.perf_counter() - t0
    GW, GH = (int(v) for v in args.frame_size.lower().split("x"))
    GT = args.tile
    for s in srs:
        s.tilesize, s.prepadding, s.scale = GT, PREPAD, SCALE
        s._push_params()
        if args.merge is not None:
            s.set_option("merge", args.merge)
        s.set_option("max_lanes", max(16, args.jobs_proc))
    frames = args.frames
    jobs = args.jobs_proc
    src = [synth.make_image(1235 + i, GW, GH) for i in range(min(frames, 4))]
    pin_in = []
    for im in src:
        p = R.PinnedArray(im.shape)
        p.array[:] = im
        pin_in.append(p)
    outs = [[R.PinnedArray((GH * SCALE, GW * SCALE, 3)) for _ in range(jobs)] for _ in srs]
    for gi, s in enumerate(srs):  # warm-up: plans, workspaces
        s.process(pin_in[0].array, out=outs[gi][0].array, push_params=False)
    # ... and the lanes of every proc thread (stream, events, device buffers, pinned staging are created on a thread's first call)
    wth = [threading.Thread(target=lambda gi=gi, ji=ji: srs[gi].process(pin_in[0].array, out=outs[gi][ji].array, push_params=False))
           for gi in range(len(srs)) for ji in range(jobs)]
    for t in wth:
        t.start()
    for t in wth:
        t.join()
    merged0 = [(s.get_stat("merged_batches"), s.get_stat("merged_images")) for s in srs]
    lock = threading.Lock()
    nxt = [0]
    done = [[0] * jobs for _ in srs]
    sums = {}

    def proc(gi, ji):  # main.cpp:311-331 -- every proc thread of every GPU pops the one shared queue
        while True:
            with lock:
                i = nxt[0]
                if i >= frames:
                    return
                nxt[0] = i + 1
            srs[gi].process(pin_in[i % len(pin_in)].array, out=outs[gi][ji].array, push_params=False)
            done[gi][ji] += 1
            if i < len(pin_in):
                sums[i] = int(outs[gi][ji].array[::97, ::89].astype(np.int64).sum())

    ths = [threading.Thread(target=proc, args=(gi, ji)) for gi in range(len(srs)) for ji in range(jobs)]
    t1 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t1
    # one large image over all contexts, tiles dealt by rsr_process_group
    big_out = R.PinnedArray((GH * SCALE, GW * SCALE, 3))
    R.process_group(srs, pin_in[0].array, out=big_out.array)
    t2 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        R.process_group(srs, pin_in[0].array, out=big_out.array)
    dt_one = (time.perf_counter() - t2) / reps
    one_ok = int(big_out.array[::97, ::89].astype(np.int64).sum()) == sums.get(0)
    out_mpix = GW * SCALE * GH * SCALE / 1e6
    mb = sum(s.get_stat("merged_batches") - m[0] for s, m in zip(srs, merged0))
    mi = sum(s.get_stat("merged_images") - m[1] for s, m in zip(srs, merged0))
    res = {"metric": "output Mpix/s (4x upscale) DF2K tile=%d" % GT, "mode": "group", "value": round(out_mpix * frames / dt, 3), "unit": "Mpix/s",
           "n_gpus": n, "frames": frames, "seconds": round(dt, 3), "higher_is_better": True, "scaling": "strong", "dtype": "f16", "data": "synthetic",
           # small images of concurrent calls walk the network as one tile batch (engine.h): how many did
           "merged": {"images": int(mi), "batches": int(mb), "images_per_batch": round(mi / mb, 2) if mb else None,
                      "merge_option": args.merge if args.merge is not None else 16},
           "config": {"workload": "%s%d x (%dx%d -> %dx%d) frames, models-DF2K, tile %d, host memory -> host memory (pinned), "
                                  "%d GPUs x jobs_proc %d threads on one shared queue (main.cpp:811-828)" % (
                                      "C4: " if (GW, GH, GT) == (1920, 1080, 200) else "", frames, GW, GH, GW * SCALE, GH * SCALE, GT, n, jobs),
                      "parallelism": "one process, rsr_create_group: weights by %s; frames from a shared queue, no data-path collective" % transport,
                      "group_transport": transport, "group_members": len(srs), "visible_devices": have, "load_seconds": round(load_s, 3),
                      "frames_per_gpu": [sum(x) for x in done]},
           "single_image_over_group": {"what": "ONE frame, its tiles dealt over the %d contexts by rsr_process_group (host -> host)" % n,
                                       "ms": round(dt_one * 1e3, 2), "value": round(out_mpix / dt_one, 2), "unit": "Mpix/s",
                                       "bytes_equal_single_context": bool(one_ok)}}
    print(json.dumps(res), flush=True)
    for p in pin_in + [big_out] + [o for row in outs for o in row]:
        p.free()
    for s in srs:
        s.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=("ranks", "group"), default="ranks")
    ap.add_argument("--frames", type=int, default=64, help="group mode: frames in the queue (C4: 64)")
    ap.add_argument("--jobs-proc", type=int, default=2, help="group mode: proc threads per GPU (the reference's default, main.cpp:708-711)")
    ap.add_argument("--frame-size", default="%dx%d" % (W_IN, H_IN), help="group mode: WxH of the frames (default the C2 / C4 frame)")
    ap.add_argument("--tile", type=int, default=TILE, help="group mode: tile size")
    ap.add_argument("--merge", type=int, default=None, help="group mode: rsr_set_option merge (1 = small images of concurrent calls are not merged)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-host", action="store_true", help="skip the host->host leg")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C5 legs")
    ap.add_argument("--no-group", action="store_true", help="N > 1: skip the product-path (rsr_create_group) leg behind the rank run")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    if args.mode == "group":
        return group_mode(args)

    # ---- N ranks requested but no torch.distributed environment: launch them ourselves (one process per GPU) ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            sys.stdout.write(r.stdout)
            raise SystemExit("bench.py: the %d-rank run failed (rc %d)" % (args.gpus, r.returncode))
        j = json.loads(lines[-1])
        if j.get("n_gpus") != args.gpus:
            raise SystemExit("bench.py: asked for %d GPUs, the run reports n_gpus=%r" % (args.gpus, j.get("n_gpus")))
        print(lines[-1], flush=True)
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    import realsr_ncnn_vulkan_amd as R
    from realsr_ncnn_vulkan_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or without torch.distributed.run: "
                         "this script then starts the ranks itself)" % (args.gpus, world, args.gpus))
    # Test hook for a 1-GPU box: RSR_BENCH_SAME_GPU=1 puts every rank on cuda:0 and RSR_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device), so the multi-rank control flow can be exercised without 2 GPUs.
    same_gpu = os.environ.get("RSR_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local = 0
    backend = os.environ.get("RSR_BENCH_BACKEND", "gloo" if same_gpu else "nccl")
    if not same_gpu and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but only %d device(s) visible" % (world, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    models_root = os.environ.get("RSR_MODELS", "/tmp/rsr_models")
    d = synth.make_model_dir(models_root, "models-DF2K", 42) if rank == 0 else None
    if world > 1:
        dist.barrier()
        d = synth.make_model_dir(models_root, "models-DF2K", 42)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")

    # ---- weights: rank 0 parses + packs once, ONE broadcast over xGMI, every rank loads the blob ----
    def bcast(t):
        if backend == "nccl":
            dist.broadcast(t, 0)
        else:  # gloo test hook: through host memory
            h = t.cpu()
            dist.broadcast(h, 0)
            t.copy_(h)

    if rank == 0:
        blob = torch.from_numpy(R.model_pack(pp, bp)).to(dev)
        n = torch.tensor([blob.numel()], dtype=torch.int64, device=dev)
    else:
        n = torch.zeros(1, dtype=torch.int64, device=dev)
    if world > 1:
        bcast(n)
        if rank != 0:
            blob = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
        bcast(blob)
    torch.cuda.synchronize()
    blob_bytes = int(blob.numel())
    sr = R.RealSR(local)
    sr.load_packed(blob.numel(), device_ptr=blob.data_ptr())
    sr.tilesize, sr.prepadding, sr.scale = TILE, PREPAD, SCALE
    if same_gpu and world > 1:
        sr.set_option("max_workspace_mb", 24 * 1024)  # several contexts share one GPU in the test hook

    img = synth.make_image(1235 + rank, W_IN, H_IN)  # image seed 1234 + cfg
    d_in = torch.from_numpy(img).to(dev)
    d_out = torch.empty((H_IN * SCALE, W_IN * SCALE, 3), dtype=torch.uint8, device=dev)

    def step():
        sr.process_device(d_in.data_ptr(), W_IN, H_IN, 3, d_out.data_ptr())  # synchronous

    def timed_steps():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; seconds of this rank."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        step()
    # Pass 1 -- the product as it runs: no instrumentation at all.  `value` and `ms_per_step` come from here.
    wall0 = time.time()
    dt = timed_steps()
    if rank == 0 and os.environ.get("RSR_BENCH_MARKS"):  # tools/power_sampler.py: power / clock over exactly the timed region
        with open(os.environ["RSR_BENCH_MARKS"], "w") as f:
            f.write("%.6f %.6f\n" % (wall0, time.time()))
    # Pass 2 -- the same steps again with hipEvents around every kernel launch on the engine's launch stream (~1 us per launch,
    # 352 launches per frame): the per-kernel breakdown of the roofline block, and `ms_per_step_profiled` next to `ms_per_step`.
    prof = conv_ms = None
    dt_prof = None
    if not args.no_profile:
        sr.set_profiling(True)
        sr.get_profile(reset=True)
        sr.get_conv_times(reset=True)
        dt_prof = timed_steps()
        prof = sr.get_profile(reset=True)
        conv_ms = sr.get_conv_times(reset=True)
        sr.set_profiling(False)

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = max_over_ranks(dt)
    if dt_prof is not None:
        dt_prof = max_over_ranks(dt_prof)
    checksum = int(d_out[::97, ::89].to(torch.int64).sum().item())

    # ---- host memory -> host memory: rsr_process incl. H2D / D2H, same frame, same step count ----
    host = None
    if not args.no_host:
        host = {}
        pin_in, pin_out = R.PinnedArray(img.shape), R.PinnedArray((H_IN * SCALE, W_IN * SCALE, 3))
        pin_in.array[:] = img
        page_out = np.empty((H_IN * SCALE, W_IN * SCALE, 3), dtype=np.uint8)
        for name, src, dst in (("pinned", pin_in.array, pin_out.array), ("pageable", img, page_out)):
            sr.process(src, out=dst)
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                sr.process(src, out=dst, push_params=False)
            host[name] = max_over_ranks(time.perf_counter() - t1)
        host["identical_to_device_path"] = bool((torch.from_numpy(pin_out.array[::97, ::89].copy()).to(torch.int64).sum().item()) == checksum)
        # two caller threads on the one context (the reference's jobs_proc = 2, main.cpp:811-828): upload of frame k+1 and
        # download of frame k-1 ride under the kernels of frame k
        if world == 1:
            pin_out2 = R.PinnedArray((H_IN * SCALE, W_IN * SCALE, 3))

            def worker(dst):
                for _ in range(args.steps):
                    sr.process(pin_in.array, out=dst, push_params=False)

            ths = [threading.Thread(target=worker, args=(o.array,)) for o in (pin_out, pin_out2)]
            t1 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            host["pinned_2threads"] = (time.perf_counter() - t1) / 2
            pin_out2.free()
        pin_in.free()
        pin_out.free()

    res = None
    if rank == 0:
        out_mpix = W_IN * SCALE * H_IN * SCALE / 1e6
        ppx = padded_px(W_IN, H_IN, TILE, PREPAD)
        res = {
            "metric": "output Mpix/s (4x upscale) DF2K tile=200",
            "value": round(out_mpix * world * args.steps / dt, 3),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "ms_per_step_profiled": round(dt_prof / args.steps * 1e3, 3) if dt_prof is not None else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "value_definition": "HBM-resident images (rsr_process_device; the bench contract); SURVEY 8(d)'s metric, host memory -> host memory, is `host_to_host_pinned`",
            "device_resident": round(out_mpix * world * args.steps / dt, 3),
            "host_to_host_pinned": round(out_mpix * world * args.steps / host["pinned"], 3) if host else None,
            "config": {
                "workload": "C2: models-DF2K, 1920x1080 RGB -> 7680x4320, scale=4, tile=200, prepadding=10, "
                            "60 tiles/frame (2,544,000 padded LR px, 91.21 TFLOP algorithmic), 1 frame per GPU per step",
                "weights": "synthetic seeded fp16-tagged x4.bin (real blobs absent from the reference checkout)",
                "io": "uint8 HWC in HBM -> uint8 HWC in HBM (rsr_process_device)",
                "ranks_seen": {"torch_distributed_world_size": (dist.get_world_size() if world > 1 else 1), "backend": backend if world > 1 else None,
                               "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if world > 1 and backend == "nccl" else None),
                               "visible_devices": torch.cuda.device_count()},
                "parallelism": ("%d ranks (torch.distributed %s, one process per GPU), frames sharded 1/GPU/step, weights by one broadcast of %.1f MB, "
                                "no data-path collective" % (world, "nccl = RCCL" if backend == "nccl" else backend, blob_bytes / 1e6)) if world > 1 else "single GPU",
                "frame_tflop": round(ppx * FLOP_PER_PADDED_LR_PX / 1e12, 2),
                "whole_path_tflops": round(ppx * FLOP_PER_PADDED_LR_PX * world * args.steps / dt / 1e12, 1),
                "whole_path_frac_of_peak": round(ppx * FLOP_PER_PADDED_LR_PX * world * args.steps / dt / 1e12 / PEAK_F16_TFLOPS / world, 4),
                # the same frame executes fewer FLOPs than the algorithmic count: -13.9 % of the 4x-level blocks (dead-output elimination),
                # +1.9 % at the LR level (32-pixel MFMA rows: 220 -> 224 columns; the 140-wide edge tiles' last 12 columns run as folded
                # blocks since round 5: 144 instead of 160) -- executed_flop() is the model, PMC SQ_INSTS_VALU_MFMA_MOPS_F16 the measurement
                "whole_path_frac_of_peak_executed": round(executed_flop(W_IN, H_IN, TILE, PREPAD) * args.steps / dt / 1e12 / PEAK_F16_TFLOPS, 4),
                "frac_of_peak_caveat": "one binary measured 88.7-93.9 ms on five boards of round 3, the round-4 binary 87.5-93.8 ms on ten, the round-5 code 86.1-90.7 ms on four "
                                       "(41.2-42.4 % whole path in its three bench lines): the boards differ in the clock their power management grants under the 1,400 W cap "
                                       "(1,531-1,656 MHz seen); a single run is one board's number, not a floor",
                "checksum": checksum,
            },
        }
        if host:
            res["host_to_host"] = {
                "what": "rsr_process: uint8 in host memory -> uint8 in host memory incl. H2D 6.2 MB + D2H 99.5 MB per frame (SURVEY 8(d))",
                "pinned": {"value": round(out_mpix * world * args.steps / host["pinned"], 3), "ms_per_step": round(host["pinned"] / args.steps * 1e3, 3),
                           "buffers": "rsr_host_alloc (what the CLI allocates)"},
                "pageable": {"value": round(out_mpix * world * args.steps / host["pageable"], 3), "ms_per_step": round(host["pageable"] / args.steps * 1e3, 3),
                             "buffers": "malloc'd: staged through the call's pinned lane, download in 16 MB chunks"},
                "unit": "Mpix/s", "bytes_identical_to_device_path": host["identical_to_device_path"],
            }
            if "pinned_2threads" in host:
                res["host_to_host"]["pinned_2_caller_threads"] = {
                    "value": round(out_mpix * args.steps / host["pinned_2threads"], 3), "ms_per_step": round(host["pinned_2threads"] / args.steps * 1e3, 3),
                    "what": "two threads calling rsr_process on the one context (jobs_proc = 2): transfers of neighbouring frames overlap the kernels"}
        if prof and prof["conv_ms"] > 0:
            # Dominant kernel class: the 276 dense-block convs cin in {64,96,128,160} -> 32 (conv indices 1+5j+k, k<4),
            # ~50 % of the frame, all launches of ONE kernel: rsr::conv3x3_flow<1,1,false,1,true,true>.  Their launches are bracketed
            # by hipEvents on the launch stream inside the engine during the PROFILED pass of the same steps (ms_per_step_profiled).
            steps = args.steps
            batches = max(1, round(prof["conv_launches"] / (351.0 * max(prof["calls"], 1))))  # tile batches per frame (1 for C2)
            plane_bytes = ppx * 32.0  # one 16-channel fp16 plane of all 60 tiles: 81.4 MB
            classes = {}
            for k in range(4):
                cin = 64 + 32 * k
                ms = float(sum(conv_ms[1 + 5 * j + k] for j in range(69)))
                n = 69 * steps * batches
                fl = 2.0 * 9 * cin * 32 * ppx  # per launch
                by = (cin / 16 + 2) * plane_bytes  # cin / 16 planes read + 2 written, each exactly once (algorithmic)
                us = ms * 1e3 / n
                classes["%d->32" % cin] = {
                    "launches": n, "avg_launch_us": round(us, 2), "flop_per_launch": round(fl), "bytes_per_launch": round(by),
                    "intensity_flop_per_byte": round(fl / by, 1), "mfma_frac": round(fl / (us * 1e-6) / 1e12 / PEAK_F16_TFLOPS, 4),
                    "hbm_TBps": round(by / (us * 1e-6) / 1e12, 3), "hbm_frac": round(by / (us * 1e-6) / 1e12 / HBM_PEAK_TBPS, 4),
                    "bound": "hbm" if fl / by < RIDGE_FLOP_PER_BYTE else "mfma"}
            ring_idx = [1 + 5 * j + k for j in range(69) for k in range(4)]
            ring_ms = float(sum(conv_ms[i] for i in ring_idx))
            ring_launches = len(ring_idx) * steps * batches
            ring_flops = sum(2.0 * 9 * (64 + 32 * k) * 32 for k in range(4)) * 69 * ppx * steps
            ring_bytes_alg = sum((4 + 2 * k) + 2 for k in range(4)) / 4.0 * plane_bytes  # average per launch: 732.7 MB
            avg_us = ring_ms * 1e3 / max(ring_launches, 1)
            ach = ring_flops / (ring_ms * 1e-3) / 1e12
            ach_all = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
            c5_idx = [5 + 5 * j for j in range(69)]
            c5_ms = float(sum(conv_ms[i] for i in c5_idx))
            c5_flops = 2.0 * 9 * 192 * 64 * 69 * ppx * steps
            c5_us = c5_ms * 1e3 / (69 * steps * batches)
            c5_bytes = (12 + 4) * plane_bytes
            tail_ms = float(sum(conv_ms[i] for i in (347, 348, 349, 350)))
            traffic, traffic_src = pmc_traffic()
            hbm_bytes = traffic if traffic else ring_bytes_alg
            intensity = (ring_flops / max(ring_launches, 1)) / hbm_bytes
            res["roofline"] = {
                # BASELINE.json's second metric is "achieved MFMA % of roofline": achieved / peak / frac are that number.  What BINDS
                # this kernel is the other roof: its intensity lies below the ridge, see `bound`, `hbm_frac` and `classes`.
                "bound": "hbm" if intensity < RIDGE_FLOP_PER_BYTE else "mfma",
                "achieved": round(ach, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F16_TFLOPS, 4),
                "metric_roof": "mfma (BASELINE.json: 'achieved MFMA % of roofline'); binding roof: see `bound`",
                "intensity_flop_per_byte": round(intensity, 1), "ridge": round(RIDGE_FLOP_PER_BYTE, 1),
                "hbm_achieved_TBps": round(hbm_bytes / (avg_us * 1e-6) / 1e12, 3), "hbm_peak_TBps": HBM_PEAK_TBPS,
                "hbm_frac": round(hbm_bytes / (avg_us * 1e-6) / 1e12 / HBM_PEAK_TBPS, 4),
                "hbm_frac_note": "of the 8 TB/s spec; the guide's measured stream rate is 6.29 TB/s, a 3:1 read:write mix of 1-KiB pieces like this kernel's "
                                 "reaches 5.4 TB/s on this chip (tools/ubench/ldsdma_rate.hip, profiles/r03_ldsdma_rate.txt)",
                "traffic": traffic,
                "traffic_source": traffic_src or "no tracked PMC summary found",
                "traffic_unit": "B/launch, HBM (PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes); algorithmic (3.5 + 1) x 162.8 MB = 732.7e6",
                "kernel": "rsr::" + DOMINANT_KERNEL + " (276 of the 351 convs: cin 64..160 -> 32, LeakyReLU, 16-channel fp16 planes, weights LDS-resident)",
                "launches": ring_launches,
                "avg_launch_us": round(avg_us, 2),
                "algorithmic_flop_per_launch_avg": round(ring_flops / max(ring_launches, 1)),
                "algorithmic_bytes_per_launch_avg": round(ring_bytes_alg),
                "classes": classes,
                "flop_accounting": "algorithmic = SURVEY 8(d): every padded-tile pixel of every layer; the kernel executes ~1.9 % more (32-pixel MFMA "
                                   "rows; narrow last block columns run folded) and the 2x / 4x convs ~14 % less (blocks that only feed cropped halo pixels are not computed)",
                "second_kernel": {"kernel": "rsr::conv3x3_flow<2, 1, false, 2, false, false> (69 x conv5 192 -> 64)",
                                  "achieved": round(c5_flops / (c5_ms * 1e-3) / 1e12, 1), "frac": round(c5_flops / (c5_ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                                  "avg_launch_us": round(c5_us, 2),
                                  "intensity_flop_per_byte": round(2.0 * 9 * 192 * 64 * ppx / c5_bytes, 1),
                                  "hbm_frac": round(c5_bytes / (c5_us * 1e-6) / 1e12 / HBM_PEAK_TBPS, 4),
                                  "bound": "mfma under the board's power cap (intensity above the ridge; 86 % of the vendor GEMM on the same board)"},
                "all_convs": {"achieved": round(ach_all, 1), "frac": round(ach_all / PEAK_F16_TFLOPS, 4), "launches": prof["conv_launches"],
                              "conv_ms_per_step": round(prof["conv_ms"] / steps, 3)},
                "tail_ms_per_step": round(tail_ms / steps, 3),
                "pre_ms_per_step": round(prof["pre_ms"] / steps, 4),
                "post_ms_per_step": round(prof["post_ms"] / steps, 4),
                "post_GBps": round(prof["post_bytes"] / max(prof["post_ms"], 1e-9) / 1e6, 1),
                "timing": "hipEvents on the launch stream around every kernel, in a SECOND pass of the same %d steps (ms_per_step_profiled); "
                          "`value` / `ms_per_step` come from the first, un-instrumented pass" % steps,
                "consistency": {"dominant_kernel_ms_per_step": round(ring_ms / steps, 3), "ms_per_step_profiled": round(dt_prof / steps * 1e3, 3),
                                "holds": bool(ring_ms / steps <= dt_prof / steps * 1e3)},
                "note": "the board sits at its 1400 W cap during this workload (profiles/r05_power_clock.txt: 1400 W mean, sclk ~1.63 GHz of 2.4 over the timed region): "
                        "at that clock the MFMA peak is ~1.7 PFLOP/s; fed entirely from the L2 the same kernels reach 45-51 % of 2.5 PF (profiles/r04_l2_bound.txt)",
            }
            try:
                g = board_gemm_ceiling(dev)
                res["roofline"]["vendor_gemm_same_board"] = {
                    "what": "torch.mm fp16 8192^3 (hipBLASLt/rocBLAS), measured right after the timed region",
                    "randn_tflops": round(g["randn"], 1), "zeros_tflops": round(g["zeros"], 1),
                    "achieved_over_randn_gemm": round(ach / g["randn"], 4), "all_convs_over_randn_gemm": round(ach_all / g["randn"], 4)}
            except Exception as e:  # noqa: BLE001 -- context only, never fail the bench on it
                res["roofline"]["vendor_gemm_same_board"] = {"what": "failed: %r" % (e,)}
    sr.close()
    del d_in, d_out, blob
    torch.cuda.empty_cache()
    kept = {}
    if rank == 0 and world == 1 and not same_gpu and not args.no_other_configs:
        # the other single-GPU BASELINE configs, same definition as `value`.  C1 is the config the reference runs on its CPU path
        # (-g -1, one thread): its GPU side here, its CPU side in cpu_baseline.c1_single_thread; C4 needs 8 GPUs.
        res["other_configs"] = {}
        for name, cfg, steps in (("C1", (256, 256, 128, False, 1234, "models-DF2K_JPEG", 43), 20),
                                 ("C3", (3840, 2160, 400, False, 1236, "models-DF2K", 42), 3),
                                 ("C5", (1920, 1080, 200, True, 1239, "models-DF2K_JPEG", 43), 3)):
            try:
                res["other_configs"][name] = other_config(name, dev, *cfg, steps=steps, keep=kept if name == "C1" else None)
            except Exception as e:  # noqa: BLE001 -- never lose the C2 line over a side leg
                res["other_configs"][name] = {"value": None, "error": repr(e)}
        # PRECISE mode (rsr_set_option precise 1: the residual trunk keeps a byte of rounding residue per element, conv_last's fp32 result goes to
        # uint8 unrounded): C2 and C1 again -- what half the distance to the fp32 CPU path costs
        res["precise_mode"] = {"what": "the same engine with option precise = 1 (default 0 = fp16 storage, the reference GPU path's own, realsr.cpp:44-46)"}
        for name, cfg, steps in (("C2", (W_IN, H_IN, TILE, False, 1235, "models-DF2K", 42), args.steps),
                                 ("C1", (256, 256, 128, False, 1234, "models-DF2K_JPEG", 43), 20)):
            try:
                res["precise_mode"][name] = other_config(name + "p", dev, *cfg, steps=steps, keep=kept if name == "C1" else None, precise=True)
            except Exception as e:  # noqa: BLE001
                res["precise_mode"][name] = {"value": None, "error": repr(e)}
        try:
            res["precise_mode"]["c2_cost_pct"] = round(100.0 * (res["precise_mode"]["C2"]["ms_per_frame"] / res["ms_per_step"] - 1.0), 2)
        except Exception:  # noqa: BLE001
            pass
        res["other_configs"]["C4"] = {"value": None, "status": "unmeasured: n_gpus = 1",
                                      "config": "64 x C2 frames over 8 GPUs, -j 4:8:4 (one process, rsr_create_group, proc threads on one shared queue)",
                                      "n1_anchor": "group_mode below: the same pipeline with ONE GPU, 16 frames, jobs_proc 2, host -> host"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(pp, bp, gpu_c1=kept.get("C1"), gpu_c1_precise=kept.get("C1p"))
        except Exception as e:  # the oracle is optional here; never fail the GPU number on it
            res["cpu_baseline"] = {"value": None, "unit": "Mpix/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0 and world == 1 and not same_gpu and not args.no_group:
        # The product's own multi-GPU pipeline (config C4: rsr_create_group + jobs_proc threads per GPU on one shared frame queue,
        # host memory -> host memory) with ONE GPU: the N = 1 anchor of a future scaling curve.  Child process with a deadline.
        cmd = [sys.executable, os.path.abspath(__file__), "--mode", "group", "--gpus", "1", "--frames", "16", "--jobs-proc", "2"]
        try:
            # RSR_GROUP_FORCE_RCCL: take the RCCL branch of rsr_create_group (ncclCommInitAll + ncclBroadcast) also for one member
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240, env=dict(os.environ, RSR_GROUP_FORCE_RCCL="1"))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            res["group_mode"] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"value": None, "error": "rc %d: %s" % (r.returncode, r.stderr[-600:])}
        except Exception as e:  # noqa: BLE001
            res["group_mode"] = {"value": None, "error": repr(e)}
        # The reference's small-image mode (README.md:61 "-j 4:4:4 for many small images"; main.cpp:811-828): 64 frames of 256 x 256 at tile 128
        # (the C1 geometry) from 16 proc threads on one context, host -> host.  Concurrent calls are merged into one tile batch; the same
        # run with merging off (the calls queue up on the one compute stream) beside it.
        res["small_images"] = {}
        for key, merge in (("merged", "16"), ("not_merged", "1")):
            cmd = [sys.executable, os.path.abspath(__file__), "--mode", "group", "--gpus", "1", "--frames", "64", "--frame-size", "256x256",
                   "--tile", "128", "--jobs-proc", "16", "--merge", merge]
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                j = json.loads(lines[-1]) if (r.returncode == 0 and lines) else None
                res["small_images"][key] = ({"value": j["value"], "unit": "Mpix/s", "frames": j["frames"], "seconds": j["seconds"], "merged": j["merged"],
                                             "workload": j["config"]["workload"]} if j else {"value": None, "error": "rc %d: %s" % (r.returncode, r.stderr[-400:])})
            except Exception as e:  # noqa: BLE001
                res["small_images"][key] = {"value": None, "error": repr(e)}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and world > 1 and not args.no_group:
        # The product's own multi-GPU path (rsr_create_group: ncclCommInitAll + one ncclBroadcast inside ONE process, proc threads
        # on a shared queue = config C4) in a child process with a deadline, after every rank has released its GPU: a first-contact
        # failure there must not cost the rank-mode line.
        cmd = [sys.executable, os.path.abspath(__file__), "--mode", "group", "--gpus", str(world), "--frames", str(max(8 * world, 16))]
        try:
            time.sleep(2.0)  # the other ranks are tearing down their contexts
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=180)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            res["group_mode"] = json.loads(lines[-1]) if (r.returncode == 0 and lines) else {"value": None, "error": "rc %d: %s" % (r.returncode, r.stderr[-600:])}
        except Exception as e:  # noqa: BLE001
            res["group_mode"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
