#!/usr/bin/env python3
"""bench.py -- RealSR x4 tiled inference on MI355X: BASELINE.json's metric on its config C2.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (preproc -> 351 fused convs -> postproc over all 60 tiles) over one synthetic
1920x1080 RGB frame per GPU.  One process per GPU; the only collective is the broadcast of the packed weights (RCCL) at
load.  Weak scaling: every rank upsamples its own frame each step; value = total output Mpix / max-rank time.

`value` is the HBM-resident rate (input and output images in device memory when the timed region starts -- the driver
contract).  SURVEY.md 8(d) defines the end-user metric as host memory -> host memory: that run (rsr_process, the
reference's RealSR::process boundary, H2D + network + D2H) is timed right behind it and reported in `host_to_host`
(pinned buffers = what the CLI uses; pageable = a caller that hands over malloc'd memory).

Prints ONE JSON line (rank 0).  The CPU oracle is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

W_IN, H_IN, TILE, PREPAD, SCALE = 1920, 1080, 200, 10, 4
FLOP_PER_PADDED_LR_PX = 35853696  # 2 x 17,926,848 MAC
PEAK_F16_TFLOPS = 2500.0  # gfx950 dense f16 MFMA peak, MI355X_MICROARCH.md
DOMINANT_KERNEL = "conv3x3_flow<1, 1, false, 1, true>"


def padded_px(w, h, T, P):
    n = 0
    for y0 in range(0, h, T):
        for x0 in range(0, w, T):
            n += (min(x0 + T, w) - x0 + 2 * P) * (min(y0 + T, h) - y0 + 2 * P)
    return n


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the tracked rocprofv3 PMC summary (tools/gpu_round.sh writes it:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md).  None when the file is not there."""
    path = os.path.join(ROOT, "profiles", "r02_pmc_traffic.txt")
    try:
        txt = open(path).read()
    except OSError:
        return None, None
    m = re.search(r"^dominant\s+\S.*?bytes_per_launch=([0-9.eE+]+)", txt, re.M)
    return (float(m.group(1)), "profiles/r02_pmc_traffic.txt") if m else (None, None)


def cpu_baseline(pp, bp):
    """CPU restatement (NOT ncnn: the reference's -g -1 path cannot be built here) on bounded samples of the workload:
    (1) the oracle, 16 OpenMP threads, one padded 220x220 tile of the C2 frame; (2) the oracle single-threaded on one
    84x84 padded tile (C1's code path; C1's four 148x148 tiles are 12.4x that); (3) PyTorch-CPU
    (oneDNN) on the 220x220 tile, an independent construction of the same graph."""
    import oracle
    net = oracle.OracleNet(pp, bp)
    img = synth.make_image(1234, 200, 200)
    net.process(synth.make_image(1, 24, 24), 200)  # warm the thread pool
    t = time.time()
    out = net.process(img, 200)
    dt = time.time() - t
    threads = oracle.max_threads()
    res = {"value": round(out.shape[0] * out.shape[1] / 1e6 / dt, 5), "unit": "Mpix/s", "cores": threads,
           "kind": "port", "host_cpus": os.cpu_count(),
           "sample": "oracle/realsr_oracle.c (CPU restatement of RealSR::process_cpu, not ncnn), one 200x200 "
                     "image at tile=200 = one padded 220x220 tile of the C2 frame, %.1f s, %.1f GFLOP/s" % (
                         dt, 220 * 220 * FLOP_PER_PADDED_LR_PX / dt / 1e9)}
    try:
        oracle.set_threads(1)
        small = synth.make_image(1235, 64, 64)
        t = time.time()
        o1 = net.process(small, 128)
        d1 = time.time() - t
        res["c1_single_thread"] = {
            "value": round(o1.shape[0] * o1.shape[1] / 1e6 / d1, 6), "unit": "Mpix/s", "cores": 1,
            "sample": "oracle, 1 thread, 64x64 image at tile=128 = one padded 84x84 tile (0.253 TFLOP), %.1f s, %.1f GFLOP/s; "
                      "C1 (256x256, four 148x148 tiles, 3.141 TFLOP) extrapolates to %.0f s" % (
                          d1, 84 * 84 * FLOP_PER_PADDED_LR_PX / d1 / 1e9, d1 * 4 * 148 * 148 / (84 * 84))}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-host", action="store_true", help="skip the host->host leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Test hook for a 1-GPU box: RSR_BENCH_SAME_GPU=1 puts every rank on cuda:0 and RSR_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one device), so the multi-rank control flow can be exercised without 2 GPUs.
    same_gpu = os.environ.get("RSR_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local = 0
    backend = os.environ.get("RSR_BENCH_BACKEND", "nccl")
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
        blob = torch.from_numpy(R.model_pack(pp, bp, with_w32=False)).to(dev)
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

    for _ in range(args.warmup):
        step()
    if not args.no_profile:
        sr.set_profiling(True)
        sr.get_profile(reset=True)
        sr.get_conv_times(reset=True)
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
    dt = time.perf_counter() - t0
    prof = sr.get_profile(reset=True) if not args.no_profile else None
    conv_ms = sr.get_conv_times(reset=True) if not args.no_profile else None
    sr.set_profiling(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        if backend == "nccl":
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        else:
            h = tmax.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            tmax = h
    dt = float(tmax.item())
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
            th = time.perf_counter() - t1
            tt = torch.tensor([th], dtype=torch.float64)
            if world > 1:
                if backend == "nccl":
                    tt = tt.to(dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                else:
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            host[name] = float(tt.item())
        host["identical_to_device_path"] = bool((torch.from_numpy(pin_out.array[::97, ::89].copy()).to(torch.int64).sum().item()) == checksum)
        # two caller threads on the one context (the reference's jobs_proc = 2, main.cpp:811-828): upload of frame k+1 and
        # download of frame k-1 ride under the kernels of frame k
        if world == 1:
            import threading
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
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "value_definition": "images resident in HBM (rsr_process_device); SURVEY 8(d)'s host->host rate is host_to_host below",
            "config": {
                "workload": "C2: models-DF2K, 1920x1080 RGB -> 7680x4320, scale=4, tile=200, prepadding=10, "
                            "60 tiles/frame (2,544,000 padded LR px, 91.21 TFLOP algorithmic), 1 frame per GPU per step",
                "weights": "synthetic seeded fp16-tagged x4.bin (real blobs absent from the reference checkout)",
                "io": "uint8 HWC in HBM -> uint8 HWC in HBM (rsr_process_device)",
                "parallelism": ("frames sharded 1/GPU, weights by one RCCL broadcast of %.1f MB" % (blob_bytes / 1e6)) if world > 1 else "single GPU",
                "frame_tflop": round(ppx * FLOP_PER_PADDED_LR_PX / 1e12, 2),
                "whole_path_tflops": round(ppx * FLOP_PER_PADDED_LR_PX * world * args.steps / dt / 1e12, 1),
                "whole_path_frac_of_peak": round(ppx * FLOP_PER_PADDED_LR_PX * world * args.steps / dt / 1e12 / PEAK_F16_TFLOPS / world, 4),
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
            # ~50 % of the frame, all launches of ONE kernel: rsr::conv3x3_flow<1,1,false,1,true>.  Its launches are bracketed
            # by hipEvents on the launch stream inside the engine (per-conv sums over the timed steps; the events sit inside
            # the timed region, so ms_per_step includes their ~1 us per launch).
            ring_idx = [1 + 5 * j + k for j in range(69) for k in range(4)]
            ring_ms = float(sum(conv_ms[i] for i in ring_idx))
            batches = max(1, round(prof["conv_launches"] / (351.0 * max(prof["calls"], 1))))  # tile batches per frame (1 for C2)
            ring_launches = len(ring_idx) * args.steps * batches
            ring_flops = sum(2.0 * 9 * (64 + 32 * k) * 32 for k in range(4)) * 69 * ppx * args.steps
            ach = ring_flops / (ring_ms * 1e-3) / 1e12
            ach_all = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
            c5_idx = [5 + 5 * j for j in range(69)]
            c5_ms = float(sum(conv_ms[i] for i in c5_idx))
            c5_flops = 2.0 * 9 * 192 * 64 * 69 * ppx * args.steps
            traffic, traffic_src = pmc_traffic()
            res["roofline"] = {
                "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F16_TFLOPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src or "no tracked PMC summary found",
                "traffic_unit": "B/launch, HBM (PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes); algorithmic (3.5 + 1) x 162.8 MB = 732.7e6",
                "kernel": "rsr::" + DOMINANT_KERNEL + " (276 of the 351 convs: cin 64..160 -> 32, LeakyReLU, 16-channel fp16 planes)",
                "launches": ring_launches,
                "avg_launch_us": round(ring_ms * 1e3 / max(ring_launches, 1), 2),
                "algorithmic_flop_per_launch_avg": round(ring_flops / max(ring_launches, 1)),
                "second_kernel": {"kernel": "rsr::conv3x3_flow<2, 1, false, 2, false> (69 x conv5 192 -> 64)",
                                  "achieved": round(c5_flops / (c5_ms * 1e-3) / 1e12, 1), "frac": round(c5_flops / (c5_ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                                  "avg_launch_us": round(c5_ms * 1e3 / (69 * args.steps * batches), 2)},
                "all_convs": {"achieved": round(ach_all, 1), "frac": round(ach_all / PEAK_F16_TFLOPS, 4), "launches": prof["conv_launches"],
                              "conv_ms_per_step": round(prof["conv_ms"] / args.steps, 3)},
                "pre_ms_per_step": round(prof["pre_ms"] / args.steps, 4),
                "post_ms_per_step": round(prof["post_ms"] / args.steps, 4),
                "post_GBps": round(prof["post_bytes"] / max(prof["post_ms"], 1e-9) / 1e6, 1),
                "timing": "hipEvents on the launch stream around every kernel of the timed steps (rank 0), inside the timed region",
                "note": "the board sits at its 1400 W cap during this workload (sclk ~1.75 GHz of 2.4): at that clock the MFMA peak is ~1.8 PFLOP/s",
            }
            try:
                g = board_gemm_ceiling(dev)
                res["roofline"]["vendor_gemm_same_board"] = {
                    "what": "torch.mm fp16 8192^3 (hipBLASLt/rocBLAS), measured right after the timed region",
                    "randn_tflops": round(g["randn"], 1), "zeros_tflops": round(g["zeros"], 1),
                    "achieved_over_randn_gemm": round(ach / g["randn"], 4), "all_convs_over_randn_gemm": round(ach_all / g["randn"], 4)}
            except Exception as e:  # noqa: BLE001 -- context only, never fail the bench on it
                res["roofline"]["vendor_gemm_same_board"] = {"what": "failed: %r" % (e,)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(pp, bp)
            except Exception as e:  # the oracle is optional here; never fail the GPU number on it
                res["cpu_baseline"] = {"value": None, "unit": "Mpix/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(res), flush=True)
    sr.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
