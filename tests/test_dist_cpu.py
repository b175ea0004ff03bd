"""world_size=2 gloo test of the multi-GPU host logic, runnable without a GPU.

The data path has exactly one collective: rank 0 parses+packs the model once and broadcasts the blob
(RCCL over xGMI on the GPU box; gloo here).  Frames are then sharded across ranks with no further
communication, results are gathered only to be compared with a single-process run.  The per-rank worker
here is the CPU oracle (tests may use it as a checker) -- the point is the sharding/broadcast logic that
bench.py runs on GPUs with the HIP engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model_dir, n_frames, q):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import realsr_ncnn_vulkan_amd as R
    from realsr_ncnn_vulkan_amd import synth
    pp, bp = os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin")
    # rank 0 packs; size then payload are broadcast (the only collective of the data path)
    if rank == 0:
        blob = torch.from_numpy(R.model_pack(pp, bp))
        n = torch.tensor([blob.numel()], dtype=torch.int64)
    else:
        n = torch.zeros(1, dtype=torch.int64)
    dist.broadcast(n, 0)
    if rank != 0:
        blob = torch.empty(int(n.item()), dtype=torch.uint8)
    dist.broadcast(blob, 0)
    digest = int(blob.to(torch.int64).sum().item())
    # every rank must hold a valid blob: header magic + size
    hdr = blob[:24].numpy()
    ok = int(np.frombuffer(hdr[:4], np.uint32)[0] == 0x50525352 and int(np.frombuffer(hdr[16:24], np.uint64)[0]) == blob.numel())
    net = oracle.OracleNet(pp, bp)
    mine = R.shard_frames(n_frames, world, rank)
    outs = {i: net.process(synth.make_image(100 + i, 12, 10), 8) for i in mine}
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, {k: v.tobytes() for k, v in outs.items()}, digest, ok))
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_broadcast_and_frame_sharding_world2(model_dir):
    import oracle
    from realsr_ncnn_vulkan_amd import synth
    n_frames, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, model_dir, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # every frame processed exactly once, digest identical on all ranks, blobs valid
    seen = sorted(i for (mine, _, _, _) in gathered for i in mine)
    assert seen == list(range(n_frames))
    assert len({d for (_, _, d, _) in gathered}) == 1
    assert all(ok == 1 for (_, _, _, ok) in gathered)
    net = oracle.OracleNet(os.path.join(model_dir, "x4.param"), os.path.join(model_dir, "x4.bin"))
    for mine, outs, _, _ in gathered:
        for i in mine:
            want = net.process(synth.make_image(100 + i, 12, 10), 8)
            assert outs[i] == want.tobytes()
