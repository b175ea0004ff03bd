#!/usr/bin/env python3
"""Is a parity miss the engine's arithmetic or fp16 STORAGE itself?  For a synthetic model (synth.make_weights keyword arguments on the
command line) and one padded tile of the C1 frame, compares the network's pre-quantise output of
  (a) the fp32 oracle,
  (b) a PyTorch-CPU emulation of fp16 storage / fp32 accumulation -- every conv output (after bias, activation, residual stages) rounded
      to fp16 exactly where the engine (and the reference's Vulkan path, realsr.cpp:44-46) stores a feature map,
  (c) the engine (rsr_net_forward).
(c) close to (b) and both equally far from (a) = the storage format; (c) far from (b) = a kernel problem.
    python tools/fp16_storage_probe.py seed=45 chan_sigma=1.0 last_gain=0.2 [tile=148]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch_ref  # noqa: E402
import oracle  # noqa: E402
import realsr_ncnn_vulkan_amd as R  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402


def main():
    kw = {}
    tile = 148
    for a in sys.argv[1:]:
        k, v = a.split("=")
        if k == "tile":
            tile = int(v)
        else:
            kw[k] = float(v) if "." in v else int(v)
    seed = int(kw.pop("seed", 42))
    d = synth.make_model_dir("/tmp/rsr_models_probe", "m_%d_%s" % (seed, "_".join("%s%g" % kv for kv in sorted(kw.items()))), seed, **kw)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    net = oracle.OracleNet(pp, bp)
    weights = [(c["weight"], c["bias"]) for c in (net.conv(i) for i in range(net.num_convs))]
    img = synth.make_image(1234, 256, 256)
    big = np.pad(img, ((10, 10), (10, 10), (0, 0)), mode="reflect")
    t = big[:tile, :tile, :3].astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0)
    a = net.forward(np.ascontiguousarray(t))
    b = torch_ref.net_forward_fp16_storage_np(weights, t)
    sr = R.RealSR(0)
    sr.load(pp, bp)
    c = sr.net_forward(t.astype(np.float16)).astype(np.float32)
    sr.close()

    def rep(name, u, v):
        e = np.abs(u - v)
        q = lambda z: np.clip((z * 255.0 + 0.5).astype(np.int32), 0, 255)  # noqa: E731
        dq = np.abs(q(u) - q(v))
        print("%-34s max %.3e  p99.9 %.3e  mean %.3e | uint8: max %d, > 1 on %.4f %%, != on %.2f %%" % (
            name, e.max(), np.quantile(e, 0.999), e.mean(), dq.max(), 100 * (dq > 1).mean(), 100 * (dq > 0).mean()))

    print("model %s %r, one %dx%d padded tile; output range %.3f .. %.3f" % (seed, kw, tile, tile, a.min(), a.max()))
    rep("engine vs fp32 oracle", c, a)
    rep("fp16-storage emulation vs oracle", b, a)
    rep("engine vs fp16-storage emulation", c, b)


if __name__ == "__main__":
    main()
