#!/usr/bin/env python3
"""CPU only (no GPU minutes): which stored tensors does the +-1 uint8 bar against the fp32 CPU path (realsr.cpp:525-838) hang on?
For each stand-in model, one padded tile of the C1 frame goes through the fp32 oracle and through tests/torch_ref.py's
rrdbnet_forward_storage with different tensors kept in fp32 instead of fp16 (the reference's Vulkan path keeps all of them in fp16,
realsr.cpp:44-46).  Prints max / p99.9 / mean pre-quantise error, the headroom (one uint8 step / max error) and the uint8 outcome.
    python tools/storage_emulation.py [tile=148] > profiles/r06_storage_emulation.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch_ref  # noqa: E402
import oracle  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

MODELS = [
    ("42 base (models-DF2K stand-in)", 42, {}),
    ("43 (models-DF2K_JPEG stand-in)", 43, {}),
    ("44 hot=32", 44, {"hot": 32.0, "last_gain": 0.15}),
    ("45 chan_sigma=1 last_gain=0.12", 45, {"chan_sigma": 1.0, "last_gain": 0.12}),
    ("45 chan_sigma=1 last_gain=0.2", 45, {"chan_sigma": 1.0, "last_gain": 0.2}),
]
MODES = [
    ("all fp16 (reference Vulkan path = engine default)", {}),
    ("fp32 conv_last -> uint8 (no output rounding)", {"out32": True}),
    ("fp32 trunk", {"trunk": "fp32", "fea16": False}),
    ("fp32 trunk + fp32 output", {"trunk": "fp32", "fea16": False, "out32": True}),
    ("fp32 at the 23 RRDB outputs only + fp32 output", {"trunk": "rrdb", "out32": True}),
    ("hi + fp16 residue trunk, fp32 output", {"trunk": "split16", "fea16": False, "out32": True}),
    ("ENGINE precise mode: hi + bf8 residue trunk, fp32 output", {"trunk": "split", "fea16": False, "out32": True}),
]


def main():
    tile = 148
    only = None
    for a in sys.argv[1:]:
        k, v = a.split("=")
        if k == "tile":
            tile = int(v)
        if k == "model":
            only = int(v)
    img = synth.make_image(1234, 256, 256)
    big = np.pad(img, ((10, 10), (10, 10), (0, 0)), mode="reflect")
    t = np.ascontiguousarray(big[:tile, :tile, :3].astype(np.float32).transpose(2, 0, 1) * np.float32(1 / 255.0))
    q = lambda z: np.clip(np.floor(z * 255.0 + 0.5), 0, 255).astype(np.int32)  # noqa: E731
    print("# pre-quantise error of the network output vs the fp32 oracle, [0,1] units; one uint8 step = 3.92e-3; %dx%d padded tile of the C1 frame" % (tile, tile))
    for mi, (name, seed, kw) in enumerate(MODELS):
        if only is not None and only != mi:
            continue
        d = synth.make_model_dir("/tmp/rsr_models_probe", "m_%d_%s" % (seed, "_".join("%s%g" % kv for kv in sorted(kw.items()))), seed, **kw)
        pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
        net = oracle.OracleNet(pp, bp)
        weights = [(c["weight"], c["bias"]) for c in (net.conv(i) for i in range(net.num_convs))]
        a = net.forward(t)
        print("model %s: fp32 output range %.3f .. %.3f" % (name, a.min(), a.max()))
        base = None
        for mname, mkw in MODES:
            b = torch_ref.net_forward_storage_np(weights, t, **mkw)
            e = np.abs(b - a)
            dq = np.abs(q(b) - q(a))
            if base is None:
                base = e.max()
            print("  %-56s max %.3e (x%.2f)  p99.9 %.3e  mean %.3e  headroom %.2f | uint8 max %d, != on %.2f %%" % (
                mname, e.max(), base / e.max(), np.quantile(e, 0.999), e.mean(), (1 / 255.0) / e.max(), dq.max(), 100 * (dq > 0).mean()))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
