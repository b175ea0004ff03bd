"""Generates tests/golden/cases.npz: oracle outputs for small seeded cases.

The reference ships no golden vectors and cannot be built here (ncnn submodule empty, weights
missing), so these are outputs of the CPU *oracle* (oracle/realsr_oracle.c, cross-checked against
PyTorch in tests/test_oracle.py) on the seeded synthetic model -- a drift guard for the oracle and a
fixed target for the HIP path, not reference-pinned truth ("parity unpinned").

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle  # noqa: E402
from realsr_ncnn_vulkan_amd import synth  # noqa: E402

# name: (image seed, w, h, c, tilesize, tta)
CASES = {
    "rgb_multi_tile": (21, 50, 43, 3, 32, 0),
    "rgb_single_tile": (22, 24, 20, 3, 200, 0),
    "rgb_tta": (23, 36, 22, 3, 32, 1),
    "rgb_tiny": (24, 5, 3, 3, 32, 0),
    "rgba": (25, 37, 20, 4, 32, 0),
    "rgb_wide_tile64": (26, 70, 18, 3, 64, 0),
}


def main():
    d = synth.make_model_dir(os.environ.get("RSR_MODELS", "/tmp/rsr_models"), "models-DF2K", 42)
    net = oracle.OracleNet(os.path.join(d, "x4.param"), os.path.join(d, "x4.bin"))
    out = {"bin_sha256": np.frombuffer(hashlib.sha256(open(os.path.join(d, "x4.bin"), "rb").read()).digest(), np.uint8)}
    for name, (seed, w, h, c, T, tta) in CASES.items():
        img = synth.make_image(seed, w, h, c)
        res = net.process(img, T, tta=bool(tta))
        out[name + "_in"] = img
        out[name + "_out"] = res
        out[name + "_cfg"] = np.array([seed, w, h, c, T, tta], np.int32)
        print(name, img.shape, "->", res.shape, "mean", res.mean())
    np.savez_compressed(os.path.join(HERE, "cases.npz"), **out)


if __name__ == "__main__":
    main()
