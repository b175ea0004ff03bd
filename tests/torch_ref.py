"""Independent PyTorch-CPU construction of the RRDBNet graph (cross-check of the oracle only).

Built from the canonical ESRGAN definition, NOT from the oracle's parsed graph:
RRDBNet(in=3,out=3,nf=64,nb=23,gc=32), LeakyReLU 0.2, residual scale 0.2, nearest x2 twice.
"""
import numpy as np
import torch
import torch.nn.functional as F


def rrdbnet_forward(weights, x, collect=None):
    """weights: list of 351 (W,b) numpy pairs in .bin order; x: float32 tensor (1,3,h,w)."""
    it = iter(weights)

    def conv(t, act):
        W, b = next(it)
        y = F.conv2d(t, torch.from_numpy(W), torch.from_numpy(b), padding=1)
        return F.leaky_relu(y, 0.2) if act else y

    fea = conv(x, False)
    cur = fea
    for i in range(23):
        rin = cur
        for _ in range(3):
            xx = cur
            x1 = conv(xx, True)
            x2 = conv(torch.cat((xx, x1), 1), True)
            x3 = conv(torch.cat((xx, x1, x2), 1), True)
            x4 = conv(torch.cat((xx, x1, x2, x3), 1), True)
            x5 = conv(torch.cat((xx, x1, x2, x3, x4), 1), False)
            cur = x5 * 0.2 + xx
        cur = cur * 0.2 + rin
        if collect is not None:
            collect.append(float(cur.abs().mean()))
    trunk = conv(cur, False)
    s = fea + trunk
    s = conv(F.interpolate(s, scale_factor=2, mode="nearest"), True)
    s = conv(F.interpolate(s, scale_factor=2, mode="nearest"), True)
    s = conv(s, True)
    return conv(s, False)


def net_forward_np(weights, x_chw):
    with torch.no_grad():
        y = rrdbnet_forward(weights, torch.from_numpy(np.ascontiguousarray(x_chw, dtype=np.float32))[None])
    return y[0].numpy()
