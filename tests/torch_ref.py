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


def rrdbnet_forward_fp16_storage(weights, x):
    """The same graph with fp16 STORAGE / fp32 arithmetic: every feature map is rounded to fp16 exactly where the engine -- and the
    reference's Vulkan path (realsr.cpp:44-46: use_fp16_storage) -- keeps it in memory: the input, every conv output after bias /
    activation, each residual stage (RDB: 0.2 * (conv5 + b) + x evaluated in fp32, then stored; every third RDB: 0.2 * v + rrdb_in on
    the stored v; trunk + fea).  What this differs from the fp32 graph by is what fp16 storage costs, whoever implements it."""
    it = iter(weights)

    def h(t):
        return t.half().float()

    def conv(t, act):
        W, b = next(it)
        y = F.conv2d(t, torch.from_numpy(W), torch.from_numpy(b), padding=1)
        return F.leaky_relu(y, 0.2) if act else y

    x = h(x)
    fea = h(conv(x, False))
    cur = fea
    for _ in range(23):
        rin = cur
        for j in range(3):
            xx = cur
            x1 = h(conv(xx, True))
            x2 = h(conv(torch.cat((xx, x1), 1), True))
            x3 = h(conv(torch.cat((xx, x1, x2), 1), True))
            x4 = h(conv(torch.cat((xx, x1, x2, x3), 1), True))
            v = h(conv(torch.cat((xx, x1, x2, x3, x4), 1), False) * 0.2 + xx)
            if j == 2:
                v = h(v * 0.2 + rin)
            cur = v
    s = h(h(conv(cur, False)) + fea)
    s = h(conv(F.interpolate(s, scale_factor=2, mode="nearest"), True))
    s = h(conv(F.interpolate(s, scale_factor=2, mode="nearest"), True))
    s = h(conv(s, True))
    return h(conv(s, False))


def net_forward_fp16_storage_np(weights, x_chw):
    with torch.no_grad():
        y = rrdbnet_forward_fp16_storage(weights, torch.from_numpy(np.ascontiguousarray(x_chw, dtype=np.float32))[None])
    return y[0].numpy()
