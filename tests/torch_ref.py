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


def rrdbnet_forward_storage(weights, x, trunk="fp16", out32=False, fea16=True):
    """fp16 storage / fp32 arithmetic with a choice of HOW the tensors that are not only conv operands are kept (what an engine can do
    differently from the reference's Vulkan path, realsr.cpp:44-46, to come closer to the fp32 CPU path, realsr.cpp:525-838).  Conv
    operands are fp16 whatever the storage is (MFMA operands), so x1..x4 and the tensors behind the up-samplings cannot gain anything;
    what can is
      trunk -- the 64-channel residual stream (every RDB / RRDB output, fea + trunk_conv): "fp16" = rounded at every stage (the
               reference's GPU path, the engine's default), "fp32" = kept in fp32 for the residual ADDS, "split" = kept as
               hi = fp16(v) + one byte lo = e5m2((v - hi) * 2048) (what the engine's precise mode stores: hi is the plane the convs
               read), "split16" = the same with an fp16 residue (no better: profiles/r06_storage_emulation.txt), "rrdb" = fp32 only at the
               23 RRDB outputs (diagnostic);
      fea16 -- conv_first's output rounded once (it is a conv operand AND the start of the stream; the engine keeps no lo for it);
      out32 -- conv_last's result goes to the uint8 conversion without an fp16 rounding in between.
    trunk="fp16", out32=False = rrdbnet_forward_fp16_storage."""
    it = iter(weights)

    def h(t):
        return t.half().float()

    def conv(t, act):
        W, b = next(it)
        y = F.conv2d(h(t), torch.from_numpy(W), torch.from_numpy(b), padding=1)  # operands fp16 always
        return F.leaky_relu(y, 0.2) if act else y

    def split(t):
        hi = h(t)
        return hi + ((t - hi) * 2048.0).to(torch.float8_e5m2).float() / 2048.0

    def split16(t):
        hi = h(t)
        return hi + h((t - hi) * 2048.0) / 2048.0

    ht = {"fp16": h, "fp32": (lambda t: t), "split": split, "split16": split16, "rrdb": h}[trunk]
    hr = (lambda t: t) if trunk == "rrdb" else ht
    fea = conv(x, False)
    fea = h(fea) if fea16 else ht(fea)
    cur = fea
    for _ in range(23):
        rin = cur
        for j in range(3):
            xx = cur
            x1 = h(conv(xx, True))
            x2 = h(conv(torch.cat((xx, x1), 1), True))
            x3 = h(conv(torch.cat((xx, x1, x2), 1), True))
            x4 = h(conv(torch.cat((xx, x1, x2, x3), 1), True))
            v = conv(torch.cat((xx, x1, x2, x3, x4), 1), False) * 0.2 + xx
            if j == 2:
                v = hr(ht(v) * 0.2 + rin) if trunk in ("fp16", "rrdb") else ht(v * 0.2 + rin)
            else:
                v = ht(v)
            cur = v
    t = conv(cur, False)
    s = h(h(t) + fea) if trunk in ("fp16", "rrdb") else h(t + fea)
    s = h(conv(F.interpolate(s, scale_factor=2, mode="nearest"), True))
    s = h(conv(F.interpolate(s, scale_factor=2, mode="nearest"), True))
    s = h(conv(s, True))
    y = conv(s, False)
    return y if out32 else h(y)


def net_forward_storage_np(weights, x_chw, **kw):
    with torch.no_grad():
        y = rrdbnet_forward_storage(weights, torch.from_numpy(np.ascontiguousarray(x_chw, dtype=np.float32))[None], **kw)
    return y[0].numpy()
