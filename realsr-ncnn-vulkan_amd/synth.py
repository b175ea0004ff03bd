"""Synthetic model directories and images for tests / bench / smoke.

The real `models/*/x4.bin` blobs are not shipped with the reference checkout
(/root/reference/.MISSING_LARGE_BLOBS) and /root/reference does not exist on the GPU box, so
everything the tests and the bench need is generated here, deterministically from a seed:

* `write_param()`   -- emits the ncnn `.param` text of the canonical ESRGAN RRDBNet(3,3,64,23,gc=32)
                       exactly as `models/models-DF2K/x4.param` spells it (tests/test_model_io.py
                       checks byte equality against the reference file when it is present).
* `make_weights()`  -- seeded, fp16-representable conv weights/biases (351 convs, OIHW).
* `write_bin()`     -- ncnn `.bin` in either encoding ncnn can read for a type-0 blob
                       (fp16-tagged 0x01306B47, or raw fp32 tag 0) -- SURVEY.md Appendix A.2.
* `make_image()`    -- seeded RGB/RGBA uint8 test image (low-pass noise + gradients + hard edges).

Nothing here is on the product compute path.
"""
import os

import numpy as np

NB = 23  # RRDB blocks
NF = 64
GC = 32

FP16_TAG = 0x01306B47


def conv_specs():
    """(cin, cout, act) of the 351 convs in .param/.bin order (act: 0 none, 2 leakyrelu 0.2)."""
    specs = [(3, NF, 0)]
    for _ in range(NB * 3):
        for k in range(4):
            specs.append((NF + k * GC, GC, 2))
        specs.append((NF + 4 * GC, NF, 0))
    specs.append((NF, NF, 0))  # trunk_conv
    specs.append((NF, NF, 2))  # upconv1
    specs.append((NF, NF, 2))  # upconv2
    specs.append((NF, NF, 2))  # HRconv
    specs.append((NF, 3, 0))  # conv_last
    return specs


# --------------------------------------------------------------------------------------------
# .param writer
# --------------------------------------------------------------------------------------------
def _param_layers():
    """Layer list before Split insertion: (type, name, [in blobs], out blob, params string)."""
    layers = []
    node = [0]

    def blob_of(n):
        return str(703 + n)

    def conv(src, cin, cout, lrelu, out_name=None):
        n = node[0]
        node[0] += 1
        if lrelu:
            out = blob_of(node[0])  # fused LeakyRelu node owns the blob name
            node[0] += 1
        else:
            out = blob_of(n)
        if out_name:
            out = out_name
        p = "0=%d 1=3 4=1 5=1 6=%d" % (cout, cin * cout * 9)
        if lrelu:
            p += " 9=2 -23310=1,2.000000e-01"
        layers.append(("Convolution", "Conv_%d" % n, [src], out, p))
        return out

    def concat(srcs):
        n = node[0]
        node[0] += 1
        out = blob_of(n)
        layers.append(("Concat", "Concat_%d" % n, list(srcs), out, ""))
        return out

    def axpy(a, b):  # 0.2*a + 1.0*b  (ONNX: Constant, Mul, Add)
        node[0] += 2
        n = node[0]
        node[0] += 1
        out = blob_of(n)
        layers.append(("Eltwise", "Add_%d" % n, [a, b], out, "0=1 -23301=2,2.000000e-01,1.000000e+00"))
        return out

    layers.append(("Input", "input.1", [], "data", ""))
    fea = conv("data", 3, NF, False)
    cur = fea
    for _ in range(NB):
        rrdb_in = cur
        for _ in range(3):
            x = cur
            feats = [x]
            for k in range(4):
                src = x if k == 0 else concat(feats)
                feats.append(conv(src, NF + k * GC, GC, True))
            x5 = conv(concat(feats), NF + 4 * GC, NF, False)
            cur = axpy(x5, x)
        cur = axpy(cur, rrdb_in)
    trunk = conv(cur, NF, NF, False)
    n = node[0]
    node[0] += 1
    s = blob_of(n)
    layers.append(("BinaryOp", "Add_%d" % n, [fea, trunk], s, ""))
    for _ in range(2):
        node[0] += 28  # ONNX shape-computation nodes feeding Resize
        n = node[0]
        node[0] += 1
        up = blob_of(n)
        layers.append(("Interp", "Resize_%d" % n, [s], up, "0=1 1=2.0 2=2.0"))
        s = conv(up, NF, NF, True)
    s = conv(s, NF, NF, True)  # HRconv
    conv(s, NF, 3, False, out_name="output")
    return layers


def param_text():
    layers = _param_layers()
    # consumers per blob, in layer order
    consumers = {}
    for li, (_, _, ins, _, _) in enumerate(layers):
        for k, b in enumerate(ins):
            consumers.setdefault(b, []).append((li, k))
    out_layers = []
    rename = {}  # (layer idx, input slot) -> split output name
    nsplit = 0
    nblobs = 0
    for li, (typ, name, ins, out, p) in enumerate(layers):
        ins2 = [rename.get((li, k), b) for k, b in enumerate(ins)]
        out_layers.append((typ, name, ins2, [out], p))
        nblobs += 1
        cons = consumers.get(out, [])
        if len(cons) > 1:
            outs = ["%s_splitncnn_%d" % (out, i) for i in range(len(cons))]
            # ncnn2onnx assigns split outputs to consumers in reverse order
            for i, key in enumerate(cons):
                rename[key] = outs[len(cons) - 1 - i]
            out_layers.append(("Split", "splitncnn_%d" % nsplit, [out], outs, ""))
            nsplit += 1
            nblobs += len(outs)
    lines = ["7767517", "%d %d" % (len(out_layers), nblobs)]
    for typ, name, ins, outs, p in out_layers:
        s = "%-24s %-24s %d %d" % (typ, name, len(ins), len(outs))
        for b in ins + outs:
            s += " " + b
        if p:
            s += " " + p
        lines.append(s)
    return "\n".join(lines) + "\n"


def write_param(path):
    with open(path, "w") as f:
        f.write(param_text())


# --------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------
def make_weights(seed, rdb_gain=0.6, io_gain=1.0, trunk_gain=0.02, last_gain=0.12, bias_std=0.02, round_fp16=True, hot=1.0, chan_sigma=0.0):
    """Seeded synthetic weights, rounded to fp16-representable fp32.

    He-normal (fan_in, leaky slope 0.2) scaled by `rdb_gain` inside the dense blocks (ESRGAN initialises
    them at 0.1 x He and training grows them) and by `io_gain` for conv_first / trunk / up / HR convs, so that
    the dense-block convs contribute ~20 % of the final trunk signal (the canonical RRDB residual wiring grows the
    trunk by ~1.2x per RRDB, i.e. ~66x over 23 blocks, whatever the weights are -- trunk_conv is therefore scaled by
    `trunk_gain` to bring it back next to `fea`).  `last_gain` sizes conv_last so the output is ~0.55 +- 0.2 and
    over/undershoots [0,1] on a few % of the pixels: the +-1 uint8 parity check is then sensitive to every layer
    and exercises both clamps.  conv_last gets a bias of 0.5.

    `hot` (weight-statistics headroom, tests/test_gpu_round2.py): conv_first is multiplied by it and conv_last divided by it -- every
    feature map in between (the network is positively homogeneous up to its biases) is `hot` times larger, the output is not.
    hot = 256 puts the trunk activations at ~1e3..1e4 of fp16's 65,504; hot = 8192 overflows fp16 storage.

    `chan_sigma` > 0: every conv's OUTPUT channels get log-normal gains exp(sigma * N(0,1)), normalised to unit RMS per conv -- the
    channel-to-channel spread trained ESRGAN weights show (a few loud channels, many quiet ones) instead of i.i.d. filters; sigma = 1
    spreads the channel scales over ~50x.
    """
    rng = np.random.default_rng(seed)
    ws = []
    specs = conv_specs()
    for i, (cin, cout, act) in enumerate(specs):
        he = np.sqrt(2.0 / ((1 + 0.2 ** 2) * cin * 9))
        in_rdb = 1 <= i <= NB * 15
        gain = rdb_gain if in_rdb else io_gain
        if i == NB * 15 + 1:
            gain = trunk_gain
        if i == len(specs) - 1:
            gain = last_gain
        w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * np.float32(he * gain)
        b = rng.standard_normal(cout).astype(np.float32) * np.float32(bias_std)
        if chan_sigma > 0 and i != len(specs) - 1:
            g = np.exp(chan_sigma * rng.standard_normal(cout)).astype(np.float32)
            g /= np.sqrt(np.mean(g * g))
            w, b = w * g[:, None, None, None], b * g
        if i == 0:
            w, b = w * np.float32(hot), b * np.float32(hot)
        if i == len(specs) - 1:
            w = w / np.float32(hot)
            b = b + np.float32(0.5)
        if round_fp16:  # False: full fp32 weights (a raw-fp32 x4.bin whose values the fp16 packer has to round)
            w = w.astype(np.float16).astype(np.float32)
            b = b.astype(np.float16).astype(np.float32)
        ws.append((w, b))
    return ws


def write_bin(path, weights, encoding="fp16"):
    """ncnn ModelBin stream: per conv, type-0 weight blob then raw-fp32 bias (Appendix A.2)."""
    with open(path, "wb") as f:
        for w, b in weights:
            if encoding == "fp16":
                f.write(np.uint32(FP16_TAG).tobytes())
                h = w.astype(np.float16).reshape(-1)
                f.write(h.tobytes())
                if (h.size * 2) % 4:
                    f.write(b"\0" * (4 - (h.size * 2) % 4))
            elif encoding == "fp32":
                f.write(np.uint32(0).tobytes())
                f.write(w.astype(np.float32).tobytes())
            else:
                raise ValueError(encoding)
            f.write(b.astype(np.float32).tobytes())


def make_model_dir(root, name="models-DF2K", seed=42, encoding="fp16", **kw):
    """Create <root>/<name>/x4.param + x4.bin; returns the directory.  Reuses an existing one."""
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    pp, bp = os.path.join(d, "x4.param"), os.path.join(d, "x4.bin")
    stamp = os.path.join(d, ".seed")
    tag = "%d %s %r" % (seed, encoding, sorted(kw.items()))
    if os.path.exists(stamp) and open(stamp).read() == tag and os.path.exists(pp) and os.path.exists(bp):
        return d
    write_param(pp)
    write_bin(bp, make_weights(seed, **kw), encoding)
    with open(stamp, "w") as f:
        f.write(tag)
    return d


# --------------------------------------------------------------------------------------------
# images
# --------------------------------------------------------------------------------------------
def make_image(seed, w, h, c=3):
    """uint8 HWC image: smooth noise + gradients + a few hard edges, spanning 0..255."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.empty((h, w, c), dtype=np.float32)
    for q in range(c):
        n = rng.random((h + 8, w + 8)).astype(np.float32)
        # cheap separable low-pass (box 5) x2
        for _ in range(2):
            n = (n[:, :-4] + n[:, 1:-3] + n[:, 2:-2] + n[:, 3:-1] + n[:, 4:]) / 5
            n = (n[:-4] + n[1:-3] + n[2:-2] + n[3:-1] + n[4:]) / 5
        n = n[:h, :w]
        n = (n - n.min()) / max(float(n.max() - n.min()), 1e-6)
        g = (xx / max(w - 1, 1)) if q % 2 == 0 else (yy / max(h - 1, 1))
        v = 0.65 * n + 0.35 * g
        # hard edges / saturated patches
        v = np.where(((xx // 7 + yy // 5 + q) % 11) == 0, 1.0, v)
        v = np.where(((xx // 5 + yy // 9 + 2 * q) % 13) == 0, 0.0, v)
        img[:, :, q] = v
    img += rng.normal(0, 0.02, img.shape).astype(np.float32)
    return np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)
