"""ctypes loader for the CPU oracle (oracle/realsr_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (realsr-ncnn-vulkan_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librealsr_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "realsr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_net_load.restype = C.c_void_p
        L.orc_net_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_net_free.argtypes = [C.c_void_p]
        L.orc_net_num_layers.argtypes = [C.c_void_p]
        L.orc_net_num_convs.argtypes = [C.c_void_p]
        L.orc_net_bin_encoding.argtypes = [C.c_void_p]
        L.orc_net_conv_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_float),
                                        C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float))]
        L.orc_net_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_conv3x3.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_int, C.c_float, C.c_void_p]
        L.orc_bicubic.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_preproc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_postproc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int]
        L.orc_preproc_tta.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int]
        L.orc_postproc_tta.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        # The oracle's parallel regions are small (one per layer): on many-core hosts (the GPU box has
        # 256 hardware threads) an unbounded OpenMP team is ~1000x slower than 16 threads.
        L.orc_set_threads(max(1, min(16, os.cpu_count() or 1)))
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleNet:
    """Parsed x4.param + x4.bin, interpreted layer by layer (generic DAG executor)."""

    def __init__(self, param_path, bin_path):
        err = C.create_string_buffer(256)
        self._h = lib().orc_net_load(str(param_path).encode(), str(bin_path).encode(), err, 256)
        if not self._h:
            raise RuntimeError("oracle load failed: " + err.value.decode())

    def close(self):
        if self._h:
            lib().orc_net_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_layers(self):
        return lib().orc_net_num_layers(self._h)

    @property
    def num_convs(self):
        return lib().orc_net_num_convs(self._h)

    @property
    def bin_encoding(self):
        return lib().orc_net_bin_encoding(self._h)

    def conv(self, i):
        cin, cout, act = C.c_int(), C.c_int(), C.c_int()
        slope = C.c_float()
        w = C.POINTER(C.c_float)()
        b = C.POINTER(C.c_float)()
        if lib().orc_net_conv_info(self._h, i, cin, cout, act, slope, w, b) != 0:
            raise IndexError(i)
        W = np.ctypeslib.as_array(w, shape=(cout.value, cin.value, 3, 3)).copy()
        B = np.ctypeslib.as_array(b, shape=(cout.value,)).copy()
        return dict(cin=cin.value, cout=cout.value, act=act.value, slope=slope.value, weight=W, bias=B)

    def forward(self, x):
        """x: float32 CHW (3,h,w) in [0,1] -> (3,4h,4w)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        c, h, w = x.shape
        out = np.empty((3, 4 * h, 4 * w), dtype=np.float32)
        oc, oh, ow = C.c_int(), C.c_int(), C.c_int()
        rc = lib().orc_net_forward(self._h, _p(x), c, h, w, _p(out), oc, oh, ow)
        if rc != 0:
            raise RuntimeError("orc_net_forward rc=%d" % rc)
        assert (oc.value, oh.value, ow.value) == out.shape
        return out

    def process(self, img, tilesize, prepadding=10, scale=4, tta=False, want_f32=False):
        """img: uint8 HWC (h,w,c).  Returns uint8 (4h,4w,c) [and float32 pre-quantise image]."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w, c = img.shape
        out = np.zeros((h * scale, w * scale, c), dtype=np.uint8)
        f32 = np.zeros((h * scale, w * scale, c), dtype=np.float32) if want_f32 else None
        rc = lib().orc_process(self._h, _p(img), w, h, c, _p(out), scale, tilesize, prepadding, int(bool(tta)), _p(f32))
        if rc != 0:
            raise RuntimeError("orc_process rc=%d" % rc)
        return (out, f32) if want_f32 else out


def conv3x3(x, weight, bias, act=0, slope=0.2):
    x = np.ascontiguousarray(x, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    cin, h, w = x.shape
    cout = weight.shape[0]
    out = np.empty((cout, h, w), dtype=np.float32)
    lib().orc_conv3x3(_p(x), cin, h, w, _p(weight), _p(bias), cout, act, slope, _p(out))
    return out


def bicubic(a, oh, ow):
    a = np.ascontiguousarray(a, dtype=np.float32)
    h, w = a.shape
    out = np.empty((oh, ow), dtype=np.float32)
    lib().orc_bicubic(_p(a), h, w, oh, ow, _p(out))
    return out


def preproc(band, outw, outh, pad_top, pad_left, crop_x, crop_y, alphaw=0, alphah=0, bgr=0):
    """Scalar restatement of realsr_preproc.comp.  band: uint8 (h,w,c).  Returns fp16 (3,outh,outw)[, alpha]."""
    band = np.ascontiguousarray(band, dtype=np.uint8)
    h, w, c = band.shape
    top = np.zeros((3, outh, outw), dtype=np.float16)
    alpha = np.zeros((alphah, alphaw), dtype=np.float16) if c == 4 else None
    lib().orc_preproc(_p(band), w, h, c, _p(top), outw, outh, outw * outh, pad_top, pad_left, crop_x, crop_y,
                      _p(alpha), alphaw, alphah, bgr)
    return (top, alpha) if c == 4 else top


def postproc(bottom, out_band, offset_x, gx_max, crop_x, crop_y, alpha=None, bgr=0):
    """Scalar restatement of realsr_postproc.comp; writes into out_band (uint8 (outh,outw,c)) in place."""
    bottom = np.ascontiguousarray(bottom, dtype=np.float16)
    _, h, w = bottom.shape
    outh, outw, c = out_band.shape
    assert out_band.flags.c_contiguous and out_band.dtype == np.uint8
    aw = alpha.shape[1] if alpha is not None else 0
    ah = alpha.shape[0] if alpha is not None else 0
    if alpha is not None:
        alpha = np.ascontiguousarray(alpha, dtype=np.float16)
    lib().orc_postproc(_p(bottom), w, h, w * h, _p(alpha), aw, ah, _p(out_band), outw, outh, offset_x, gx_max,
                       crop_x, crop_y, c, bgr)
    return out_band


def preproc_tta(band, outw, outh, pad_top, pad_left, crop_x, crop_y, bgr=0):
    band = np.ascontiguousarray(band, dtype=np.uint8)
    h, w, c = band.shape
    tops = [np.zeros((3, outh, outw) if k < 4 else (3, outw, outh), dtype=np.float16) for k in range(8)]
    arr = (C.c_void_p * 8)(*[t.ctypes.data for t in tops])
    lib().orc_preproc_tta(_p(band), w, h, c, arr, outw, outh, outw * outh, pad_top, pad_left, crop_x, crop_y,
                          None, 0, 0, bgr)
    return tops


def postproc_tta(bottoms, out_band, offset_x, gx_max, crop_x, crop_y, bgr=0):
    bottoms = [np.ascontiguousarray(b, dtype=np.float16) for b in bottoms]
    _, h, w = bottoms[0].shape
    outh, outw, c = out_band.shape
    arr = (C.c_void_p * 8)(*[b.ctypes.data for b in bottoms])
    lib().orc_postproc_tta(arr, w, h, w * h, None, 0, 0, _p(out_band), outw, outh, offset_x, gx_max, crop_x,
                           crop_y, c, bgr)
    return out_band


def max_threads():
    return lib().orc_max_threads()


def set_threads(n):
    lib().orc_set_threads(int(n))
