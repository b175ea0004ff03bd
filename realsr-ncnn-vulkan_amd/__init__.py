"""realsr-ncnn-vulkan_amd -- MI355X-native RealSR x4 tiled inference.

The product is the C-ABI shared library `lib/librealsr_hip.so` (include/realsr_hip.h), built from
`csrc/` with hipcc for gfx950.  This module is only a thin ctypes binding used by the tests, the
bench and `__graft_entry__`; the reference-shaped C++ host class lives in csrc/realsr.h.

There is no CPU fallback: importing works anywhere (the library links against libamdhip64 only), but
creating a `RealSR` without a gfx950 device raises.  The directory name carries a hyphen (it is the
reference's name + `_amd`); import it through the repo-root shim `realsr_ncnn_vulkan_amd.py`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSR_LIB: load an alternative build of the same library (kernel experiments: tools/build_variant.sh)
LIB_PATH = os.environ.get("RSR_LIB") or os.path.join(_HERE, "lib", "librealsr_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

EXPORTS = [
    "rsr_create", "rsr_destroy", "rsr_load", "rsr_set_params", "rsr_process", "rsr_process_device",
    "rsr_model_pack", "rsr_load_packed", "rsr_model_info", "rsr_preproc", "rsr_preproc_tta", "rsr_postproc",
    "rsr_postproc_tta", "rsr_net_forward", "rsr_conv3x3", "rsr_set_profiling", "rsr_get_profile", "rsr_get_conv_times", "rsr_get_trace",
    "rsr_set_option", "rsr_last_error", "rsr_version", "rsr_host_alloc", "rsr_host_free",
    "rsr_set_progress_callback", "rsr_conv3x3_res", "rsr_create_group", "rsr_group_transport", "rsr_process_rows",
    "rsr_process_group", "rsr_device_memory", "rsr_process_tiles", "rsr_tile_partition", "rsr_rccl_probe", "rsr_get_stat",
    "rsr_net_forward_f32", "rsr_conv3x3_res_precise", "rsr_process_many",
]

RSR_OK, RSR_E_ARG, RSR_E_IO, RSR_E_FORMAT, RSR_E_GRAPH, RSR_E_DEVICE, RSR_E_STATE, RSR_E_NOMEM = 0, -1, -2, -3, -4, -5, -6, -7


class Profile(C.Structure):
    _fields_ = [("conv_ms", C.c_double), ("conv_flops", C.c_double), ("conv_launches", C.c_longlong),
                ("pre_ms", C.c_double), ("post_ms", C.c_double), ("pre_bytes", C.c_double),
                ("post_bytes", C.c_double), ("total_ms", C.c_double), ("tiles", C.c_longlong),
                ("calls", C.c_longlong)]


class RealSRError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rsr error %d: %s" % (code, msg))
        self.code = code


def build(force=False, verbose=False):
    """Compile csrc/ into lib/librealsr_hip.so with hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if force:
        cmd.append("-B")
    if not verbose:
        cmd.append("-s")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def lib():
    """Load the C-ABI library.  Fails loudly if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # When PyTorch shares the process (tests, bench: device tensors + torch.distributed), its bundled HIP
    # runtime must be the one that is loaded first; loading /opt/rocm's copy first leaves torch without a GPU.
    if os.environ.get("RSR_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C realsr-ncnn-vulkan_amd/csrc)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ip, cp = C.c_void_p, C.c_int, C.c_char_p
    L.rsr_create.argtypes = [C.POINTER(vp), ip, ip, ip]
    L.rsr_destroy.argtypes = [vp]
    L.rsr_destroy.restype = None
    L.rsr_load.argtypes = [vp, cp, cp]
    L.rsr_set_params.argtypes = [vp, ip, ip, ip]
    L.rsr_process.argtypes = [vp, vp, ip, ip, ip, vp]
    L.rsr_process_device.argtypes = [vp, vp, ip, ip, ip, vp, vp]
    L.rsr_model_pack.argtypes = [cp, cp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.rsr_device_memory.argtypes = [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.rsr_host_alloc.argtypes = [C.c_size_t]
    L.rsr_host_alloc.restype = vp
    L.rsr_host_free.argtypes = [vp]
    L.rsr_host_free.restype = None
    L.rsr_set_progress_callback.argtypes = [vp, vp, vp]
    L.rsr_load_packed.argtypes = [vp, vp, C.c_size_t, ip]
    L.rsr_model_info.argtypes = [cp, cp, C.POINTER(ip), C.POINTER(ip), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_longlong), C.POINTER(ip)]
    L.rsr_preproc.argtypes = [vp, vp, ip, ip, ip, vp, ip, ip, ip, ip, ip, ip, vp, ip, ip]
    L.rsr_preproc_tta.argtypes = [vp, vp, ip, ip, ip, C.POINTER(vp), ip, ip, ip, ip, ip, ip]
    L.rsr_postproc.argtypes = [vp, vp, ip, ip, vp, ip, ip, vp, ip, ip, ip, ip, ip, ip, ip]
    L.rsr_postproc_tta.argtypes = [vp, C.POINTER(vp), ip, ip, vp, ip, ip, ip, ip, ip, ip, ip]
    L.rsr_net_forward.argtypes = [vp, vp, ip, ip, vp]
    L.rsr_conv3x3.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, ip, ip, vp]
    L.rsr_conv3x3_res.argtypes = [vp, vp, ip, ip, ip, vp, vp, ip, C.c_float, ip, vp, C.c_float, vp]
    L.rsr_process_many.argtypes = [vp, ip, C.POINTER(vp), C.POINTER(ip), C.POINTER(ip), C.POINTER(ip), C.POINTER(vp), C.POINTER(ip)]
    L.rsr_net_forward_f32.argtypes = [vp, vp, ip, ip, vp]
    L.rsr_conv3x3_res_precise.argtypes = [vp, vp, vp, ip, ip, ip, vp, vp, C.c_float, ip, vp, vp, C.c_float, vp, vp]
    L.rsr_create_group.argtypes = [C.POINTER(vp), C.POINTER(ip), ip, ip, cp, cp]
    L.rsr_group_transport.restype = cp
    L.rsr_process_rows.argtypes = [vp, vp, ip, ip, ip, vp, ip, ip]
    L.rsr_process_tiles.argtypes = [vp, vp, ip, ip, ip, vp, ip, ip]
    L.rsr_process_group.argtypes = [C.POINTER(vp), ip, vp, ip, ip, ip, vp]
    L.rsr_tile_partition.argtypes = [ip, ip, ip, ip, ip, C.POINTER(ip)]
    L.rsr_set_profiling.argtypes = [vp, ip]
    L.rsr_get_profile.argtypes = [vp, C.POINTER(Profile), ip]
    L.rsr_get_conv_times.argtypes = [vp, C.POINTER(C.c_double), ip, ip]
    L.rsr_get_trace.argtypes = [vp, C.POINTER(C.c_ulonglong), ip]
    L.rsr_set_option.argtypes = [vp, cp, C.c_longlong]
    L.rsr_get_stat.argtypes = [vp, cp, C.POINTER(C.c_double)]
    L.rsr_rccl_probe.argtypes = []
    L.rsr_last_error.argtypes = [vp]
    L.rsr_last_error.restype = cp
    L.rsr_version.restype = cp
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def model_info(param_path, bin_path):
    """Host-only parse + graph validation (no GPU).  Returns dict."""
    L = lib()
    nl, nc, enc = C.c_int(), C.c_int(), C.c_int()
    nw, nb = C.c_longlong(), C.c_longlong()
    rc = L.rsr_model_info(str(param_path).encode(), str(bin_path).encode(), nl, nc, nw, nb, enc)
    if rc != 0:
        raise RealSRError(rc, L.rsr_last_error(None).decode())
    return dict(n_layers=nl.value, n_convs=nc.value, n_weights=nw.value, n_biases=nb.value, bin_encoding=enc.value)


def model_pack(param_path, bin_path):
    """Host-only: parse, validate and pack the model into one relocatable blob (np.uint8 array, 33.5 MB): what rsr_load
    uploads and what the multi-GPU broadcast carries."""
    L = lib()
    need = C.c_size_t()
    rc = L.rsr_model_pack(str(param_path).encode(), str(bin_path).encode(), None, 0, need)
    if rc != 0:
        raise RealSRError(rc, L.rsr_last_error(None).decode())
    buf = np.zeros(need.value, dtype=np.uint8)
    rc = L.rsr_model_pack(str(param_path).encode(), str(bin_path).encode(), _p(buf), buf.size, need)
    if rc != 0:
        raise RealSRError(rc, L.rsr_last_error(None).decode())
    return buf


class PinnedArray:
    """A numpy uint8 view of rsr_host_alloc'ed (pinned) memory; free() or let it go out of scope."""

    def __init__(self, shape):
        self._L = lib()
        n = int(np.prod(shape))
        self._p = self._L.rsr_host_alloc(n)
        if not self._p:
            raise MemoryError("rsr_host_alloc(%d)" % n)
        self.array = np.ctypeslib.as_array((C.c_uint8 * n).from_address(self._p)).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            self._L.rsr_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class RealSR:
    """Python mirror of the reference's `class RealSR` (realsr.h:13-42) over the C-ABI.

    RealSR(gpuid, tta_mode=False, num_threads=1); load(parampath, modelpath); fields scale / tilesize /
    prepadding; process(in HWC uint8) -> out HWC uint8.
    """

    def __init__(self, gpuid, tta_mode=False, num_threads=1, _adopt=None):
        self._L = lib()
        if _adopt is not None:
            h = _adopt
        else:
            h = C.c_void_p()
            rc = self._L.rsr_create(C.byref(h), int(gpuid), int(bool(tta_mode)), int(num_threads))
            if rc != 0:
                raise RealSRError(rc, self._L.rsr_last_error(None).decode())
        self._h = h
        self.scale, self.tilesize, self.prepadding = 4, 200, 10
        self.tta_mode = bool(tta_mode)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rsr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RealSRError(rc, self._L.rsr_last_error(self._h).decode())

    def load(self, parampath, modelpath):
        self._ck(self._L.rsr_load(self._h, str(parampath).encode(), str(modelpath).encode()))
        return 0

    def load_packed(self, blob, device_ptr=None):
        """blob: np.uint8 array (host), or pass device_ptr (int) + blob = nbytes."""
        if device_ptr is not None:
            self._ck(self._L.rsr_load_packed(self._h, C.c_void_p(int(device_ptr)), int(blob), 1))
        else:
            blob = np.ascontiguousarray(blob, dtype=np.uint8)
            self._ck(self._L.rsr_load_packed(self._h, _p(blob), blob.size, 0))

    def _push_params(self):
        self._ck(self._L.rsr_set_params(self._h, int(self.scale), int(self.tilesize), int(self.prepadding)))

    def set_option(self, key, value):
        self._ck(self._L.rsr_set_option(self._h, key.encode(), int(value)))

    def get_stat(self, key):
        """rsr_get_stat: read-only engine state ("plan_batches", "workspace_mb", "lane_out_mb", "last_test_us", ...)."""
        v = C.c_double(0)
        self._ck(self._L.rsr_get_stat(self._h, key.encode(), C.byref(v)))
        return v.value

    def process(self, img, out=None, push_params=True):
        """out: optional preallocated (4h, 4w, c) uint8 array (e.g. PinnedArray(...).array)."""
        img = np.ascontiguousarray(img, dtype=np.uint8) if not (isinstance(img, np.ndarray) and img.flags.c_contiguous and img.dtype == np.uint8) else img
        h, w, c = img.shape
        if push_params:
            self._push_params()
        if out is None:
            out = np.empty((h * self.scale, w * self.scale, c), dtype=np.uint8)
        assert out.shape == (h * self.scale, w * self.scale, c) and out.dtype == np.uint8 and out.flags.c_contiguous
        self._ck(self._L.rsr_process(self._h, _p(img), w, h, c, _p(out)))
        return out

    def process_many(self, imgs):
        """rsr_process_many: a list of uint8 HWC images in ONE call (small ones share tile batches); returns the list of outputs."""
        imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in imgs]
        n = len(imgs)
        outs = [np.empty((im.shape[0] * self.scale, im.shape[1] * self.scale, im.shape[2]), dtype=np.uint8) for im in imgs]
        self._push_params()
        ins_p = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
        outs_p = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        ws = (C.c_int * n)(*[im.shape[1] for im in imgs])
        hs = (C.c_int * n)(*[im.shape[0] for im in imgs])
        cs = (C.c_int * n)(*[im.shape[2] for im in imgs])
        rcs = (C.c_int * n)()
        self._ck(self._L.rsr_process_many(self._h, n, ins_p, ws, hs, cs, outs_p, rcs))
        return outs

    def process_device(self, d_in, w, h, c, d_out, stream=None):
        """d_in/d_out: integer device pointers (e.g. torch tensor .data_ptr())."""
        self._push_params()
        self._ck(self._L.rsr_process_device(self._h, C.c_void_p(int(d_in)), w, h, c, C.c_void_p(int(d_out)),
                                            C.c_void_p(int(stream)) if stream else None))

    def process_rows(self, img, out, row0, row1):
        """Tile rows [row0, row1) of img's tile grid into the full-size `out` (see rsr_process_rows)."""
        h, w, c = img.shape
        self._push_params()
        self._ck(self._L.rsr_process_rows(self._h, _p(img), w, h, c, _p(out), int(row0), int(row1)))
        return out

    def process_tiles(self, img, out, tile0, tile1):
        """Tiles [tile0, tile1) of img's row-major tile grid into the full-size `out` (see rsr_process_tiles)."""
        h, w, c = img.shape
        self._push_params()
        self._ck(self._L.rsr_process_tiles(self._h, _p(img), w, h, c, _p(out), int(tile0), int(tile1)))
        return out

    def net_forward(self, x):
        """x: float16 planar (3,h,w) -> float16 (3,4h,4w)."""
        x = np.ascontiguousarray(x, dtype=np.float16)
        _, h, w = x.shape
        out = np.empty((3, 4 * h, 4 * w), dtype=np.float16)
        self._ck(self._L.rsr_net_forward(self._h, _p(x), w, h, _p(out)))
        return out

    def net_forward_f32(self, x):
        """x: float16 planar (3,h,w) -> float32 (3,4h,4w): conv_last's unrounded result (option "precise" = 1 only)."""
        x = np.ascontiguousarray(x, dtype=np.float16)
        _, h, w = x.shape
        out = np.empty((3, 4 * h, 4 * w), dtype=np.float32)
        self._ck(self._L.rsr_net_forward_f32(self._h, _p(x), w, h, _p(out)))
        return out

    def conv3x3_res_precise(self, x, weight, bias, s1, own_input_residual=False, x_lo=None, res=None, res_lo=None, s2=1.0, want_lo=True):
        """rsr_conv3x3_res_precise: the residual forms on the hi + lo / 2048 stream (hi float16, lo uint8 = bf8 bytes);
        returns (hi, lo), lo None unless want_lo."""
        x = np.ascontiguousarray(x, dtype=np.float16)
        weight = np.ascontiguousarray(weight, dtype=np.float32)
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        cin, h, w = x.shape
        assert weight.shape[0] == 64
        u8 = lambda t: None if t is None else np.ascontiguousarray(t, dtype=np.uint8)  # noqa: E731
        x_lo, res_lo = u8(x_lo), u8(res_lo)
        res = None if res is None else np.ascontiguousarray(res, dtype=np.float16)
        out = np.empty((64, h, w), dtype=np.float16)
        out_lo = np.empty((64, h, w), dtype=np.uint8) if want_lo else None
        self._ck(self._L.rsr_conv3x3_res_precise(self._h, _p(x), _p(x_lo), cin, h, w, _p(weight), _p(bias), float(s1),
                                                 int(bool(own_input_residual)), _p(res), _p(res_lo), float(s2), _p(out), _p(out_lo)))
        return out, out_lo

    def conv3x3(self, x, weight, bias, lrelu=False, upsample2x=False):
        x = np.ascontiguousarray(x, dtype=np.float16)
        weight = np.ascontiguousarray(weight, dtype=np.float32)
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        cin, h, w = x.shape
        cout = weight.shape[0]
        s = 2 if upsample2x else 1
        out = np.empty((cout, h * s, w * s), dtype=np.float16)
        self._ck(self._L.rsr_conv3x3(self._h, _p(x), cin, h, w, int(upsample2x), _p(weight), _p(bias), cout, int(lrelu), _p(out)))
        return out

    def conv3x3_res(self, x, weight, bias, s1, own_input_residual=False, res=None, s2=1.0):
        """v = s1*(conv+b) [+ x[:cout]] [; v = s2*v + res | v + res]  -- see rsr_conv3x3_res."""
        x = np.ascontiguousarray(x, dtype=np.float16)
        weight = np.ascontiguousarray(weight, dtype=np.float32)
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        cin, h, w = x.shape
        cout = weight.shape[0]
        if res is not None:
            res = np.ascontiguousarray(res, dtype=np.float16)
            assert res.shape == (cout, h, w)
        out = np.empty((cout, h, w), dtype=np.float16)
        self._ck(self._L.rsr_conv3x3_res(self._h, _p(x), cin, h, w, _p(weight), _p(bias), cout, float(s1),
                                         int(bool(own_input_residual)), _p(res), float(s2), _p(out)))
        return out

    def preproc(self, band, outw, outh, pad_top, pad_left, crop_x, crop_y, alphaw=0, alphah=0):
        band = np.ascontiguousarray(band, dtype=np.uint8)
        h, w, c = band.shape
        top = np.zeros((3, outh, outw), dtype=np.float16)
        alpha = np.zeros((alphah, alphaw), dtype=np.float16) if c == 4 else None
        self._ck(self._L.rsr_preproc(self._h, _p(band), w, h, c, _p(top), outw, outh, pad_top, pad_left, crop_x, crop_y,
                                     _p(alpha), alphaw, alphah))
        return (top, alpha) if c == 4 else top

    def preproc_tta(self, band, outw, outh, pad_top, pad_left, crop_x, crop_y):
        band = np.ascontiguousarray(band, dtype=np.uint8)
        h, w, c = band.shape
        tops = [np.zeros((3, outh, outw) if k < 4 else (3, outw, outh), dtype=np.float16) for k in range(8)]
        arr = (C.c_void_p * 8)(*[t.ctypes.data for t in tops])
        self._ck(self._L.rsr_preproc_tta(self._h, _p(band), w, h, c, arr, outw, outh, pad_top, pad_left, crop_x, crop_y))
        return tops

    def postproc(self, bottom, out_band, offset_x, gx_max, crop_x, crop_y, alpha=None):
        bottom = np.ascontiguousarray(bottom, dtype=np.float16)
        _, h, w = bottom.shape
        outh, outw, c = out_band.shape
        assert out_band.flags.c_contiguous and out_band.dtype == np.uint8
        if alpha is not None:
            alpha = np.ascontiguousarray(alpha, dtype=np.float16)
        aw = alpha.shape[1] if alpha is not None else 0
        ah = alpha.shape[0] if alpha is not None else 0
        self._ck(self._L.rsr_postproc(self._h, _p(bottom), w, h, _p(alpha), aw, ah, _p(out_band), outw, outh, offset_x,
                                      gx_max, crop_x, crop_y, c))
        return out_band

    def postproc_tta(self, bottoms, out_band, offset_x, gx_max, crop_x, crop_y):
        bottoms = [np.ascontiguousarray(b, dtype=np.float16) for b in bottoms]
        _, h, w = bottoms[0].shape
        outh, outw, c = out_band.shape
        arr = (C.c_void_p * 8)(*[b.ctypes.data for b in bottoms])
        self._ck(self._L.rsr_postproc_tta(self._h, arr, w, h, _p(out_band), outw, outh, offset_x, gx_max, crop_x, crop_y, c))
        return out_band

    def set_profiling(self, on):
        self._ck(self._L.rsr_set_profiling(self._h, int(bool(on))))

    def get_conv_times(self, reset=True):
        arr = (C.c_double * 351)()
        self._ck(self._L.rsr_get_conv_times(self._h, arr, 351, int(bool(reset))))
        return np.array(arr[:])

    def get_trace(self, n=1024):
        arr = (C.c_ulonglong * n)()
        self._ck(self._L.rsr_get_trace(self._h, arr, n))
        return np.array(arr[:], dtype=np.uint64)

    def get_profile(self, reset=True):
        p = Profile()
        self._ck(self._L.rsr_get_profile(self._h, C.byref(p), int(bool(reset))))
        return {k: getattr(p, k) for k, _ in Profile._fields_}


def device_memory(gpuid=0):
    """(free MiB, total MiB) of HIP device gpuid (rsr_device_memory)."""
    f, t = C.c_longlong(0), C.c_longlong(0)
    rc = lib().rsr_device_memory(int(gpuid), C.byref(f), C.byref(t))
    if rc != RSR_OK:
        raise RealSRError(rc, lib().rsr_last_error(None).decode())
    return f.value, t.value


def rccl_probe():
    """rsr_rccl_probe (host-only): None when librccl can be dlopen'ed with every entry point rsr_create_group's RCCL branch
    calls, else the reason."""
    L = lib()
    rc = L.rsr_rccl_probe()
    return None if rc == RSR_OK else L.rsr_last_error(None).decode()


def create_group(gpuids, parampath, modelpath, tta_mode=False):
    """rsr_create_group: one context per GPU, the model packed once and broadcast (RCCL).  Returns (list of RealSR, transport)."""
    L = lib()
    n = len(gpuids)
    hs = (C.c_void_p * n)()
    ids = (C.c_int * n)(*[int(g) for g in gpuids])
    rc = L.rsr_create_group(hs, ids, n, int(bool(tta_mode)), str(parampath).encode(), str(modelpath).encode())
    if rc != 0:
        raise RealSRError(rc, L.rsr_last_error(None).decode())
    srs = [RealSR(int(g), tta_mode, _adopt=C.c_void_p(hs[i])) for i, g in enumerate(gpuids)]
    return srs, L.rsr_group_transport().decode()


def process_group(srs, img, out=None):
    """rsr_process_group: ONE image, its tiles dealt over the contexts in contiguous ranges of equal load."""
    L = lib()
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    for s in srs:
        s._push_params()
    if out is None:
        out = np.empty((h * 4, w * 4, c), dtype=np.uint8)
    hs = (C.c_void_p * len(srs))(*[s._h for s in srs])
    rc = L.rsr_process_group(hs, len(srs), _p(img), w, h, c, _p(out))
    if rc != 0:
        raise RealSRError(rc, L.rsr_last_error(None).decode())
    return out


def tile_partition(w, h, tilesize, prepadding, parts):
    """rsr_tile_partition: the contiguous tile ranges rsr_process_group deals to `parts` contexts (host-only)."""
    b = (C.c_int * (parts + 1))()
    n = lib().rsr_tile_partition(int(w), int(h), int(tilesize), int(prepadding), int(parts), b)
    if n < 0:
        raise RealSRError(n, lib().rsr_last_error(None).decode())
    return list(b[:n + 1])


# ---- tile sharding for multi-GPU runs: pure host logic, shared by bench + tests ----
def shard_frames(n_frames, world_size, rank):
    """Frames dealt round-robin to ranks (the reference's work-stealing queue, main.cpp:811-828,
    made deterministic).  Returns the list of frame indices this rank processes."""
    return list(range(rank, n_frames, world_size))
