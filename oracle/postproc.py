"""ctypes front-end of oracle/libhvn_oracle.so (CPU restatement of
/root/reference/models/hovernet/post_proc.py:26-90).  TEST INFRASTRUCTURE ONLY.

Works under both interpreters on the build box (python3.10 main, conda python3.9
used for golden generation): only numpy + ctypes.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/hvn_oracle.c -> libhvn_oracle.so (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libhvn_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def normalize_32f(src):
    src = _c(src, np.float32)
    dst = np.empty_like(src)
    lib().hvn_o_normalize_32f(_p(src), _p(dst), ctypes.c_size_t(src.size))
    return dst


def normalize_64f32f(src):
    src = _c(src, np.float64)
    dst = np.empty(src.shape, np.float32)
    lib().hvn_o_normalize_64f32f(_p(src), _p(dst), ctypes.c_size_t(src.size))
    return dst


def sobel21(src, dx):
    src = _c(src, np.float32)
    H, W = src.shape
    dst = np.empty((H, W), np.float64)
    lib().hvn_o_sobel21(_p(src), _p(dst), H, W, int(dx))
    return dst


def gauss3_64f(src):
    src = _c(src, np.float64)
    H, W = src.shape
    dst = np.empty((H, W), np.float64)
    lib().hvn_o_gauss3_64f(_p(src), _p(dst), H, W)
    return dst


def morph_open5(src):
    src = _c(src, np.uint8)
    H, W = src.shape
    dst = np.empty((H, W), np.uint8)
    lib().hvn_o_morph_open5(_p(src), _p(dst), H, W)
    return dst


def label4(binimg):
    b = _c(binimg != 0, np.int32)
    H, W = b.shape
    lab = np.empty((H, W), np.int32)
    n = lib().hvn_o_label4(_p(b), _p(lab), H, W)
    return lab, n


def fill_holes(m):
    m = _c(m, np.int32)
    H, W = m.shape
    out = np.empty((H, W), np.uint8)
    lib().hvn_o_fill_holes(_p(m), _p(out), H, W)
    return out


def watershed(image, markers, mask):
    image = _c(image, np.float64)
    markers = _c(markers, np.int32)
    mask = _c(mask, np.int32)
    H, W = image.shape
    out = np.empty((H, W), np.int32)
    lib().hvn_o_watershed(_p(image), _p(markers), _p(mask), _p(out), H, W)
    return out


def proc_np_hv(pred, taps=False):
    """pred: HxWx3 float32 [p, h, v] -> int32 HxW instance map (post_proc.py:26-90)."""
    pred = _c(pred, np.float32)
    H, W, C = pred.shape
    assert C == 3
    out = np.empty((H, W), np.int32)
    if not taps:
        lib().hvn_o_proc_np_hv(_p(pred), H, W, _p(out))
        return out
    blb = np.empty((H, W), np.int32)
    dist = np.empty((H, W), np.float64)
    marker = np.empty((H, W), np.int32)
    lib().hvn_o_proc_np_hv_ex(_p(pred), H, W, _p(out), _p(blb), _p(dist), _p(marker))
    return out, blb, dist, marker


def proc_batch(pred_maps):
    """pred_maps: [N,H,W,3|4] float32 (infer_step output) -> [N,H,W] int32."""
    pred_maps = _c(pred_maps, np.float32)
    N, H, W, C = pred_maps.shape
    out = np.empty((N, H, W), np.int32)
    lib().hvn_o_proc_batch(_p(pred_maps), N, H, W, C, C - 3, _p(out))
    return out
