"""ctypes loader for the independent scalar statement of torchvision 0.12's deform_conv2d / nms / batched_nms
(tv_ops_scalar.cpp).  TEST INFRASTRUCTURE: imported by tests/ only.  Built by `make -C oracle/independent`
(__graft_entry__.build() does it; the tests build it on demand — g++ is part of the image)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'build', 'libtvops_scalar.so')
_lib = None


def library():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, 'tv_ops_scalar.cpp')
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.run(['make', '-s', '-C', _HERE], check=True)
        L = ctypes.CDLL(_LIB)
        fp, ip, i32, i64 = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_long
        L.tv_deform_conv2d.argtypes = [fp] * 6 + [i32] * 13
        L.tv_deform_conv2d.restype = i32
        L.tv_nms.argtypes = [fp, fp, i64, ctypes.c_float, ip]
        L.tv_nms.restype = i64
        L.tv_batched_nms.argtypes = [fp, fp, fp, i64, ctypes.c_float, ip]
        L.tv_batched_nms.restype = i64
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def deform_conv2d(x, offset, weight, bias=None, stride=1, padding=0, dilation=1, mask=None):
    """numpy in / numpy out; same argument meaning as torchvision.ops.deform_conv2d (one offset group, groups = 1)."""
    x, xp = _f(x)
    offset, op = _f(offset)
    weight, wp = _f(weight)
    B, Ci, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    out = np.zeros((B, Co, Ho, Wo), np.float32)
    none = ctypes.POINTER(ctypes.c_float)()
    mp = bp = none
    if mask is not None:
        mask, mp = _f(mask)
    if bias is not None:
        bias, bp = _f(bias)
    rc = library().tv_deform_conv2d(xp, op, mp, wp, bp, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), B, Ci, H, W, Co, kh, kw,
                                    stride, stride, padding, padding, dilation, dilation)
    assert rc == 0
    return out


def nms(boxes, scores, thr):
    boxes, bp = _f(boxes)
    scores, sp = _f(scores)
    keep = np.zeros(max(1, len(scores)), np.int64)
    k = library().tv_nms(bp, sp, len(scores), float(thr), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return keep[:k].copy()


def batched_nms(boxes, scores, class_ids, thr):
    boxes, bp = _f(boxes)
    scores, sp = _f(scores)
    class_ids, cp = _f(class_ids)
    keep = np.zeros(max(1, len(scores)), np.int64)
    k = library().tv_batched_nms(bp, sp, cp, len(scores), float(thr), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return keep[:k].copy()
