"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/prop_ref.c (plain-C propagation oracle)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libprop_ref.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "prop_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(dem, weight, offset, wk, b0, scale=1.0):
    """numpy float32/float64 arrays, shapes (B,1,H,W),(B,9,H,W),(B,18,H,W),(9,) -> (B,1,H,W)."""
    dt = dem.dtype
    name, ct = ("prop_ref_f64", ctypes.c_double) if dt == np.float64 else ("prop_ref_f32", ctypes.c_float)
    B, _, H, W = dem.shape
    arrs = [np.ascontiguousarray(a, dtype=dt) for a in (dem, weight, offset, np.reshape(wk, 9))]
    out = np.empty((B, 1, H, W), dt)
    fn = getattr(lib(), name + "_forward")
    fn.argtypes = [ctypes.c_void_p] * 4 + [ct, ct, ctypes.c_void_p] + [ctypes.c_long] * 3
    fn.restype = None
    fn(*[_p(a) for a in arrs], ct(float(b0)), ct(float(scale)), _p(out), B, H, W)
    return out


def backward(gout, dem, weight, offset, wk):
    dt = dem.dtype
    name = "prop_ref_f64" if dt == np.float64 else "prop_ref_f32"
    B, _, H, W = dem.shape
    arrs = [np.ascontiguousarray(a, dtype=dt) for a in (gout, dem, weight, offset, np.reshape(wk, 9))]
    gw = np.empty((B, 9, H, W), dt)
    go = np.empty((B, 18, H, W), dt)
    gwk = np.zeros(9, np.float64)
    gb = np.zeros(1, np.float64)
    fn = getattr(lib(), name + "_backward")
    fn.argtypes = [ctypes.c_void_p] * 9 + [ctypes.c_long] * 3
    fn.restype = None
    fn(*[_p(a) for a in arrs], _p(gw), _p(go), _p(gwk), _p(gb), B, H, W)
    return gw, go, gwk.reshape(1, 1, 3, 3), gb
