"""ctypes access to oracle/_build/liboracle.so (the plain-C restatement).  TEST
INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by machisplin_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
    return _lib


def tps_eval_grid(model, xmin, ymax, xres, yres, r0, r1, c0, c1, threads=1) -> np.ndarray:
    kn = np.asfortranarray(model["knots"], dtype=np.float64)
    c = np.ascontiguousarray(model["c"], dtype=np.float64)
    d = np.ascontiguousarray(model["d"], dtype=np.float64)
    ce = np.ascontiguousarray(model["center"], dtype=np.float64)
    sc = np.ascontiguousarray(model["scale"], dtype=np.float64)
    out = np.empty((r1 - r0, c1 - c0))
    f = lib().oracle_tps_eval_grid
    f.restype = None
    f.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 2 + [C.c_double] * 4 + [C.c_int64] * 4 + [C.c_int, C.c_void_p]
    f(kn.ctypes.data, c.ctypes.data, d.ctypes.data, kn.shape[0], ce.ctypes.data, sc.ctypes.data,
      xmin, ymax, xres, yres, r0, r1, c0, c1, int(threads), out.ctypes.data)
    return out


def tps_gram(knots, threads=1) -> np.ndarray:
    kn = np.asfortranarray(knots, dtype=np.float64)
    n = kn.shape[0]
    K = np.empty((n, n))
    f = lib().oracle_tps_gram
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    f(kn.ctypes.data, n, int(threads), K.ctypes.data)
    return K
