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


def _p(a):
    return a.ctypes.data


def predict(model, X, threads=1, return_visits=False):
    """C restatement of oracle.ensemble.predict (same parameter dicts)."""
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    n, p = X.shape
    out = np.empty(n)
    L = lib()
    k = model["kind"]
    visits = C.c_int64(0)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    if k == "lm":
        c = f64(model["coef"])
        L.oracle_predict_lm(C.c_void_p(_p(c)), p, C.c_void_p(_p(X)), C.c_int64(n), int(threads), C.c_void_p(_p(out)))
    elif k == "nnet":
        w = f64(model["wts"])
        L.oracle_predict_nnet(C.c_void_p(_p(w)), p, int(model["size"]), C.c_double(model["y_scale"]), C.c_double(model["y_shift"]),
                              C.c_void_p(_p(X)), C.c_int64(n), int(threads), C.c_void_p(_p(out)))
    elif k == "earth":
        c, d, u = f64(model["coef"]), i32(model["dirs"]), f64(model["cuts"])
        L.oracle_predict_earth(C.c_void_p(_p(c)), C.c_void_p(_p(d)), C.c_void_p(_p(u)), int(c.size), p, C.c_void_p(_p(X)),
                               C.c_int64(n), int(threads), C.c_void_p(_p(out)))
    elif k == "svr":
        a, sv, xc, xs = f64(model["alpha"]), f64(model["sv"]), f64(model["x_center"]), f64(model["x_scale"])
        L.oracle_predict_svr(C.c_void_p(_p(a)), C.c_void_p(_p(sv)), C.c_int64(a.size), p, C.c_double(model["b"]),
                             C.c_double(model["sigma"]), C.c_void_p(_p(xc)), C.c_void_p(_p(xs)), C.c_double(model["y_center"]),
                             C.c_double(model["y_scale"]), C.c_void_p(_p(X)), C.c_int64(n), int(threads), C.c_void_p(_p(out)))
    elif k == "gbm":
        off, var, val = i64(model["tree_offsets"]), i32(model["split_var"]), f64(model["split_val"])
        l, r, m = i32(model["left"]), i32(model["right"]), i32(model["missing"])
        L.oracle_predict_gbm(C.c_double(model["init_f"]), C.c_int64(off.size - 1), C.c_void_p(_p(off)), C.c_void_p(_p(var)),
                             C.c_void_p(_p(val)), C.c_void_p(_p(l)), C.c_void_p(_p(r)), C.c_void_p(_p(m)), p, C.c_void_p(_p(X)),
                             C.c_int64(n), int(threads), C.c_void_p(_p(out)), C.byref(visits))
    elif k == "rf":
        off, l, r, st = i64(model["tree_offsets"]), i32(model["left"]), i32(model["right"]), i32(model["status"])
        bv, sp, npred = i32(model["best_var"]), f64(model["split"]), f64(model["node_pred"])
        L.oracle_predict_rf(C.c_int64(off.size - 1), C.c_void_p(_p(off)), C.c_void_p(_p(l)), C.c_void_p(_p(r)), C.c_void_p(_p(st)),
                            C.c_void_p(_p(bv)), C.c_void_p(_p(sp)), C.c_void_p(_p(npred)), p, C.c_void_p(_p(X)), C.c_int64(n),
                            int(threads), C.c_void_p(_p(out)), C.byref(visits))
    else:
        raise ValueError(k)
    return (out, visits.value) if return_visits else out


def ensemble(models, weights, wt_total, X, threads=1):
    acc = None
    for m, w in zip(models, weights):
        pk = predict(m, X, threads) * w
        acc = pk if acc is None else acc + pk
    return acc / wt_total
