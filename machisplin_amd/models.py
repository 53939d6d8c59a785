"""Host-side mirror of the six fitted ensemble members as the hot path sees them:
flat parameter arrays in, ``terra::predict(rast_stack, model)`` /
``predict(model, data.frame)`` out (V73:447-619).  Model FITTING is out of scope and stays
in the CRAN packages; these classes wrap what the fitted R objects contain.  All
arithmetic runs in libmachisplin_hip.so; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .raster import RasterStack


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


class Model:
    """Base: owns an ``mhs_model*``.  ``label`` is the reference's one-letter code
    (b, g, n, m, r, v -- V73:340-362)."""
    label = "?"

    def __init__(self, handle, p):
        self._h = handle
        self.p = int(p)

    def predict_points(self, X) -> np.ndarray:
        """predict(model, data.frame): X is n x p, columns in rast_stack order (covariates,
        LONG, LAT).  Used for the station residuals (V73:477-482, 501-505, ...)."""
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        if X.ndim != 2 or X.shape[1] != self.p:
            raise ValueError(f"X must be n x {self.p}")
        out = np.empty(X.shape[0])
        _lib.check(_lib.lib().mhs_predict_points(self._h, X.ctypes.data, X.shape[0], out.ctypes.data))
        return out

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and _lib is not None and _lib._lib is not None:      # (module globals are gone at interpreter exit)
            _lib._lib.mhs_model_free(h)
            self._h = None


class Gam(Model):
    """mgcv::gam(resp ~ a + b + ...): no smooth terms (V73:195,600) => coefficients[p+1]."""
    label = "g"

    def __init__(self, coefficients):
        c = _f64(coefficients)
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_lm_load(c.ctypes.data, c.size - 1, C.byref(h)))
        super().__init__(h, c.size - 1)
        self.coefficients = c

    @classmethod
    def fit(cls, X, y) -> "Gam":
        """mgcv::gam(resp ~ a + b + ..., data) (V73:252, V73:600): least squares on the device (Householder QR of
        [1 X]).  X is n x p in rast_stack order, rows with NA already dropped (V73:154)."""
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        y = _f64(y)
        if X.ndim != 2 or X.shape[0] != y.size:
            raise ValueError("X must be n x p with one response per row")
        coef = np.empty(X.shape[1] + 1)
        _lib.check(_lib.lib().mhs_lm_fit(X.ctypes.data, y.ctypes.data, X.shape[0], X.shape[1], coef.ctypes.data))
        return cls(coef)


class Nnet(Model):
    """nnet::nnet(size=10, linout=TRUE) (V73:463) with the response un-scaling
    ``pred * max2.resp.f + min.resp.f`` (V73:469-470) folded in."""
    label = "n"

    def __init__(self, wts, p, size=10, max2_resp=1.0, min_resp=0.0):
        w = _f64(wts)
        if w.size != (p + 1) * size + size + 1:
            raise ValueError("wts has the wrong length for (p, size)")
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_nnet_load(w.ctypes.data, p, size, float(max2_resp), float(min_resp), C.byref(h)))
        super().__init__(h, p)

    @classmethod
    def fit(cls, X, y, wts0, size=10, maxit=10000, abstol=1e-4, reltol=1e-8) -> "Nnet":
        """nnet::nnet(mod.form, data = trainNN, size = 10, linout = TRUE, maxit = 10000) with the response scaling of
        V73:455-459 (resp - min, / max) around it, on the device (R's vmmin in one resident kernel).  wts0: the
        initial weights, nnet order (nnet draws runif(-0.7, 0.7)).  The object carries .wts, .value, .counts, .fail."""
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        y = _f64(y)
        if X.ndim != 2 or X.shape[0] != y.size:
            raise ValueError("X must be n x p with one response per row")
        n, p = X.shape
        w = _f64(wts0).copy()
        if w.size != (p + 1) * size + size + 1:
            raise ValueError("wts0 has the wrong length for (p, size)")
        mn = float(y.min())
        mx = float((y - mn).max())
        ys = np.ascontiguousarray((y - mn) / mx)
        val, counts, fail = C.c_double(), (C.c_int * 2)(), C.c_int()
        _lib.init()
        _lib.check(_lib.lib().mhs_nnet_fit(X.ctypes.data, ys.ctypes.data, n, p, int(size), w.ctypes.data, int(maxit), float(abstol),
                                           float(reltol), C.byref(val), counts, C.byref(fail)))
        m = cls(w, p, size, mx, mn)
        m.wts, m.value, m.counts, m.fail = w, val.value, (counts[0], counts[1]), fail.value
        return m


class Earth(Model):
    """earth::earth (V73:539): coefficients, dirs and cuts of the SELECTED terms."""
    label = "m"

    def __init__(self, coefficients, dirs, cuts):
        c, d, k = _f64(coefficients), _i32(dirs), _f64(cuts)
        if d.ndim != 2 or d.shape != k.shape or d.shape[0] != c.size:
            raise ValueError("dirs/cuts must be nterms x p")
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_earth_load(c.ctypes.data, d.ctypes.data, k.ctypes.data, d.shape[0], d.shape[1], C.byref(h)))
        super().__init__(h, d.shape[1])


class Ksvm(Model):
    """kernlab::ksvm eps-svr / rbfdot / scaled=TRUE (V73:560)."""
    label = "v"

    def __init__(self, alpha, xmatrix, b, sigma, x_center, x_scale, y_center, y_scale):
        a, sv = _f64(alpha), _f64(xmatrix)
        xc, xs = _f64(x_center), _f64(x_scale)
        if sv.ndim != 2 or sv.shape[0] != a.size or xc.size != sv.shape[1] or xs.size != sv.shape[1]:
            raise ValueError("xmatrix must be nSV x p with matching alpha / scaling vectors")
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_svr_load(a.ctypes.data, sv.ctypes.data, sv.shape[0], sv.shape[1], float(b),
                                           float(sigma), xc.ctypes.data, xs.ctypes.data, float(y_center),
                                           float(y_scale), C.byref(h)))
        super().__init__(h, sv.shape[1])

    @classmethod
    def fit(cls, X, y, sigma, C_=1.0, epsilon=0.1, tol=1e-3, max_iter=0) -> "Ksvm":
        """kernlab::ksvm(mod.form, data) (V73:251, V73:560) on the device: eps-svr, rbfdot, scaled = TRUE, kernlab's
        defaults for C / epsilon / tol.  sigma is kpar$sigma (kernlab's automatic value is drawn by sigest() from a
        random half of the rows).  The fitted object carries .beta (n), .n_iter and the support-vector bundle."""
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        y = _f64(y)
        if X.ndim != 2 or X.shape[0] != y.size:
            raise ValueError("X must be n x p with one response per row")
        n, p = X.shape
        beta, xc, xs = np.empty(n), np.empty(p), np.empty(p)
        b, yc, ys, it = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        _lib.init()
        _lib.check(_lib.lib().mhs_svr_fit(X.ctypes.data, y.ctypes.data, n, p, float(sigma), float(C_), float(epsilon), float(tol),
                                          int(max_iter), beta.ctypes.data, C.byref(b), xc.ctypes.data, xs.ctypes.data,
                                          C.byref(yc), C.byref(ys), C.byref(it)))
        sv = np.flatnonzero(beta != 0.0)
        Z = (np.ascontiguousarray(X)[sv] - xc) / xs
        m = cls(beta[sv], Z, b.value, sigma, xc, xs, yc.value, ys.value)
        m.beta, m.n_iter, m.sv_index = beta, int(it.value), sv
        m.params = {"kind": "svr", "alpha": beta[sv], "sv": Z, "b": b.value, "sigma": float(sigma), "x_center": xc, "x_scale": xs,
                    "y_center": yc.value, "y_scale": ys.value}
        return m


class Gbm(Model):
    """gbm object evaluated at n.trees = best.trees, type="response" (V73:497)."""
    label = "b"

    def __init__(self, init_f, tree_offsets, split_var, split_val, left, right, missing, p):
        off = _i64(tree_offsets)
        sv, val, l, r, m = _i32(split_var), _f64(split_val), _i32(left), _i32(right), _i32(missing)
        if not (sv.size == val.size == l.size == r.size == m.size == off[-1]):
            raise ValueError("node arrays must all have tree_offsets[-1] entries")
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_gbm_load(float(init_f), off.size - 1, off.ctypes.data, sv.ctypes.data,
                                           val.ctypes.data, l.ctypes.data, r.ctypes.data, m.ctypes.data, p, C.byref(h)))
        super().__init__(h, p)
        self.n_trees = int(off.size - 1)

    def staged_predict_points(self, X, step: int) -> np.ndarray:
        """predict.gbm(model, X, n.trees = step, 2 step, ...) in one walk (mhs_gbm_staged_points): (n_trees // step, n);
        the hold-out predictions machisplin.gbm.step's tree-count search is run on (V73:1843, 1919)."""
        X = np.asfortranarray(np.asarray(X, dtype=np.float64))
        if X.ndim != 2 or X.shape[1] != self.p:
            raise ValueError("X must be n x p")
        out = np.empty((self.n_trees // int(step), X.shape[0]))
        if out.size:
            _lib.check(_lib.lib().mhs_gbm_staged_points(self._h, X.ctypes.data, X.shape[0], int(step), out.ctypes.data))
        return out


class RandomForest(Model):
    """randomForest regression forest (V73:517), prediction = mean over trees."""
    label = "r"

    def __init__(self, tree_offsets, left, right, status, best_var, split, node_pred, p):
        off = _i64(tree_offsets)
        l, r, st, bv = _i32(left), _i32(right), _i32(status), _i32(best_var)
        sp, npred = _f64(split), _f64(node_pred)
        if not (l.size == r.size == st.size == bv.size == sp.size == npred.size == off[-1]):
            raise ValueError("node arrays must all have tree_offsets[-1] entries")
        h = C.c_void_p()
        _lib.check(_lib.lib().mhs_rf_load(off.size - 1, off.ctypes.data, l.ctypes.data, r.ctypes.data, st.ctypes.data,
                                          bv.ctypes.data, sp.ctypes.data, npred.ctypes.data, p, C.byref(h)))
        super().__init__(h, p)


def from_param_dict(m: dict) -> Model:
    """Build a device model from the plain parameter dict the tests and bench.py use
    (same fields as the flat R arrays; see the loaders in include/machisplin_hip.h)."""
    k = m["kind"]
    if k == "lm":
        return Gam(m["coef"])
    if k == "nnet":
        return Nnet(m["wts"], m["p"], m["size"], m["y_scale"], m["y_shift"])
    if k == "earth":
        return Earth(m["coef"], m["dirs"], m["cuts"])
    if k == "svr":
        return Ksvm(m["alpha"], m["sv"], m["b"], m["sigma"], m["x_center"], m["x_scale"], m["y_center"], m["y_scale"])
    if k == "gbm":
        return Gbm(m["init_f"], m["tree_offsets"], m["split_var"], m["split_val"], m["left"], m["right"],
                   m["missing"], m["p"])
    if k == "rf":
        return RandomForest(m["tree_offsets"], m["left"], m["right"], m["status"], m["best_var"], m["split"],
                            m["node_pred"], m["p"])
    raise ValueError(k)


def _window(stack: RasterStack, window):
    g = stack.geom
    return window if window is not None else (0, g.nrow, 0, g.ncol)


def _out(stack, window, out):
    import torch
    r0, r1, c0, c1 = window
    if out is None:
        out = torch.empty((r1 - r0, c1 - c0), dtype=torch.float64, device=stack.planes.device)
    if out.dtype != torch.float64 or not out.is_cuda or out.dim() != 2 or out.stride(1) != 1 \
            or tuple(out.shape) != (r1 - r0, c1 - c0):
        raise ValueError("out must be a float64 device tensor of the window's shape with unit column stride")
    return out


def predict(stack: RasterStack, model: Model, window=None, weight: float = 1.0, accumulate: bool = False,
            out=None, stream=None):
    """terra::predict(rast_stack, model) over the whole raster or a window (r0, r1, c0, c1).
    ``out = pred * weight`` or, with accumulate, ``out += pred * weight`` (V73:471/475 ...)."""
    import torch
    if model.p != stack.n_layers + 2:
        raise ValueError("model expects p = layers + 2 predictors (covariates, LONG, LAT)")
    window = _window(stack, window)
    out = _out(stack, window, out)
    g, s = stack.geom.c_struct(), stack.c_struct()
    st = stream if stream is not None else torch.cuda.current_stream(out.device).cuda_stream
    _lib.check(_lib.lib().mhs_predict_dev(model._h, C.byref(g), C.byref(s), *window, float(weight),
                                          int(bool(accumulate)), out.data_ptr(), out.stride(0), st))
    return out


def ensemble_predict(stack: RasterStack, models, weights, wt_total: float, window=None, out=None, stream=None):
    """The Step-2 raster loop (V73:447-619): ``(((p1 w1) + p2 w2) + ...) / wt_total`` with the
    models in ``mods.run`` order, the rounded kept weights, and the UNROUNDED total."""
    import torch
    window = _window(stack, window)
    out = _out(stack, window, out)
    n = len(models)
    if n == 0 or n != len(weights):
        raise ValueError("need one weight per model")
    hs = (C.c_void_p * n)(*[m._h for m in models])
    ws = (C.c_double * n)(*[float(w) for w in weights])
    g, s = stack.geom.c_struct(), stack.c_struct()
    st = stream if stream is not None else torch.cuda.current_stream(out.device).cuda_stream
    _lib.check(_lib.lib().mhs_ensemble_predict_dev(hs, ws, n, float(wt_total), C.byref(g), C.byref(s), *window,
                                                   out.data_ptr(), out.stride(0), st))
    return out


def members_predict(stack: RasterStack, models, weights, window=None, accumulate: bool = False, out=None, stream=None):
    """``out (+)= sum_k w_k pred_k`` over the window, members in order: the accumulation lines of the Step-2 loop
    (V73:471 ... 605) without the final division.  Consecutive gam / nnet / earth members share one pass over the planes
    (bit-identical to calling :func:`predict` member by member)."""
    import torch
    window = _window(stack, window)
    out = _out(stack, window, out)
    n = len(models)
    if n == 0 or n != len(weights):
        raise ValueError("need one weight per model")
    hs = (C.c_void_p * n)(*[m._h for m in models])
    ws = (C.c_double * n)(*[float(w) for w in weights])
    g, s = stack.geom.c_struct(), stack.c_struct()
    st = stream if stream is not None else torch.cuda.current_stream(out.device).cuda_stream
    _lib.check(_lib.lib().mhs_members_predict_dev(hs, ws, n, C.byref(g), C.byref(s), *window, int(bool(accumulate)),
                                                  out.data_ptr(), out.stride(0), st))
    return out


def fit_reserve_cus(n_cus: int) -> int:
    """mhs_fit_reserve_cus: while n_cus > 0 the first tree / ksvm member of every ensemble call on a large window
    leaves n_cus compute units free for a concurrent Tps fit (results unchanged).  Returns the previous setting."""
    _lib.init()
    prev = C.c_int(0)
    _lib.check(_lib.lib().mhs_fit_reserve_cus(int(n_cus), C.byref(prev)))
    return int(prev.value)


def select_weights(p_opt, labels="bgnmrv"):
    """V73:336-362 / 375-392: keep model k iff round(p_k, 2) > 0.05 * sum(p); the kept
    weight is round(p_k, 2); the divisor stays the unrounded sum over ALL candidates."""
    p_opt = np.asarray(p_opt, dtype=np.float64)
    tot = float(p_opt.sum())
    kept, wts = "", []
    for lab, pk in zip(labels, p_opt):
        r = float(np.round(pk, 2))
        if r > 0.05 * tot:
            kept += lab
            wts.append(r)
    return kept, wts, tot
