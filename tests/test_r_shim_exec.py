"""The R `.Call()` shim EXECUTED (integration/r/src/machisplin_shim.c), against a stand-in R runtime: tests/rstub/rstub.c
implements the documented subset of R's C API the shim uses (REALSXP / INTSXP / VECSXP / EXTPTRSXP objects, Rf_error's
longjmp, R_alloc, external-pointer finalizers).  The shim + stub are compiled here with gcc, linked against
libmachisplin_hip.so, and every mhsr_* entry point on the hot path is driven through ctypes with fake SEXPs:

  * CPU: it links, exports what backend_hip.R / multi_gpu.R call, and a failing mhs_* status comes back as an R error
    (no crash, transient storage reclaimed);
  * GPU: its planes equal the direct C-ABI call bit for bit (fields::Tps / terra::interpolate replacements, the six
    loaders + the Step-2 raster loop, Step 3 + 4 in one call, tiles.merge, and the two multi-device calls), and
    releasing a handle runs its finalizer.

What this cannot show is R itself: its garbage collector, PROTECT discipline (a counter here) and terra's objects."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "r", "src", "machisplin_shim.c")
STUB = os.path.join(ROOT, "tests", "rstub", "rstub.c")
REALSXP, INTSXP, VECSXP, EXTPTRSXP = 14, 13, 19, 22


@pytest.fixture(scope="module")
def R(tmp_path_factory):
    """The shim + stub runtime as one shared object, loaded AFTER the product library (one HIP runtime in the process)."""
    from machisplin_amd import _lib
    _lib.load()
    out = str(tmp_path_factory.mktemp("rstub") / "libmhsr.so")
    libdir = os.path.join(ROOT, "machisplin_amd")
    cmd = ["gcc", "-shared", "-fPIC", "-O1", "-std=gnu99", "-Wall", "-Wextra", "-Werror=implicit-function-declaration",
           "-I", os.path.join(ROOT, "integration", "r", "check"), "-I", os.path.join(ROOT, "include"), "-o", out, STUB, SHIM,
           "-L", libdir, "-lmachisplin_hip", "-Wl,-rpath," + libdir]
    pr = subprocess.run(cmd, capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    return Rstub(C.CDLL(out))


class Rstub:
    """A few lines of 'R': make SEXPs from numpy, .Call a shim function, read results back."""

    def __init__(self, lib):
        self.lib = lib
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
        for name, res, args in [("rstub_real", vp, [vp, C.c_ssize_t]), ("rstub_real_matrix", vp, [vp, C.c_int, C.c_int]),
                                ("rstub_int", vp, [vp, C.c_ssize_t]), ("rstub_int_matrix", vp, [vp, C.c_int, C.c_int]),
                                ("rstub_list", vp, [C.c_ssize_t]), ("rstub_list_set", None, [vp, C.c_ssize_t, vp]),
                                ("rstub_list_get", vp, [vp, C.c_ssize_t]), ("rstub_null", vp, []), ("rstub_na_real", C.c_double, []),
                                ("rstub_type", C.c_int, [vp]), ("rstub_len", C.c_ssize_t, [vp]), ("rstub_nrow", C.c_int, [vp]),
                                ("rstub_ncol", C.c_int, [vp]), ("rstub_data", vp, [vp]), ("rstub_extptr", vp, [vp]),
                                ("rstub_last_error", C.c_char_p, []), ("rstub_finalizers_run", C.c_int, []),
                                ("rstub_protect_balance", C.c_int, []), ("rstub_release", None, [vp]),
                                ("rstub_call", vp, [vp, C.c_int, C.POINTER(vp)])]:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        self.NULL = lib.rstub_null()
        self.NA = lib.rstub_na_real()

    def num(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        return self.lib.rstub_real(a.ctypes.data, a.size)

    def mat(self, a):
        """an R matrix (column-major) from a 2-D numpy array"""
        a = np.asarray(a, dtype=np.float64)
        f = np.asfortranarray(a)
        return self.lib.rstub_real_matrix(f.ctypes.data, a.shape[0], a.shape[1])

    def int(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32).ravel()
        return self.lib.rstub_int(a.ctypes.data, a.size)

    def imat(self, a):
        a = np.asarray(a, dtype=np.int32)
        f = np.asfortranarray(a)
        return self.lib.rstub_int_matrix(f.ctypes.data, a.shape[0], a.shape[1])

    def list(self, items):
        l = self.lib.rstub_list(len(items))
        for i, it in enumerate(items):
            self.lib.rstub_list_set(l, i, it)
        return l

    def call(self, name, *args):
        fn = C.cast(getattr(self.lib, name), C.c_void_p)
        arr = (C.c_void_p * max(len(args), 1))(*args)
        out = self.lib.rstub_call(fn, len(args), arr)
        if not out:
            raise RuntimeError(self.lib.rstub_last_error().decode(errors="replace"))
        assert self.lib.rstub_protect_balance() == 0, name        # every PROTECT has its UNPROTECT on the normal path
        return out

    def values(self, s):
        t, n = self.lib.rstub_type(s), self.lib.rstub_len(s)
        if t == REALSXP:
            v = np.ctypeslib.as_array(C.cast(self.lib.rstub_data(s), C.POINTER(C.c_double)), shape=(n,)).copy()
        elif t == INTSXP:
            v = np.ctypeslib.as_array(C.cast(self.lib.rstub_data(s), C.POINTER(C.c_int)), shape=(n,)).copy()
        elif t == VECSXP:
            return [self.lib.rstub_list_get(s, i) for i in range(n)]
        else:
            raise TypeError(t)
        if self.lib.rstub_nrow(s):
            v = v.reshape((self.lib.rstub_ncol(s), self.lib.rstub_nrow(s))).T      # column-major -> numpy
        return v

    def geom(self, g):
        return self.num([g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol])


def test_shim_links_and_exports_what_the_r_files_call(R):
    r = "\n".join(open(os.path.join(ROOT, "integration", "r", "R", n)).read() for n in ("backend_hip.R", "multi_gpu.R"))
    names = sorted(set(re.findall(r'\.Call\(\s*"(mhsr_\w+)"', r)))
    assert len(names) >= 18 and "mhsr_mltps_grid_multi" in names and "mhsr_tiles_units_multi" in names
    for n in names:
        assert hasattr(R.lib, n), n


def test_a_failing_status_becomes_an_r_error_not_a_crash(R):
    import torch
    if torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="device index out of range"):
            R.call("mhsr_init", R.int([99]))
    else:
        with pytest.raises(RuntimeError, match="no HIP device"):
            R.call("mhsr_init", R.int([0]))
    # argument checks of the shim itself unwind the same way (and R_alloc storage is reclaimed by the stub's .Call)
    with pytest.raises(RuntimeError, match="n must be 1..16"):
        R.call("mhsr_init_devices", R.int([0]), R.NULL)
    with pytest.raises(RuntimeError, match="1..12 predictors"):
        R.call("mhsr_svr_fit", R.mat(np.zeros((5, 13))), R.num(np.zeros(5)), R.num([1.0]), R.num([1.0]), R.num([0.1]), R.num([1e-3]))


# ------------------------------------------------------------------------------------------------- GPU --
def _load_members(R, params):
    """what .mhs_model does in R: flat arrays out of the fitted objects into the loaders"""
    hs = []
    for m in params:
        k = m["kind"]
        if k == "lm":
            hs.append(R.call("mhsr_lm_load", R.num(m["coef"])))
        elif k == "nnet":
            hs.append(R.call("mhsr_nnet_load", R.num(m["wts"]), R.int([m["p"]]), R.int([m["size"]]), R.num([m["y_scale"]]), R.num([m["y_shift"]])))
        elif k == "earth":
            hs.append(R.call("mhsr_earth_load", R.num(m["coef"]), R.int(np.asarray(m["dirs"]).ravel()), R.num(np.asarray(m["cuts"]).ravel()),
                             R.int([np.asarray(m["dirs"]).shape[1]])))
        elif k == "svr":
            hs.append(R.call("mhsr_svr_load", R.num(m["alpha"]), R.num(np.asarray(m["sv"]).ravel()), R.int([np.asarray(m["sv"]).shape[1]]),
                             R.num([m["b"]]), R.num([m["sigma"]]), R.num(m["x_center"]), R.num(m["x_scale"]), R.num([m["y_center"]]),
                             R.num([m["y_scale"]])))
        elif k == "gbm":
            hs.append(R.call("mhsr_gbm_load", R.num([m["init_f"]]), R.num(m["tree_offsets"]), R.int(m["split_var"]), R.num(m["split_val"]),
                             R.int(m["left"]), R.int(m["right"]), R.int(m["missing"]), R.int([m["p"]])))
        elif k == "rf":
            hs.append(R.call("mhsr_rf_load", R.num(m["tree_offsets"]), R.int(m["left"]), R.int(m["right"]), R.int(m["status"]),
                             R.int(m["best_var"]), R.num(m["split"]), R.num(m["node_pred"]), R.int([m["p"]])))
    return hs


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_shim_planes_equal_the_direct_c_abi_calls_bit_for_bit(R, hip):
    import torch
    from machisplin_amd import multi, synth
    R.call("mhsr_init", R.int([0]))
    g = synth.grid(160, 208)
    planes, nodata = synth.covariates(g, 3, 31, dtype="f64")
    xy, rows, cols, uv = synth.stations(g, 320, 31)
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().T
    X = np.column_stack([cov, xy])
    resp = synth.response(X, uv, 31)
    params = synth.ensemble_params(X, resp, 31, n_gbm_trees=120, n_rf_trees=6)
    models = [hip.models.from_param_dict(p) for p in params]
    _, weights, wt_total = hip.models.select_weights(synth.OPTX_WEIGHTS)
    stack = hip.RasterStack(g, planes, nodata)
    values = planes.cpu().numpy().reshape(3, -1).T                  # terra::values(covar.ras): ncell x C
    before = R.lib.rstub_finalizers_run()

    # fields::Tps / terra::interpolate (V73:751-753)
    resid = synth.tps_residual(uv, 31)
    fit = hip.Tps(xy, resid)
    t = R.call("mhsr_tps_fit", R.mat(xy), R.num(resid), R.num([R.NA]), R.int([0]))
    got = R.values(R.call("mhsr_tps_predict_grid", t, R.geom(g), R.int([0, g.nrow, 0, g.ncol])))
    assert np.array_equal(got.reshape(g.nrow, g.ncol), hip.interpolate(g, fit).cpu().numpy())
    pts = R.values(R.call("mhsr_tps_predict_points", t, R.mat(xy[:50])))
    assert np.array_equal(pts, fit.predict(xy[:50]))
    # several fits in one call (the tiles of V73:690-738): a list of external pointers, NULL where a fit failed
    sets = [(xy[:120], resid[:120]), (xy[100:330], resid[100:330]), (np.column_stack([np.linspace(0, 1, 20)] * 2), np.ones(20))]
    many = R.values(R.call("mhsr_tps_fit_many", R.list([R.mat(a) for a, _ in sets]), R.list([R.num(b) for _, b in sets]), R.num([R.NA]), R.int([0])))
    want = hip.tps.fit_many([a for a, _ in sets], [b for _, b in sets])
    assert want[2] is None and R.lib.rstub_type(many[2]) != EXTPTRSXP          # collinear stations: NULL in the list
    for k in (0, 1):
        assert R.lib.rstub_type(many[k]) == EXTPTRSXP
        assert np.array_equal(R.values(R.call("mhsr_tps_predict_points", many[k], R.mat(xy[:50]))), want[k].predict(xy[:50]))

    # the six loaders + the Step-2 raster loop (V73:447-619) and the station predictions
    hs = _load_members(R, params)
    assert all(R.lib.rstub_type(h) == EXTPTRSXP and R.lib.rstub_extptr(h) for h in hs)
    pred = R.values(R.call("mhsr_ensemble_predict", R.list(hs), R.num(weights), R.num([wt_total]), R.geom(g), R.mat(values)))
    assert np.array_equal(pred.reshape(g.nrow, g.ncol), hip.ensemble_predict(stack, models, weights, wt_total).cpu().numpy(), equal_nan=True)
    for h, m in zip(hs, models):
        assert np.array_equal(R.values(R.call("mhsr_predict_points", h, R.mat(X[:64]))), m.predict_points(X[:64]))

    # Step 3 + 4 in one call (V73:636-897), reference-tiled
    res = hip.mltps.ensemble_residuals(models, weights, wt_total, X, resp)
    surf = R.values(R.call("mhsr_tps_surface", R.geom(g), R.mat(xy), R.num(res), R.num(X[:, 0]), R.int([64]), R.num([R.NA]), R.int([0])))
    want = hip.tps_residual_surface(g, xy, res, cov1_at_stations=X[:, 0], tile_edge=64).cpu().numpy()
    assert np.array_equal(surf.reshape(g.nrow, g.ncol), want, equal_nan=True)

    # machisplin.tiles.merge (V73:1392-1548)
    tl = hip.tiles.tiles_create(g, xy, out_ncol=2, out_nrow=2, feather_d=16)
    rng = np.random.default_rng(1)
    tiles = [rng.standard_normal((int(w[1] - w[0]), int(w[3] - w[2]))) for w in tl["win"]]
    merged = R.values(R.call("mhsr_tiles_merge", R.geom(g), R.list([R.num(t_.ravel()) for t_ in tiles]), R.imat(np.asarray(tl["win"]).T),
                             R.int([2]), R.int([2])))
    want = hip.tiles.tiles_merge(g, tl["win"], [torch.from_numpy(t_).cuda() for t_ in tiles], in_ncol=2, in_nrow=2).cpu().numpy()
    assert np.array_equal(merged.reshape(g.nrow, g.ncol), want, equal_nan=True)

    # the two multi-device calls, two slots on GPU 0 (multi_gpu.R: mhs_mltps_multi, mhs_tiles_mltps)
    assert R.values(R.call("mhsr_init_devices", R.int([2]), R.int([0, 0])))[0] == 2
    out = R.values(R.call("mhsr_mltps_grid_multi", R.list(hs), R.num(weights), R.num([wt_total]), R.geom(g), R.mat(values), R.mat(X), R.num(resp),
                          R.int([0]), R.num([R.NA]), R.int([0]), R.num([R.NA])))
    final, info = multi.mltps_grid_multi(g, planes.cpu().numpy(), nodata, models, weights, wt_total, X, resp)
    assert np.array_equal(R.values(out[0]).reshape(g.nrow, g.ncol), final, equal_nan=True)
    assert R.values(out[1])[0] == info["rsq_model"] and R.values(out[2])[0] == info["rsq_final"] and R.values(out[5])[0] == 2
    with pytest.raises(RuntimeError, match="X must be n x"):
        R.call("mhsr_mltps_grid_multi", R.list(hs), R.num(weights), R.num([wt_total]), R.geom(g), R.mat(values), R.mat(X[:, :3]), R.num(resp),
               R.int([0]), R.num([R.NA]), R.int([0]), R.num([R.NA]))
    units_py, units_r = [[None] * 4], []
    for t_ in range(4):
        sel = tl["dat"][t_]
        gt = tl["geom"][t_]
        tr, tc = hip.tiles.cells_from_xy(gt, xy[sel])
        Xt = np.column_stack([cov[sel], gt.x_from_col(tc), gt.y_from_row(tr)])
        units_py[0][t_] = {"models": models, "weights": weights, "wt_total": wt_total, "X": Xt, "resp": resp[sel]}
        units_r.append(R.list([R.list(hs), R.num(weights), R.num([wt_total]), R.mat(Xt), R.num(resp[sel])]))
    outs, rsq, _ = multi.tiles_units_multi(g, planes.cpu().numpy(), nodata, 2, 2, 16, units_py, 1, tile_edge=64)
    ru = R.values(R.call("mhsr_tiles_units_multi", R.geom(g), R.mat(values), R.int([2]), R.int([2]), R.num([16.0]), R.list(units_r), R.int([1]),
                         R.int([1]), R.int([64]), R.num([R.NA]), R.int([0])))
    assert np.array_equal(R.values(R.values(ru[0])[0]).reshape(g.nrow, g.ncol), outs[0], equal_nan=True)
    assert np.array_equal(R.values(ru[1]).T.reshape(1, 4, 2), rsq)
    R.call("mhsr_multi_trim")                        # multi_gpu.R: mhs_multi_trim -- the kept buffers go, the next call builds them again
    out2 = R.values(R.call("mhsr_mltps_grid_multi", R.list(hs), R.num(weights), R.num([wt_total]), R.geom(g), R.mat(values), R.mat(X), R.num(resp),
                           R.int([0]), R.num([R.NA]), R.int([0]), R.num([R.NA])))
    assert np.array_equal(R.values(out2[0]).reshape(g.nrow, g.ncol), final, equal_nan=True)
    multi.init_devices(1, [0])

    # what R's garbage collector does with unreachable handles: finalizers run, pointers are cleared
    for h in hs + [t]:
        R.lib.rstub_release(h)
    assert R.lib.rstub_finalizers_run() - before == len(hs) + 1
