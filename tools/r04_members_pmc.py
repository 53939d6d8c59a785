"""Torch-free driver of the ensemble members for rocprofv3 passes (round 4: also the fused small members and the reference's
own rasters):
    python tools/r04_members_pmc.py [side=10000] [members=brv] [rasters=8d|bundled] [outdir]
members: b gbm, r randomForest, v ksvm (each on its own), s = the fused small members (gam + nnet + earth in one call).
rasters: 8d = SURVEY 8d planes; bundled = slope / TWI from the bundled overviews (tests/golden/cfg1_extdata.npz, mirrored
mosaic), alt synthetic.
cfg3's synthetic models (5 000 stations, 10 000 gbm trees, 500 forest trees, ~3 000 support vectors) over cfg3's float32
planes (the BASELINE generator evaluated with numpy), each member on its own through the host-pointer entry point
mhs_ensemble_predict (one band).  Writes the units the counters are divided by to gpurun_out/r3/pmc/units.json."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as m  # noqa: E402
from machisplin_amd import _lib, synth  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
which = sys.argv[2] if len(sys.argv) > 2 else "brv"
rasters = sys.argv[3] if len(sys.argv) > 3 else "8d"
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
outdir = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "r4", "pmc_" + rasters)
os.environ.setdefault("MHS_HOST_BANDS", "1")
m.init()
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3                      # cfg3's rasters (synth.covariates, here with numpy and cached between the passes)
cache = "/tmp/r04_pmc_planes_%d.npy" % side
if os.path.exists(cache):
    planes = np.load(cache)
else:
    rng = np.random.default_rng(seed + 7)
    col = (np.arange(side, dtype=np.float64) / side)[None, :]
    row = (np.arange(side, dtype=np.float64) / side)[:, None]
    planes = np.empty((3, side, side), dtype=np.float32)
    for k in range(3):
        planes[k] = synth._cov_layer(rng, col, row, k, np).astype(np.float32)
    np.save(cache, planes)
if rasters == "bundled":
    d = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_extdata.npz"))

    def mosaic(a):
        a = a.astype(np.float32)
        a[a == -32768] = np.nan
        ny, nx = -(-side // a.shape[0]), -(-side // a.shape[1])
        rws = []
        for iy in range(ny):
            t = a[::-1] if iy & 1 else a
            rws.append(np.concatenate([t[:, ::-1] if ix & 1 else t for ix in range(nx)], axis=1))
        return np.ascontiguousarray(np.concatenate(rws, axis=0)[:side, :side])
    smooth = planes
    planes = planes.copy()
    planes[1], planes[2] = mosaic(d["slope"]), mosaic(d["TWI"])
else:
    smooth = planes
xy, rows, cols, uv = synth.stations(g, 5000, seed)
X = np.column_stack([smooth[:, rows, cols].T.astype(np.float64), xy])      # the models of the bench: fitted on the 8d planes
y = synth.response(X, uv, seed)
small = "s" in which
params = synth.ensemble_params(X, y, seed, which=which.replace("s", "")) if which.replace("s", "") else []
small_params = synth.ensemble_params(X, y, seed, which="gnm") if small else []
out = np.empty((side, side))
st = _lib.Stack(planes.ctypes.data, 3, _lib.F32, side * side, side, float("nan"))
gs = g.c_struct()
for prm in params:
    mod = m.models.from_param_dict(prm)
    hs = (C.c_void_p * 1)(mod._h)
    ws = (C.c_double * 1)(1.0)
    for _ in range(2):
        _lib.check(_lib.lib().mhs_ensemble_predict(hs, ws, 1, 1.0, C.byref(gs), C.byref(st), 0, side, 0, side, out.ctypes.data))
    print(prm["kind"], float(np.nanmean(out)), flush=True)
if small_params:
    mods = [m.models.from_param_dict(p) for p in small_params]
    hs = (C.c_void_p * 3)(*[q._h for q in mods])
    ws = (C.c_double * 3)(0.2, 0.1, 0.2)
    for _ in range(2):
        _lib.check(_lib.lib().mhs_ensemble_predict(hs, ws, 3, 1.0, C.byref(gs), C.byref(st), 0, side, 0, side, out.ctypes.data))
    print("small members", float(np.nanmean(out)), flush=True)

units = {"side": side, "rasters": rasters}
for prm in params:
    if prm["kind"] == "gbm": units["gbm"] = float(len(prm["tree_offsets"]) - 1)
    if prm["kind"] == "svr": units["svr"] = float(prm["sv"].shape[0])
    if prm["kind"] == "rf":
        off, total = prm["tree_offsets"], 0
        for t in range(len(off) - 1):
            o, cnt = int(off[t]), int(off[t + 1] - off[t])
            L, R, stt = prm["left"][o:o + cnt] - 1, prm["right"][o:o + cnt] - 1, prm["status"][o:o + cnt]
            d = np.zeros(cnt, dtype=np.int64)
            for kk in np.flatnonzero(stt != -1):
                d[L[kk]] = d[R[kk]] = d[kk] + 1
            total += int(d.max())
        units["rf"] = float(total)
        units["rf_trees"] = float(len(off) - 1)
import json
os.makedirs(outdir, exist_ok=True)
with open(os.path.join(outdir, "units.json"), "w") as f:
    json.dump(units, f)
