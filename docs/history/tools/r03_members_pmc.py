"""Torch-free driver of the three heavy ensemble members (gbm, randomForest, ksvm) for rocprofv3 passes:
    python tools/r03_members_pmc.py [side=4000] [members=brv]
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
os.environ.setdefault("MHS_HOST_BANDS", "1")
m.init()
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3                      # cfg3's rasters (synth.covariates, here with numpy and cached between the passes)
cache = "/tmp/r03_pmc_planes_%d.npy" % side
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
xy, rows, cols, uv = synth.stations(g, 5000, seed)
X = np.column_stack([planes[:, rows, cols].T.astype(np.float64), xy])
y = synth.response(X, uv, seed)
params = synth.ensemble_params(X, y, seed, which=which)
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

units = {"side": side}
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
import json
os.makedirs(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r3", "pmc"), exist_ok=True)
with open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r3", "pmc", "units.json"), "w") as f:
    json.dump(units, f)
