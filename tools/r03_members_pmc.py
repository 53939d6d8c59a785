"""Torch-free driver of the three heavy ensemble members (gbm, randomForest, ksvm) for rocprofv3 passes:
    python tools/r03_members_pmc.py [side=4000] [members=brv]
cfg3-shaped synthetic models (5 000 stations, 10 000 gbm trees, 500 forest trees, ~3 000 support vectors) over numpy-made
float32 planes, each member on its own through the host-pointer entry point mhs_ensemble_predict (one band)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as m  # noqa: E402
from machisplin_amd import _lib, synth  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
which = sys.argv[2] if len(sys.argv) > 2 else "brv"
os.environ.setdefault("MHS_HOST_BANDS", "1")
m.init()
g = synth.grid(side, side)
rng = np.random.default_rng(7)
col = (np.arange(side) + 0.5) / side
row = (np.arange(side) + 0.5) / side
planes = np.empty((3, side, side), dtype=np.float32)
for k in range(3):
    a, b, c = rng.uniform(2, 9, 3)
    planes[k] = (np.sin(a * col)[None, :] * np.cos(b * row)[:, None] + 0.3 * np.sin(c * (col[None, :] + row[:, None]))
                 + 0.05 * rng.standard_normal((side, side))).astype(np.float32) * 100 + 300
xy, rows, cols, uv = synth.stations(g, 5000, 11)
X = np.column_stack([planes[:, rows, cols].T.astype(np.float64), xy])
y = synth.response(X, uv, 11)
params = synth.ensemble_params(X, y, 11, which=which)
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
