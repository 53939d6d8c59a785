import os, sys, time
import numpy as np, torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, ROOT)
import machisplin_amd as m
from machisplin_amd import synth
m.init()
side = 8000
g10 = synth.grid(10000, 10000); seed = synth.BASE_SEED + 3
xy, rows, cols, uv = synth.stations(g10, 5000, seed)
X = np.column_stack([synth.covariates_at(g10, 3, seed, rows, cols), xy]); y = synth.response(X, uv, seed)
prm = synth.rf_params(X, y, seed); mod = m.models.from_param_dict(prm)
g = synth.grid(side, side)
base, nodata = synth.covariates(g10, 3, seed, dtype="f32", window=(0, side, 0, side))
stack = m.RasterStack(g, base, nodata)
out = torch.empty((side, side), dtype=torch.float64, device="cuda")
for name, env in (("full", {}), ("setup only", {"MHS_RF_LD_FLAGS": "128"}), ("setup only, no prefix (keys only)", {"MHS_RF_LD_FLAGS": "128", "MHS_RF_NO_PREFIX": "1"}),
                  ("walks + staging off", {"MHS_RF_LD_FLAGS": "6"}), ("walks + staging + loader / walker synchronisation off", {"MHS_RF_LD_FLAGS": "14"}),
                  ("staging + synchronisation off (walks on garbage)", {"MHS_RF_LD_FLAGS": "12"})):
    os.environ.update(env)
    m.predict(stack, mod, out=out); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.time(); m.predict(stack, mod, out=out); torch.cuda.synchronize(); best = min(best, time.time() - t0)
    for k in env: del os.environ[k]
    print(f"{name:40s} {best*1e3:8.2f} ms", flush=True)
