import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as m
from machisplin_amd import synth
m.init()
side = 8000
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
xy, rows, cols, uv = synth.stations(g, 5000, seed)
cov_at = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov_at, xy])
y = synth.response(X, uv, seed)
mod = m.models.from_param_dict(synth.rf_params(X, y, seed))
stack = m.RasterStack(g, planes, nodata)
out = torch.empty((side, side), dtype=torch.float64, device="cuda")
def timed(env):
    for k, v in env.items(): os.environ[k] = v
    try:
        m.predict(stack, mod, out=out); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); m.predict(stack, mod, out=out); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best * 1e3
    finally:
        for k in env: del os.environ[k]
SUB = {"MHS_RF_KERNEL": "sub"}
for label, env in (("sub", dict(SUB)), ("sub, no level loops (wrong planes)", dict(SUB, MHS_RF_DIAG="4")),
                   ("sub, no level loops, no staging (wrong planes)", dict(SUB, MHS_RF_DIAG="6")), ("sub plain (whole trees)", dict(SUB, MHS_RF_PLAIN="1")),
                   ("ld (default)", {}), ("ld plain", {"MHS_RF_PLAIN": "1"})):
    print(f"{label:48s} {timed(env):7.2f} ms on {side}^2", flush=True)
