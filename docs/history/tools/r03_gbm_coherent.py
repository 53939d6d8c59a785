"""Round 3: gbm_coherent_kernel against the tree-order row-tile kernel on cfg3's model (10 000 trees) over side x side cells of
(a) the BASELINE synthetic rasters, (b) the same rasters with white noise of a given amplitude added to every covariate
(spatially incoherent: the worst case for the coherent kernel).
    python tools/r03_gbm_coherent.py [side=8000] [reps=3]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import synth

m.init()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
xy, rows, cols, uv = synth.stations(g, 5000, seed)
cov_at = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov_at, xy])
y = synth.response(X, uv, seed)
prm = synth.ensemble_params(X, y, seed, which="b")[0]
mod = m.models.from_param_dict(prm)
out = torch.empty((side, side), dtype=torch.float64, device="cuda")
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
for noise in (0.0, 0.01, 0.1, 1.0):
    pl = planes
    if noise > 0:
        pl = planes.clone()
        for k in range(3):
            span = float(planes[k].max() - planes[k].min())
            pl[k] += (torch.rand(planes[k].shape, device="cuda", generator=gen) - 0.5) * (noise * span)
    stack = m.RasterStack(g, pl, nodata)
    ref = None
    for name, env in (("coherent", {}), ("tree order, row tiles", {"MHS_GBM_NO_COHERENT": "1"})):
        for k, v in env.items(): os.environ[k] = v
        m.predict(stack, mod, out=out); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            t0 = time.time(); m.predict(stack, mod, out=out); torch.cuda.synchronize(); best = min(best, time.time() - t0)
        for k in env: del os.environ[k]
        note = ""
        if ref is None: ref = out.clone()
        else: note = f"  max |diff| / max |pred| = {float((out - ref).abs().max() / ref.abs().max()):.1e}"
        print(f"noise {noise:5.3f} of the range  {name:24s} {best*1e3:9.2f} ms -> 1e8 cells: {best*1e8/(side*side)*1e3:8.1f} ms{note}", flush=True)
