import time, numpy as np, torch, sys
sys.path.insert(0, '.'); import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import synth
m.init()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
stack = m.RasterStack(g, planes, nodata)
xy, rows, cols, uv = synth.stations(g, 5000, seed)
cov_at = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov_at, xy])
y = synth.response(X, uv, seed)
t0 = time.time(); params = synth.ensemble_params(X, y, seed); print("gen", time.time() - t0)
out = torch.empty((side, side), dtype=torch.float64, device="cuda")
for prm in params:
    mod = m.models.from_param_dict(prm)
    m.predict(stack, mod, out=out); torch.cuda.synchronize()
    t0 = time.time(); m.predict(stack, mod, out=out); torch.cuda.synchronize(); dt = time.time() - t0
    extra = ""
    if prm["kind"] == "gbm": extra = f"trees={len(prm['tree_offsets'])-1}"
    if prm["kind"] == "rf": extra = f"trees={len(prm['tree_offsets'])-1} nodes/tree={prm['tree_offsets'][-1]/(len(prm['tree_offsets'])-1):.0f}"
    if prm["kind"] == "svr": extra = f"nsv={len(prm['alpha'])} pairs/s={len(prm['alpha'])*side*side/dt/1e12:.3f}T"
    print(f"{prm['kind']:6s} {dt*1e3:9.2f} ms  {side*side/dt/1e6:9.1f} Mcells/s  -> 1e8 cells: {dt*1e8/(side*side):.3f} s  {extra}")
