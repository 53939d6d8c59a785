"""randomForest walk on a cfg5-shaped forest (20 000 stations, 5 covariates + x + y; trees of ~10^4 nodes, BIG form):
the COMPACT-form kernel (split nodes only in LDS) against the two BIG-form kernels (tree-major: MHS_RF_NO_COMPACT=1;
one batch per staging: + MHS_RF_BIG_PER_BATCH=1), same planes.
   python tools/rf_big_speed.py [trees] [rows]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as mhs  # noqa: E402
from machisplin_amd import synth  # noqa: E402

trees = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
mhs.init()
side = 20000
geom = synth.grid(rows, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(geom, 5, seed, dtype="f32")
stack = mhs.RasterStack(geom, planes, nodata)
xy, r, c, uv = synth.stations(geom, 20000, seed)
cov = planes[:, torch.from_numpy(r).cuda(), torch.from_numpy(c).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
y = synth.response(X, uv, seed)
t0 = time.perf_counter()
prm = synth.rf_params(X, y, seed, n_trees=trees)
print(f"forest: {trees} trees, max nodes {np.diff(prm['tree_offsets']).max()}, built in {time.perf_counter() - t0:.1f} s", flush=True)
model = mhs.models.from_param_dict(prm)
res = {}
for name, envs in (("compact", ()), ("tree-major", ("MHS_RF_NO_COMPACT",)), ("per-batch", ("MHS_RF_NO_COMPACT", "MHS_RF_BIG_PER_BATCH"))):
    for env in envs:
        os.environ[env] = "1"
    out = mhs.predict(stack, model)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = mhs.predict(stack, model, out=out)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    for env in envs:
        del os.environ[env]
    res[name] = out.clone()
    print(f"{name:11s}: {best * 1e3:8.1f} ms for {rows} x {side} cells x {trees} trees  "
          f"-> {best * 1e3 * (500 / trees) * (side / rows):8.0f} ms for the cfg5 grid and 500 trees", flush=True)
print("identical:", all(torch.equal(torch.nan_to_num(res[k]), torch.nan_to_num(res["per-batch"])) for k in res))
