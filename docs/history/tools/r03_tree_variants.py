"""Round 3: stand-alone times of the ksvm, gbm and randomForest kernel variants on a cfg3-shaped model set (5 000 stations,
10 000 gbm trees, 500 forest trees) over side x side cells.  Variants are selected with the library's environment
switches; every variant's plane is compared bit for bit with the first one's.
    python tools/r03_tree_variants.py [side=8000] [reps=3]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import synth

m.init()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shape = sys.argv[3] if len(sys.argv) > 3 else "cfg3"          # cfg5: 20 000 stations, 5 covariates (the forest's COMPACT form)
ncov, nst = (5, 20000) if shape == "cfg5" else (3, 5000)
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, ncov, seed, dtype="f32")
stack = m.RasterStack(g, planes, nodata)
xy, rows, cols, uv = synth.stations(g, nst, seed)
cov_at = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov_at, xy])
y = synth.response(X, uv, seed)
params = {p["kind"]: p for p in synth.ensemble_params(X, y, seed, which="r" if shape == "cfg5" else "brv")}
out = torch.empty((side, side), dtype=torch.float64, device="cuda")

def run(kind, env):
    for k, v in env.items(): os.environ[k] = v
    mod = m.models.from_param_dict(params[kind])
    m.predict(stack, mod, out=out); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); m.predict(stack, mod, out=out); torch.cuda.synchronize(); best = min(best, time.time() - t0)
    for k in env: del os.environ[k]
    return best, out.clone()

cfg5_variants = (("rf", [("split-node records, 4 walks on adjacent rows, prefix, early exit (default)", {}),
                         ("same, walks start at the root", {"MHS_RF_NO_PREFIX": "1"}),
                         ("walks at the root, every tree to its full depth", {"MHS_RF_FULL_DEPTH": "1"}),
                         ("4 walks a quarter of the grid apart, early exit", {"MHS_RF_FAR_WALKS": "1"}),
                         ("round 3 before the early exit", {"MHS_RF_FAR_WALKS": "1", "MHS_RF_FULL_DEPTH": "1"})]),)
for kind, variants in cfg5_variants if shape == "cfg5" else (("svr", [("row tiles: the LAT term and the wave's largest |x|^2 term once per wave and support vector (round 3)", {}),
                                ("lane per cell (round 2)", {"MHS_SVR_NO_ROWTILE": "1"})]),
                       ("gbm", [("row tiles (round 3)", {}), ("lane per cell (round 2)", {"MHS_GBM_NO_ROWTILE": "1"})]),
                       ("rf", [("three buffers, no barrier; walks on adjacent rows start where the wave's cells part ways and end at its deepest leaf (default)", {}),
                               ("two buffers + a barrier per tree, 5 walks, otherwise the same", {"MHS_RF_DOUBLE_BUFFER": "1"}),
                               ("two buffers, walks start at the root", {"MHS_RF_DOUBLE_BUFFER": "1", "MHS_RF_NO_PREFIX": "1"}),
                               ("two buffers, walks start at the root, every tree to its full depth", {"MHS_RF_DOUBLE_BUFFER": "1", "MHS_RF_NO_PREFIX": "1", "MHS_RF_FULL_DEPTH": "1"}),
                               ("the same with a lane's walks a fifth of the grid apart (round 3 before these changes)", {"MHS_RF_DOUBLE_BUFFER": "1", "MHS_RF_FAR_WALKS": "1", "MHS_RF_FULL_DEPTH": "1"}),
                               ("two buffers, 4 walks, prefix, early exit", {"MHS_RF_DOUBLE_BUFFER": "1", "MHS_RF_FOUR_WALKS": "1"}),
                               ("double-buffered, 5 walks, the compiler's loop (round 2)", {"MHS_RF_COMPILER_LOOP": "1"}),
                               ("split-node records, one buffer, 4 walks", {"MHS_RF_FORCE_COMPACT": "1"})])):
    ref = None
    for name, env in variants:
        dt, plane = run(kind, env)
        same = "" if ref is None else ("  == first" if torch.equal(torch.nan_to_num(plane), torch.nan_to_num(ref)) else
                                       "  max |diff| / max |first| = %.1e" % float((torch.nan_to_num(plane) - torch.nan_to_num(ref)).abs().max() / torch.nan_to_num(ref).abs().max()))
        if ref is None: ref = plane
        print(f"{kind:4s} {name:60s} {dt*1e3:9.2f} ms  -> 1e8 cells: {dt*1e8/(side*side)*1e3:8.1f} ms{same}", flush=True)
