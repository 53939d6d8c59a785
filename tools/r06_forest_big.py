"""randomForest walk on a cfg5-shaped forest (20 000 stations, 5 covariates + LONG + LAT; trees of ~12 000 nodes): the
block-subtree kernel (rf_walk_cbs_kernel, the default for such trees) against the whole-tree compact kernel
(MHS_RF_KERNEL=compact), each also without prefixes (MHS_RF_PLAIN=1), on the SURVEY 8d planes and on the same planes with
white noise; planes compared bit for bit.
   python tools/r06_forest_big.py [trees] [rows] [stations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from machisplin_amd import _lib as L  # noqa: E402
if os.environ.get("MHS_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["MHS_LIB"])
import machisplin_amd as mhs  # noqa: E402
from machisplin_amd import synth  # noqa: E402

trees = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
stations = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
mhs.init()
side = 20000
geom = synth.grid(rows, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(geom, 5, seed, dtype="f32")
xy, r, c, uv = synth.stations(geom, stations, seed)
cov = planes[:, torch.from_numpy(r).cuda(), torch.from_numpy(c).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
y = synth.response(X, uv, seed)
t0 = time.perf_counter()
prm = synth.rf_params(X, y, seed, n_trees=trees)
sizes = np.diff(prm["tree_offsets"])
print(f"forest: {trees} trees, nodes per tree {sizes.min()}..{sizes.max()}, built in {time.perf_counter() - t0:.1f} s", flush=True)
model = mhs.models.from_param_dict(prm)
gen = torch.Generator(device="cuda")
gen.manual_seed(7)
variants = [("8d planes", planes)]
for frac in (0.01, 0.1):
    noisy = planes.clone()
    for k in range(5):
        lo, hi = synth.COV_RANGES[k % len(synth.COV_RANGES)]
        noisy[k] += (torch.rand((rows, side), device="cuda", generator=gen) - 0.5) * (frac * (hi - lo))
    variants.append(("8d + %g %% noise" % (100 * frac), noisy))
scale = 1e8 / (rows * side) * (500 / trees)
for vname, pl in variants:
    stack = mhs.RasterStack(geom, pl, nodata)
    res = {}
    line = f"{vname:16s}"
    for name, env in (("cbs", {}), ("cbs plain", {"MHS_RF_PLAIN": "1"}), ("compact", {"MHS_RF_KERNEL": "compact"}),
                      ("compact plain", {"MHS_RF_KERNEL": "compact", "MHS_RF_PLAIN": "1"})):
        os.environ.update(env)
        try:
            out = mhs.predict(stack, model)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                out = mhs.predict(stack, model, out=out)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
        finally:
            for e in env:
                del os.environ[e]
        res[name] = out.clone()
        line += f" | {name} {best * 1e3 * scale:7.1f}"
    same = all(torch.equal(torch.nan_to_num(res[k]), torch.nan_to_num(res["compact"])) for k in res)
    print(line + f"   ms per 1e8 cells and 500 trees; identical: {same}", flush=True)
