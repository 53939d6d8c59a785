"""Round 5: the price list of gbm_coherent_kernel's probe, re-measured.  For the cfg3 gbm (10 000 trees, synthetic structures or a
scikit-learn fit) on the 8d planes, the 8d planes + white noise and the reference's bundled rasters: the share of (wave tile,
tree) pairs with 0 / 1 / 2 / >= 3 straddling splits (numpy, on sampled 16 x 16 tiles) beside the times of the coherent kernel
(forced), the tree-order kernel and the default (probe's choice), ms per 1e8 cells.  A least-squares fit of
    coherent / tree-order = f0 c0 + f1 c1 + f2 c2 + sum_{n >= 3} f_n (c3 + d3 n)
over the variants gives the constants of the PROBE instantiation.
    python tools/r05_gbm_coherent_calib.py [side=4000] [fitted=0]"""
import os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import machisplin_amd as m
from machisplin_amd import synth

m.init()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
fitted = len(sys.argv) > 2 and sys.argv[2] == "1"
FULL = 10000
g = synth.grid(FULL, FULL)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
xy, rows, cols, uv = synth.stations(g, 5000, seed)
cov_at = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov_at, xy])
y = synth.response(X, uv, seed)
if fitted:
    from sklearn.ensemble import GradientBoostingRegressor
    from tests import modelgen
    gbr = GradientBoostingRegressor(n_estimators=10000, max_leaf_nodes=6, max_depth=None, learning_rate=0.001, subsample=0.5, random_state=1).fit(X, y)
    prm = modelgen.gbm_from_sklearn(gbr, X.shape[1])
else:
    prm = synth.ensemble_params(X, y, seed, which="b")[0]
mod = m.models.from_param_dict(prm)
var, val, off = np.asarray(prm["split_var"]), np.asarray(prm["split_val"]), np.asarray(prm["tree_offsets"])
split = var >= 0
tree_of = np.repeat(np.arange(len(off) - 1), np.diff(off))[split]
sv, st, nt = var[split], val[split], len(off) - 1

gw = synth.grid(side, side)
base = planes[:, :side, :side].contiguous()
del planes
variants = [("8d planes", base)]
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
for frac in (0.01, 0.03, 0.1, 0.3):
    noisy = base.clone()
    for k in range(3):
        lo, hi = synth.COV_RANGES[k]
        noisy[k] += (torch.rand((side, side), device="cuda", generator=gen) - 0.5) * (frac * (hi - lo))
    variants.append(("8d + %g %% noise" % (100 * frac), noisy))
fx = os.path.join(ROOT, "tests", "golden", "cfg1_extdata.npz")
if os.path.exists(fx):
    d = np.load(fx)
    def mosaic(a):
        a = a.astype(np.float32); a[a == -32768] = np.nan
        ny, nx = -(-side // a.shape[0]), -(-side // a.shape[1])
        rws = []
        for iy in range(ny):
            t = a[::-1] if iy & 1 else a
            rws.append(np.concatenate([t[:, ::-1] if ix & 1 else t for ix in range(nx)], axis=1))
        return np.ascontiguousarray(np.concatenate(rws, axis=0)[:side, :side])
    real = base.clone()
    real[1] = torch.from_numpy(mosaic(d["slope"])).cuda(); real[2] = torch.from_numpy(mosaic(d["TWI"])).cuda()
    variants.append(("bundled", real))

out = torch.empty((side, side), dtype=torch.float64, device="cuda")
scale = 1e8 / (side * side)
def timed(stack, env):
    for k in env: os.environ[k] = "1"
    try:
        m.predict(stack, mod, out=out); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); m.predict(stack, mod, out=out); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    finally:
        for k in env: del os.environ[k]
    return best * 1e3 * scale

rng = np.random.default_rng(5)
rowsA, ys = [], []
for name, pl in variants:
    hist = np.zeros(7)
    for w in range(200):
        r0 = int(rng.integers(0, side // 16)) * 16; c0 = int(rng.integers(0, side // 16)) * 16
        cv = pl[:, r0:r0 + 16, c0:c0 + 16].cpu().numpy().astype(np.float64).reshape(3, -1)
        rr, cc = np.meshgrid(np.arange(r0, r0 + 16), np.arange(c0, c0 + 16), indexing="ij")
        Xw = np.column_stack([cv[0], cv[1], cv[2], gw.x_from_col(cc.ravel()), gw.y_from_row(rr.ravel())])
        mn, mx = np.nanmin(Xw, 0), np.nanmax(Xw, 0)
        straddle = (mn[sv] < st) & (st <= mx[sv])
        nstr = np.bincount(tree_of, weights=straddle, minlength=nt).astype(int)
        hist += np.bincount(np.minimum(nstr, 6), minlength=7)[:7]
    f = hist / hist.sum()
    stack = m.RasterStack(gw, pl, float("nan"))
    tc, tt, td = timed(stack, ("MHS_GBM_FORCE_COHERENT",)), timed(stack, ("MHS_GBM_NO_COHERENT",)), timed(stack, ())
    n3 = (f[3:] * np.arange(3, 7)).sum()
    print(f"{name:18s} straddling splits per (tile, tree) 0/1/2/3+: {f[0]:.3f} {f[1]:.3f} {f[2]:.3f} {f[3:].sum():.3f} (sum n f_n, n >= 3: {n3:.3f})   "
          f"coherent {tc:7.1f}  tree order {tt:7.1f}  default {td:7.1f} ms per 1e8 cells   ratio {tc / tt:.3f}", flush=True)
    rowsA.append([f[0], f[1], f[2], f[3:].sum(), n3]); ys.append(tc / tt)
A, yv = np.array(rowsA), np.array(ys)
sol, res, rank, _ = np.linalg.lstsq(A, yv, rcond=None)
print("least squares  c0 %.3f  c1 %.3f  c2 %.3f  c3 %.3f  d3 %.3f   (ratio = f0 c0 + f1 c1 + f2 c2 + f3+ c3 + d3 sum n f_n)" % tuple(sol))
print("fitted ratios ", np.round(A @ sol, 3), " measured ", np.round(yv, 3))
