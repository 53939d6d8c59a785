"""Round 5, CPU study (round-4 verdict item 4): per 16 x 16-cell wave tile, what gbm's coherent kernel would cost relative to the
tree-order kernel (the probe's own price list: a tree without a straddling split 0.12, with n >= 1 straddling splits 0.85 + 0.26 n
of a tree-order evaluation), on the SURVEY 8d planes, on the reference's bundled TWI / slope overviews (mirrored; synthetic alt)
and on 8d + white noise.  The histogram says how much a PER-TILE choice (coherent where it is cheaper, tree order elsewhere) could
gain over today's per-window choice.     python tools/r05_gbm_tile_hist.py [n_tiles]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from machisplin_amd import synth
SIDE, N, LAYERS = 10000, 5000, 3
SEED = synth.BASE_SEED + 3
class NP: sin = staticmethod(np.sin)
def planes_at(rows, cols, noise=0.0, noise_seed=0):
    rng = np.random.default_rng(SEED + 7)
    out = []
    for k in range(LAYERS):
        z = synth._cov_layer(rng, np.asarray(cols, float) / SIDE, np.asarray(rows, float) / SIDE, k, NP)
        if noise > 0:
            lo, hi = synth.COV_RANGES[k]
            z = z + noise * (hi - lo) * np.random.default_rng(noise_seed + k).standard_normal(z.shape)
        out.append(z.astype(np.float32).astype(np.float64))
    return out
def main():
    n_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    geom = synth.grid(SIDE, SIDE)
    xy, rows, cols, uv = synth.stations(geom, N, SEED)
    X = np.column_stack([np.column_stack(planes_at(rows, cols)), xy])
    y = synth.response(X, uv, SEED)
    prm = synth.gbm_params(X, y, SEED)
    var, val = np.asarray(prm["split_var"]), np.asarray(prm["split_val"])
    off = np.asarray(prm["tree_offsets"])
    split = var >= 0
    tree_of = np.repeat(np.arange(len(off) - 1), np.diff(off))[split]
    sv, st = var[split], val[split]
    nt = len(off) - 1
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cfg1_extdata.npz"))
    arrs = [k for k in z.keys() if z[k].ndim == 2][:2]
    rng = np.random.default_rng(5)
    for name, noise in (("8d planes", 0.0), ("8d + 1% noise", 0.01), ("8d + 10% noise", 0.1), ("bundled", None)):
        costs = []
        for w in range(n_tiles):
            r0 = int(rng.integers(0, SIDE // 16)) * 16; c0 = int(rng.integers(0, SIDE // 16)) * 16
            rr, cc = np.meshgrid(np.arange(r0, r0 + 16), np.arange(c0, c0 + 16), indexing="ij")
            if noise is None:
                pl = []
                for k in arrs:
                    a = z[k]
                    ri, ci = rr % (2 * a.shape[0]), cc % (2 * a.shape[1])
                    ri = np.where(ri >= a.shape[0], 2 * a.shape[0] - 1 - ri, ri); ci = np.where(ci >= a.shape[1], 2 * a.shape[1] - 1 - ci, ci)
                    pl.append(a[ri, ci].astype(np.float64))
                syn = planes_at(rr, cc)
                cv = [syn[0], pl[1] if len(pl) > 1 else syn[1], pl[0]]
            else:
                cv = planes_at(rr, cc, noise, w * 7)
            Xw = np.column_stack([c.ravel() for c in cv] + [geom.x_from_col(cc.ravel()), geom.y_from_row(rr.ravel())])
            mn, mx = Xw.min(0), Xw.max(0)
            straddle = (mn[sv] < st) & (st <= mx[sv])
            nstr = np.bincount(tree_of, weights=straddle, minlength=nt)
            costs.append(np.where(nstr == 0, 0.12, 0.85 + 0.26 * nstr).mean())
        costs = np.array(costs)
        per_tile = np.minimum(costs, 1.0).mean()
        print(f"{name:16s}: coherent cost / tree-order cost per tile: mean {costs.mean():.2f}  median {np.median(costs):.2f}  "
              f"deciles {np.round(np.percentile(costs, [10, 25, 50, 75, 90]), 2)}  tiles cheaper coherent {100 * (costs < 1).mean():.0f} %;  "
              f"per-window choice min(mean, 1) = {min(costs.mean(), 1.0):.2f}, per-tile choice = {per_tile:.2f} of the tree-order kernel")
if __name__ == "__main__":
    main()
