"""Round 4, CPU study for the wave-slice forest kernel (rf_walk_ws_kernel): for a sample of waves (64 columns x 4 adjacent rows)
of cfg3's grid and cfg3's 500-tree forest, descend every tree while its split is uniform over the wave (the lane = tree
prefix of rf_prefix_entries) and report the size of the SUBTREE below the entry node -- what a wave would have to stage
privately -- and the levels its cells still walk.  Rasters: the SURVEY 8d planes, the bundled TWI / slope overviews
(tests/golden/cfg1_extdata.npz, mirrored), 8d + white noise.
    python tools/r04_rf_slice_sim.py [n_waves]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from machisplin_amd import synth  # noqa: E402

SIDE, N, LAYERS = 10000, 5000, 3
SEED = synth.BASE_SEED + 3


class NP:          # numpy stand-in for the torch module _cov_layer expects
    sin = staticmethod(np.sin)


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
    n_waves = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    TW = int(sys.argv[2]) if len(sys.argv) > 2 else 64          # wave tile: TW columns x TH rows (TW x TH = 256 cells)
    TH = int(sys.argv[3]) if len(sys.argv) > 3 else 256 // TW   # (a third argument: rows, e.g. 240 16 = a block of 15 wave tiles)
    geom = synth.grid(SIDE, SIDE)
    xy, rows, cols, uv = synth.stations(geom, N, SEED)
    cov = np.column_stack(planes_at(rows, cols))
    X = np.column_stack([cov, xy])
    y = synth.response(X, uv, SEED)
    prm = synth.rf_params(X, y, SEED)
    off = prm["tree_offsets"]
    left, right, var, thr = prm["left"] - 1, prm["right"] - 1, prm["best_var"] - 1, prm["split"]
    term = prm["status"] == -1
    nt = len(off) - 1
    # subtree sizes (nodes) per node, depth
    size = np.ones(len(left), dtype=np.int64)
    for t in range(nt):
        o, e = off[t], off[t + 1]
        for k in range(e - 1, o - 1, -1):          # children are created after their parent
            if not term[k]:
                size[k] = 1 + size[o + left[k]] + size[o + right[k]]
    print(f"forest: {nt} trees, {np.diff(off).mean():.0f} nodes per tree (max {np.diff(off).max()})")

    bundled = None
    gpath = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "cfg1_extdata.npz")
    if os.path.exists(gpath):
        z = np.load(gpath)
        print("bundled keys:", list(z.keys()))
        bundled = z

    rng = np.random.default_rng(5)
    for name, noise in (("8d planes", 0.0), ("8d + 1% noise", 0.01), ("8d + 10% noise", 0.1), ("bundled", None)):
        if noise is None and bundled is None:
            continue
        sizes, rest, terminal, plens = [], [], 0, []
        for w in range(n_waves):
            r0 = int(rng.integers(0, SIDE // TH)) * TH
            c0 = int(rng.integers(0, SIDE // TW)) * TW
            rr, cc = np.meshgrid(np.arange(r0, r0 + TH), np.arange(c0, c0 + TW), indexing="ij")
            if noise is None:
                arrs = [k for k in bundled.keys() if bundled[k].ndim == 2][:2]
                pl = []
                for k, (lo, hi) in zip(arrs, ((-207.0, 152.0), (-1.0, 877.0))):
                    a = bundled[k]
                    ri, ci = rr % (2 * a.shape[0]), cc % (2 * a.shape[1])
                    ri = np.where(ri >= a.shape[0], 2 * a.shape[0] - 1 - ri, ri)
                    ci = np.where(ci >= a.shape[1], 2 * a.shape[1] - 1 - ci, ci)
                    pl.append(a[ri, ci].astype(np.float64))
                syn = planes_at(rr, cc)
                # order of cfg3's planes: alt, slope, TWI; the bundled arrays replace slope / TWI by name when recognisable
                cv = [syn[0], pl[1] if len(pl) > 1 else syn[1], pl[0]]
            else:
                cv = planes_at(rr, cc, noise, w * 7)
            Xw = np.column_stack([c.ravel() for c in cv] + [geom.x_from_col(cc.ravel()), geom.y_from_row(rr.ravel())])
            mn, mx = Xw.min(0), Xw.max(0)
            for t in range(nt):
                o = off[t]
                k = o
                pl = 0
                while not term[k]:
                    v, s = var[k], thr[k]
                    if mx[v] <= s:
                        k = o + left[k]; pl += 1
                    elif mn[v] > s:
                        k = o + right[k]; pl += 1
                    else:
                        break
                plens.append(pl)
                if term[k]:
                    terminal += 1
                    continue
                sizes.append(size[k])
                # levels the wave's cells still walk below k (deepest leaf reached by any cell)
                cur = np.full(Xw.shape[0], k)
                lv = 0
                while True:
                    act = ~term[cur]
                    if not act.any():
                        break
                    go_r = Xw[np.arange(Xw.shape[0]), var[cur]] > thr[cur]
                    nxt = np.where(go_r, o + right[cur], o + left[cur])
                    cur = np.where(act, nxt, cur)
                    lv += 1
                rest.append(lv)
        sizes, rest = np.array(sizes), np.array(rest)
        tot = n_waves * nt
        q = np.percentile(sizes, [50, 75, 90, 95, 99]) if sizes.size else []
        print(f"tile {TW} x {TH}: prefix levels mean {np.mean(plens):.2f}; levels walked per (wave, tree) incl. skipped trees {rest.sum() / tot:.2f}")
        print(f"{name:16s}: terminal at the prefix {100 * terminal / tot:5.1f} %; subtree nodes mean {sizes.mean():7.1f} median/75/90/95/99 {q};"
              f"  <=32: {100 * (sizes <= 32).mean():.1f} %  <=64: {100 * (sizes <= 64).mean():.1f} %  <=128: {100 * (sizes <= 128).mean():.1f} %"
              f"  <=256: {100 * (sizes <= 256).mean():.1f} %;  levels left mean {rest.mean():.2f} max {rest.max()};"
              f"  bytes staged per (wave, tree) at 8 B/node, whole 64-node loads: {8 * 64 * np.ceil(sizes / 64).sum() / tot:.0f}")


if __name__ == "__main__":
    main()
