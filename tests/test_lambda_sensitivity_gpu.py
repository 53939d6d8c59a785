"""What "parity unpinned" can cost in the spline (SURVEY.md 7 "lambda parity").

The one step of fields::Tps this repo cannot pin against R is the END of the GCV search: fields' golden section stops
at its own tolerance (gcv_mode="fields" restates it), a converged minimiser (gcv_mode="converged") stops elsewhere, and
an R build with another grid or tolerance lands somewhere between.  This test MEASURES, on the BASELINE station sets
(cfg1: the 813 bundled stations, global and in the reference's 2 x 3 tiles; cfg2: 2 000; cfg3: 5 000 synthetic stations),
    d_mode = max |S_fields - S_converged| / max |S|      d_lam = max |S(lambda (1 +- 1e-3)) - S(lambda)| / max |S|
(S = the residual surface, the quantity the spline contributes to the final raster), writes them to
gpurun_out/r3/lambda_sensitivity.json (quoted in DESIGN.md section 5) and asserts what must hold whatever the numbers
are: the surface moves LINEARLY and mildly with lambda (d_lam < 1e-3 for a 1e-3 change), and the two search modes
differ by no more than their lambda difference explains."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _sets(hip):
    from machisplin_amd import synth
    z = np.load(os.path.join(HERE, "golden", "tps_sampling813.npz"))
    xy = z["xy"]
    lo, hi = xy.min(0), xy.max(0)
    res = max(hi - lo) / 300
    g1 = hip.Geometry(float(lo[0]), float(hi[1]), float(res), float(res), int(np.ceil((hi[1] - lo[1]) / res)), int(np.ceil((hi[0] - lo[0]) / res)))
    yield "cfg1 (813 bundled stations, bio_1)", xy, z["y"], g1
    for name, n, side, seed in (("cfg2 (2 000 synthetic stations)", 2000, 2000, 21), ("cfg3 (5 000 synthetic stations)", 5000, 10000, 31)):
        g = synth.grid(side, side)
        sxy, _, _, uv = synth.stations(g, n, seed)
        step = side // 400
        ge = hip.Geometry(g.xmin, g.ymax, g.xres * step, g.yres * step, 400, 400)
        yield name, sxy, synth.tps_residual(uv, seed), ge


def test_surface_sensitivity_to_the_gcv_stopping_point(hip):
    rows = []
    for name, xy, y, ge in _sets(hip):
        f = hip.Tps(xy, y, gcv_mode="fields")
        c = hip.Tps(xy, y, gcv_mode="converged")
        S = hip.interpolate(ge, f).cpu().numpy()
        Sc = hip.interpolate(ge, c).cpu().numpy()
        scale = np.abs(S).max()
        d_mode = np.abs(S - Sc).max() / scale
        d_lam = 0.0
        for fac in (1 + 1e-3, 1 - 1e-3):
            Sp = hip.interpolate(ge, hip.Tps(xy, y, lambda_=f.lambda_ * fac)).cpu().numpy()
            d_lam = max(d_lam, np.abs(Sp - S).max() / scale)
        d_fix = np.abs(hip.interpolate(ge, hip.Tps(xy, y, lambda_=f.lambda_)).cpu().numpy() - S).max() / scale
        rel_lam = abs(c.lambda_ - f.lambda_) / f.lambda_
        rows.append({"set": name, "n": int(len(y)), "lambda_fields": f.lambda_, "lambda_converged": c.lambda_, "rel_dlambda": rel_lam,
                     "d_mode": d_mode, "d_lam_1e-3": d_lam, "d_cholesky_route_same_lambda": d_fix})
        assert d_lam < 1e-3, rows[-1]                                  # mild ...
        assert d_fix < 1e-9, rows[-1]                                  # the two solve routes agree at one lambda
        assert d_mode <= 3.0 * (rel_lam / 1e-3) * d_lam + 1e-10, rows[-1]   # ... and linear: the modes differ by what their lambdas explain
    out = os.path.join(ROOT, "gpurun_out", "r3")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "lambda_sensitivity.json"), "w") as fh:
        json.dump(rows, fh, indent=1)
    for r in rows:
        print("%-38s rel dlambda %.2e   d_mode %.2e   d_lam(1e-3) %.2e" % (r["set"], r["rel_dlambda"], r["d_mode"], r["d_lam_1e-3"]))


def test_tiled_surface_sensitivity_cfg1_2x3(hip):
    """The same question for the reference's own Step 3 at the bundled rasters' full geometry (2 476 x 3 264: 2 x 3 tiles,
    V73:656-661): every tile searches its own lambda, the mosaic and the feathering mix them."""
    z = np.load(os.path.join(HERE, "golden", "tps_sampling813.npz"))
    e = np.load(os.path.join(HERE, "golden", "cfg1_extdata.npz"))
    xmin, ymax, xres, yres, nrow, ncol = e["geom"]
    g = hip.Geometry(float(xmin), float(ymax), float(xres) / 2, float(yres) / 2, int(nrow) * 2, int(ncol) * 2)
    xy, y = z["xy"], z["y"]
    inside = (xy[:, 0] > g.xmin) & (xy[:, 0] < g.xmin + g.ncol * g.xres) & (xy[:, 1] < g.ymax) & (xy[:, 1] > g.ymax - g.nrow * g.yres)
    xy, y = xy[inside], y[inside]
    S = hip.tps_residual_surface(g, xy, y, tile_edge=1500, gcv_mode="fields").cpu().numpy()
    Sc = hip.tps_residual_surface(g, xy, y, tile_edge=1500, gcv_mode="converged").cpu().numpy()
    d = float(np.abs(S - Sc).max() / np.abs(S).max())
    out = os.path.join(ROOT, "gpurun_out", "r3")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "lambda_sensitivity_tiled.json"), "w") as fh:
        json.dump({"set": "cfg1 2x3 tiles, %d stations inside the rasters" % len(y), "d_mode": d}, fh)
    print("cfg1 2 x 3 tiles: d_mode %.2e" % d)
    assert d < 1e-3
