"""Round 5: the forest's ring kernel (block-level subtree staging) against round 4's loader-wave kernel (MHS_RF_KERNEL=ld), the
double-buffered kernel and the generic walk on cfg3's forest: times per 1e8 cells and bit-for-bit equality of the planes, on the
SURVEY 8d planes and on the same planes with white noise.   python tools/r05_forest_ring.py [side]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as m
from machisplin_amd import synth
m.init()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
xy, rows, cols, uv = synth.stations(g, nst, seed)
cov_at = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov_at, xy])
y = synth.response(X, uv, seed)
prm = synth.rf_params(X, y, seed)
print("nodes per tree up to", int(np.diff(prm["tree_offsets"]).max()), flush=True)
mod = m.models.from_param_dict(prm)
gen = torch.Generator(device="cuda"); gen.manual_seed(7)
variants = [("8d planes", planes)]
for frac in (0.01, 0.1):
    noisy = planes.clone()
    for k in range(3):
        lo, hi = synth.COV_RANGES[k]
        noisy[k] += (torch.rand((side, side), device="cuda", generator=gen) - 0.5) * (frac * (hi - lo))
    variants.append(("8d + %g %% noise" % (100 * frac), noisy))
out = torch.empty((side, side), dtype=torch.float64, device="cuda")
def timed(stack, env):
    for k, v in env.items(): os.environ[k] = v
    try:
        m.predict(stack, mod, out=out); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); m.predict(stack, mod, out=out); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return best * 1e3 * 1e8 / (side * side), out.clone()
    finally:
        for k in env: del os.environ[k]
for name, pl in variants:
    stack = m.RasterStack(g, pl, nodata)
    t_ring, p_ring = timed(stack, {})
    line = f"{name:18s} default {t_ring:7.1f} ms/1e8"
    for label, env in (("cbs", {"MHS_RF_KERNEL": "cbs"}), ("sub", {"MHS_RF_KERNEL": "sub"}), ("db", {"MHS_RF_KERNEL": "db"}), ("compact", {"MHS_RF_KERNEL": "compact"})):
        t, pln = timed(stack, env)
        line += f" | {label} {t:7.1f} equal={bool(torch.equal(torch.nan_to_num(pln), torch.nan_to_num(p_ring)))}"
    print(line, flush=True)
