"""Round 5: what the GCV fit of 5 000 stations costs the member it runs beside (no compute units reserved): the member alone on
side x side cells, then the same launch with the fit started right behind it -- the difference is what the step pays for the fit
if it overlaps THAT member.     python tools/r05_fit_beside_member.py [side=10000]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import machisplin_amd as mhs
from machisplin_amd import models, synth

mhs.init()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
geom = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(geom, 3, seed, dtype="f32")
stack = mhs.RasterStack(geom, planes, nodata)
xy, r, c, uv = synth.stations(geom, 5000, seed)
cov = planes[:, torch.from_numpy(r).cuda(), torch.from_numpy(c).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
y = synth.response(X, uv, seed)
prm = synth.ensemble_params(X, y, seed, n_gbm_trees=10000, n_rf_trees=500)
mods = {p["kind"]: models.from_param_dict(p) for p in prm}
res = synth.tps_residual(uv, seed)
out = torch.zeros((side, side), dtype=torch.float64, device="cuda")
mhs.Tps(xy, res)
torch.cuda.synchronize()
t0 = time.perf_counter(); mhs.Tps(xy, res); torch.cuda.synchronize(); fit_alone = 1e3 * (time.perf_counter() - t0)
print(f"fit alone {fit_alone:.1f} ms")
torch.cuda.set_stream(torch.cuda.Stream())
for kind in ("gbm", "rf", "svr"):
    models.members_predict(stack, [mods[kind]], [1.0], out=out)
    torch.cuda.synchronize()
    def run(with_fit):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        models.members_predict(stack, [mods[kind]], [1.0], out=out)
        e1.record()
        if with_fit:
            mhs.Tps(xy, res)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1), 1e3 * (time.perf_counter() - t_start)
    runs = []
    for with_fit in (False, True):
        best = (1e9, 1e9)
        for _ in range(3):
            torch.cuda.synchronize()
            t_start = time.perf_counter()
            got = run(with_fit)
            best = min(best, got, key=lambda v: v[1])
        runs.append(best)
    (alone, wall_alone), (beside, wall_beside) = runs
    print(f"{kind:4s}: alone {alone:7.1f} ms (wall {wall_alone:6.1f})   with the fit behind it: member {beside:7.1f} ms, member AND fit done after {wall_beside:6.1f} ms"
          f"   -> the pair costs {wall_beside - wall_alone:5.1f} ms more than the member ({100 * (wall_beside - wall_alone) / fit_alone:.0f} % of the fit's stand-alone time)", flush=True)
