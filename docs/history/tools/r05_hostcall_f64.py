"""Round 5: mhs_mltps_grid_multi (host planes in, host plane out: what the R shim calls per response layer) with FLOAT64 host
planes -- what terra holds in RAM -- on cfg3, both Step-3 modes, against the resident step (MultiStack.step).
    python tools/r05_hostcall_f64.py [side=10000]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import synth, multi

side = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
multi.init_devices(1, [0])
g = synth.grid(side, side)
seed = synth.BASE_SEED + 3
planes, nodata = synth.covariates(g, 3, seed, dtype="f32")
xy, rows, cols, uv = synth.stations(g, 5000, seed)
cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
X = np.column_stack([cov, xy])
y = synth.response(X, uv, seed)
models = [m.models.from_param_dict(p) for p in synth.ensemble_params(X, y, seed)]
_, wts, tot = m.models.select_weights(synth.OPTX_WEIGHTS)
host32 = planes.cpu().numpy()
host64 = host32.astype(np.float64)
del planes
torch.cuda.empty_cache()
out = np.zeros((side, side))
for name, host in (("float32", host32), ("float64", host64)):
    for tile_edge in (None, 1500):
        ms = multi.MultiStack(g, host, nodata)
        for _ in range(2):
            info = ms.step(models, wts, tot, X, y, tile_edge=tile_edge)
        t0 = time.perf_counter(); info = ms.step(models, wts, tot, X, y, tile_edge=tile_edge); resident = (time.perf_counter() - t0) * 1e3
        want = ms.download()
        ms.free()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            _, hinfo = multi.mltps_grid_multi(g, host, nodata, models, wts, tot, X, y, tile_edge=tile_edge, out=out)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print(f"{name} planes, Step 3 {'global' if tile_edge is None else 'reference-tiled'}: resident step {resident:6.1f} ms   host call {best:6.1f} ms "
              f"(+{best - resident:5.1f}, {100 * (best - resident) / resident:4.1f} %)   copies up issued in {hinfo['upload_ms']:5.1f} ms, "
              f"left of the copies down {hinfo['download_ms']:4.1f} ms   equal: {bool(np.array_equal(out, want, equal_nan=True))}", flush=True)
