"""Round 3: phases of Step 3 + 4 (reference-tiled spline surface) of ONE cfg4 user tile (5 025 x 5 025 cells, ~1 260 stations,
4 x 4 Step-3 tiles), first layer and cached layers: MHS_TIMING prints tile fits + evaluation / mosaic + feather."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as mhs
from machisplin_amd import synth
from machisplin_amd.tps import reduction_cache
mhs.init(0)
g = synth.grid(10000, 10000)
seed = synth.BASE_SEED + 4
xy, rows, cols, uv = synth.stations(g, 5000, seed)
tiles = mhs.tiles.tiles_create(g, xy, out_ncol=2, out_nrow=2, feather_d=50)
tg = tiles["geom"][0]
sel = tiles["dat"][0]
res = synth.tps_residual(uv[sel], 3)
cov1 = np.ones(len(sel))
os.environ["MHS_TIMING"] = "1"
with reduction_cache():
    for k in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mhs.tps_residual_surface(tg, xy[sel], res + 0.01 * k, cov1_at_stations=cov1, tile_edge=1500)
        torch.cuda.synchronize(); print(f"call {k}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
