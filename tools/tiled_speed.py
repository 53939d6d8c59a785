import time, numpy as np, torch, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as m
from machisplin_amd import synth
m.init()
g = synth.grid(10000, 10000)
xy, rows, cols, uv = synth.stations(g, 5000, 3)
resid = synth.tps_residual(uv, 3)
out = torch.empty((10000, 10000), dtype=torch.float64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.tps_residual_surface(g, xy, resid, tile_edge=1500, out=out)
    torch.cuda.synchronize(); print(f"lanes={os.environ.get('MHS_TILE_LANES','8')} tiled surface: {(time.perf_counter()-t0)*1e3:.1f} ms")
