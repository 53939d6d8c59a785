"""Where one (tile, layer) unit of cfg4 spends its ~20 ms: python tools/cfg4_unit_profile.py  (cProfile of 12 units)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench  # noqa: E402
import machisplin_amd as mhs  # noqa: E402

mhs.init()
cfg = dict(bench.WORKLOADS["cfg4"])
wl = bench.TileWorkload(cfg, mhs, torch, None, 0, 1)
ops = wl.ops
out = torch.empty(ops.tile_shapes[0], dtype=torch.float64, device="cuda")
for l in range(3):
    ops.tile_layer(0, l, out)          # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for l in range(12):
    ops.tile_layer(0, l, out)
torch.cuda.synchronize()
pr.disable()
print(f"12 units: {(time.perf_counter() - t0) * 1e3 / 12:.2f} ms each")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
