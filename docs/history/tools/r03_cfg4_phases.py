"""Round 3: where a cfg4 step's time goes OUTSIDE its (tile, layer) units (the round-2 line was 714 ms for 48 units of 14.4 ms;
a round-3 run showed 818 ms with the same units): units, the statistics read-back, the merges, the reduction-cache scope exit."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import machisplin_amd as mhs
import bench
import contextlib
mhs.init(0)
cfg = bench.WORKLOADS["cfg4"]
wl = bench.TileWorkload(cfg, mhs, torch, None, 0, 1)
wl.step(); torch.cuda.synchronize()
run, ops = wl.run, wl.ops
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    scope = ops.reduction_cache
    cm = scope()
    cm.__enter__()
    for t, l in run.my_units():
        _, slot = mhs.sharded.unit_owner(t, l, run.n_tiles, 1) if hasattr(mhs, "sharded") else (0, 0)
        from machisplin_amd import sharded
        _, slot = sharded.unit_owner(t, l, run.n_tiles, 1)
        rsq = ops.tile_layer(t, l, run._plane(run.mine, slot, t))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    cm.__exit__(None, None, None)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    planes = {}
    for l in range(run.n_layers):
        rows = [sharded.unit_owner(t, l, run.n_tiles, 1)[1] for t in range(run.n_tiles)]
        planes[l] = ops.merge(l, [run._plane(run.full, rows[t], t) for t in range(run.n_tiles)])
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"units {1e3*(t1-t0):.1f} ms (sum of unit_ms {sum(ops.unit_ms.values()):.1f}), cache scope exit {1e3*(t2-t1):.1f} ms, 12 merges {1e3*(t3-t2):.1f} ms", flush=True)
t0 = time.perf_counter(); wl.step(); torch.cuda.synchronize(); print(f"whole step {1e3*(time.perf_counter()-t0):.1f} ms")
