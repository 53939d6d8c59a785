"""CPU, world_size 2 / 3 / 4 over gloo: the machisplin.tiles.* sharding of TileShardedMltps (BASELINE config 4) --
(tile, layer) units dealt round-robin over the ranks, ONE all-gather of the units' final planes (+ their two R^2
values), machisplin.tiles.merge of layer l on rank l mod N -- with the per-unit arithmetic supplied by the numpy
oracle (tiles.create windows, a linear member + single-fit TPS + Step-5 selection per tile, tiles.merge).  The
merged layers must equal the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from machisplin_amd import sharded
from oracle import ensemble as oe
from oracle import tiles as ot
from oracle import tps as otps

NROW, NCOL, N, LAYERS = 64, 90, 260, 3


class OracleTileOps:
    device = torch.device("cpu")

    def __init__(self):
        rng = np.random.default_rng(4)
        self.g = ot.Geom(-78.0, -5.0, 0.01, 0.01, NROW, NCOL)
        x, y = otps.cell_centres(-78.0, -5.0, 0.01, 0.01, NROW, NCOL)
        self.cov = rng.uniform(0, 100, (2, NROW, NCOL))
        cells = rng.choice(NROW * NCOL, N, replace=False)
        self.rows, self.cols = np.divmod(cells, NCOL)
        self.xy = np.column_stack([x[self.cols], y[self.rows]])
        base = 3 + 0.05 * self.cov[0, self.rows, self.cols] + np.sin(40 * self.xy[:, 0])
        self.resp = np.column_stack([base + 0.1 * rng.standard_normal(N) + k for k in range(LAYERS)])
        self.boxes, self.wins, self.sel = ot.tiles_create(self.g, self.xy, 2, 2, 10)
        self.tile_shapes = [(w[1] - w[0], w[3] - w[2]) for w in self.wins]
        self.n_layers = LAYERS
        self.calls = []

    def tile_layer(self, t, l, out):
        self.calls.append((t, l))
        r0, r1, c0, c1 = self.wins[t]
        tg = ot.window_geom(self.g, self.wins[t])
        x, y = otps.cell_centres(tg.xmin, tg.ymax, tg.xres, tg.yres, tg.nrow, tg.ncol)
        Xt = oe.stack_predictors(self.cov[:, r0:r1, c0:c1], (x, y))
        sel = self.sel[t]
        rows = np.array([tg.row_from_y(v) for v in self.xy[sel, 1]])
        cols = np.array([tg.col_from_x(v) for v in self.xy[sel, 0]])
        Xs, ys = Xt[rows * tg.ncol + cols], self.resp[sel, l]
        A = np.column_stack([np.ones(sel.size), Xs])
        model = oe.lm_model(np.linalg.lstsq(A, ys, rcond=None)[0])
        pred = oe.predict(model, Xt).reshape(tg.nrow, tg.ncol)
        res = ys - oe.predict(model, Xs)
        m = otps.fit(Xs[:, -2:], res, lam=1e-3)
        tps = otps.predict_grid(m, tg.xmin, tg.ymax, tg.xres, tg.yres, tg.nrow, tg.ncol)
        final, rsq_m, rsq_f, _ = ot.step5_combine(tg, pred, tps, Xs[:, -2:], ys)
        out.copy_(torch.from_numpy(final))
        return rsq_m, rsq_f

    def merge(self, l, planes):
        return torch.from_numpy(ot.tiles_merge(self.g, self.wins, [p.numpy() for p in planes], 2, 2))


def _worker(rank, world, port, q, merge_on):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ops = OracleTileOps()
        out = sharded.TileShardedMltps(ops, dist, rank, world, merge_on=merge_on).step()
        q.put((rank, {l: v.numpy().copy() for l, v in out["layers"].items()}, out["rsq_model"], out["rsq_final"], ops.calls))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_unit_owner_deals_units_round_robin():
    # 4 tiles on 4 ranks: tile t lives on rank t for every layer (one tile's covariates per GPU)
    assert all(sharded.unit_owner(t, l, 4, 4) == (t, l) for t in range(4) for l in range(12))
    # 4 tiles on 8 ranks: the layers of a tile are shared out as well, every rank still sees ONE tile
    owners = {}
    for l in range(12):
        for t in range(4):
            r, slot = sharded.unit_owner(t, l, 4, 8)
            owners.setdefault(r, set()).add(t)
            assert slot == (l * 4 + t) // 8
    assert sorted(owners) == list(range(8)) and all(len(v) == 1 for v in owners.values())
    # 48 units over 1 / 2 / 4 / 8 ranks divide evenly
    for world in (1, 2, 4, 8):
        counts = [0] * world
        for l in range(12):
            for t in range(4):
                counts[sharded.unit_owner(t, l, 4, world)[0]] += 1
        assert len(set(counts)) == 1


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,merge_on", [(2, "owner"), (3, "all"), (4, "owner")])
def test_tile_sharded_run_equals_the_single_process_run(world, merge_on):
    single = sharded.TileShardedMltps(OracleTileOps(), None, 0, 1).step()
    assert sorted(single["layers"]) == list(range(LAYERS))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, merge_on)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = set()
    all_calls = []
    for rank, layers, rsq_m, rsq_f, calls in results:
        all_calls += calls
        assert np.array_equal(rsq_m, single["rsq_model"]) and np.array_equal(rsq_f, single["rsq_final"])   # every rank has every unit's R^2
        want_layers = list(range(LAYERS)) if merge_on == "all" else [l for l in range(LAYERS) if l % world == rank]
        assert sorted(layers) == want_layers
        for l, plane in layers.items():
            assert plane.shape == (NROW, NCOL)
            assert np.array_equal(plane, single["layers"][l].numpy(), equal_nan=True)
            seen.add(l)
    assert seen == set(range(LAYERS))
    assert sorted(all_calls) == sorted((t, l) for t in range(4) for l in range(LAYERS))    # every unit ran exactly once
