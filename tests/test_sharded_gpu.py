"""GPU: the N > 1 path of ShardedMltps with the HIP ops.  A one-GPU box cannot host two RCCL ranks, so the two ranks
share GPU 0 and talk over gloo: what is checked is the sharded arithmetic and plumbing on the device (weighted and
padded row bands, the asynchronous all-gather of the ensemble bands behind rank 0's fit, the coefficient broadcast,
the spline evaluated on the whole grid by every rank, Step 5), not the transport.  The two-rank grid must equal the
one-rank grid bit for bit."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NROW, NCOL, N = 333, 420, 700


def _build(hip):
    from machisplin_amd import sharded, synth
    g = synth.grid(NROW, NCOL)
    planes, nodata = synth.covariates(g, 3, 11, dtype="f32")
    stack = hip.RasterStack(g, planes, nodata)
    xy, rows, cols, uv = synth.stations(g, N, 11)
    import torch
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
    X = np.column_stack([cov, xy])
    resp = synth.response(X, uv, 11)
    params = synth.ensemble_params(X, resp, 11, n_gbm_trees=150, n_rf_trees=8)
    models = [hip.models.from_param_dict(p) for p in params]
    _, weights, wt_total = hip.models.select_weights(synth.OPTX_WEIGHTS)
    return sharded.HipOps(stack, xy, resp, models, weights, wt_total)


def _worker(rank, world, port, share, q):
    import torch
    import torch.distributed as dist
    import machisplin_amd as hip
    from machisplin_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    hip.init(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run = sharded.ShardedMltps(_build(hip), dist, rank, world, NROW, NCOL, rank0_share=share)
        out = run.step()
        torch.cuda.synchronize()
        q.put((rank, out["final"].cpu().numpy(), out["rsq_model"], out["rsq_final"], out["lambda"]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("share", [None, 0.15, 0.0])
def test_two_ranks_on_one_gpu_equal_one_rank_bit_for_bit(hip, share):
    import torch
    import torch.multiprocessing as mp
    from machisplin_amd import sharded
    single = sharded.ShardedMltps(_build(hip), None, 0, 1, NROW, NCOL).step()
    torch.cuda.synchronize()
    want = single["final"].cpu().numpy()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, share, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert single["rsq_final"] > single["rsq_model"]
    for rank, final, rsq_m, rsq_f, lam in results:
        assert final.shape == (NROW, NCOL)
        assert np.array_equal(final, want, equal_nan=True), rank      # every rank holds the whole grid
        assert lam == single["lambda"] and rsq_m == single["rsq_model"] and rsq_f == single["rsq_final"]


# ---------------------------------------------------------------- machisplin.tiles.* sharding (BASELINE config 4) --
T_NROW, T_NCOL, T_N, T_LAYERS = 300, 380, 900, 3


def _build_tiles(hip, rank=0, world=1):
    import torch
    from machisplin_amd import sharded, synth
    g = synth.grid(T_NROW, T_NCOL)
    xy, rows, cols, uv = synth.stations(g, T_N, 21)
    tiles = hip.tiles.tiles_create(g, xy, out_ncol=2, out_nrow=2, feather_d=24)
    cov = synth.covariates_at(g, 3, 21, rows, cols)
    X = np.column_stack([cov, xy])
    base = synth.response(X, uv, 21)
    resp = np.column_stack([base + l + np.sin((2 + l) * uv[:, 0]) for l in range(T_LAYERS)])
    iv = np.column_stack([xy, resp])
    iv[17, 3] = np.nan                                    # an NA in layer 1 drops the station from every layer of its tiles
    _, wts, tot = hip.models.select_weights([0.22, 0.12, 0.18, 0.41], labels="gnmv")
    fitted = {}
    for t in range(4):
        sel = tiles["dat"][t]
        ok = ~np.isnan(iv[sel]).any(axis=1)
        fitted[t] = {}
        for l in range(T_LAYERS):
            if sharded.unit_owner(t, l, 4, world)[0] != rank:
                continue
            params = synth.ensemble_params(X[sel][ok], resp[sel, l][ok], 50 + 7 * l + t, which="gnmv")
            fitted[t][l] = {"models": [hip.models.from_param_dict(p) for p in params], "weights": wts, "wt_total": tot}

    def stack_for_tile(t):
        r0, r1, c0, c1 = (int(v) for v in tiles["win"][t])
        planes, nodata = synth.covariates(g, 3, 21, dtype="f32", window=(r0, r1, c0, c1))
        return hip.RasterStack(tiles["geom"][t], planes, nodata)

    ops = sharded.HipTileOps(g, tiles, stack_for_tile, iv, fitted, tile_edge=100, lambda_=None)
    return g, tiles, iv, fitted, ops


def _tile_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import machisplin_amd as hip
    from machisplin_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    hip.init(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, tiles, iv, fitted, ops = _build_tiles(hip, rank, world)
        out = sharded.TileShardedMltps(ops, dist, rank, world).step()
        torch.cuda.synchronize()
        q.put((rank, {l: v.cpu().numpy() for l, v in out["layers"].items()}, out["rsq_model"], out["rsq_final"]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_tile_sharded_two_ranks_equal_the_single_process_chain(hip):
    """TileShardedMltps with the HIP ops: (tile, layer) units over two ranks (sharing GPU 0, gloo) == one rank ==
    the chain written out by hand (tiles.create -> mltps per tile and layer -> tiles.merge), bit for bit; the
    synthetic covariate crops equal slices of the whole-grid planes."""
    import torch
    import torch.multiprocessing as mp
    from machisplin_amd import sharded, synth
    g, tiles, iv, fitted, ops = _build_tiles(hip)
    single = sharded.TileShardedMltps(ops, None, 0, 1).step()
    torch.cuda.synchronize()
    full, nodata = synth.covariates(g, 3, 21, dtype="f32")
    keep_all = None
    for l in range(T_LAYERS):
        planes = []
        for t in range(4):
            r0, r1, c0, c1 = (int(v) for v in tiles["win"][t])
            sub = hip.RasterStack(tiles["geom"][t], full[:, r0:r1, c0:c1].contiguous(), nodata)
            assert torch.equal(sub.planes, ops._stack(t).planes)
            sel = tiles["dat"][t]
            keep = hip.mltps.complete_cases(sub, iv[sel])
            f = fitted[t][l]
            res = hip.mltps_predict(sub, iv[sel, :2], iv[sel, 2 + l], f["models"], f["weights"], f["wt_total"], tile_edge=100, keep=keep)
            planes.append(res["final"].contiguous())
            assert res["rsq_model"] == single["rsq_model"][l, t]
        want = hip.tiles.tiles_merge(g, tiles["win"], planes, in_ncol=2, in_nrow=2)
        assert torch.equal(torch.nan_to_num(want), torch.nan_to_num(single["layers"][l]))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=800) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    seen = set()
    for rank, layers, rsq_m, rsq_f in results:
        assert np.array_equal(rsq_m, single["rsq_model"]) and np.array_equal(rsq_f, single["rsq_final"], equal_nan=True)
        for l, plane in layers.items():
            assert l % 2 == rank
            assert np.array_equal(plane, single["layers"][l].cpu().numpy(), equal_nan=True)
            seen.add(l)
    assert seen == set(range(T_LAYERS))


# ------------------------------------------------------------------------------- bench.py's N > 1 path, end to end --
def _run_bench(world, workload, port, extra=()):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MHS_BENCH_BACKEND="gloo", MHS_BENCH_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--workload", workload,
           "--steps", "2", "--warmup", "1", *extra]
    pr = subprocess.run(cmd, capture_output=True, text=True, timeout=850, env=env, cwd=root)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_n_ranks_row_bands_plumbing(hip, world):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, one process per rank), with the ranks
    sharing GPU 0 over gloo: calibration pass, weighted row bands (rank 0 carries the fit), coefficient broadcast,
    the in-place all-gather, the per-rank phase table and the predicted-vs-observed step."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = _run_bench(world, "cfg3-mini", port)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["value"] > 0
    mc = d["model_check"]
    assert len(mc["band_ms_per_rank"]) == world and sum(mc["rows_per_rank"]) == 1500
    assert mc["rows_per_rank"][0] <= max(mc["rows_per_rank"][1:])          # rank 0 also fits the spline
    assert mc["predicted_step_ms"] > 0 and mc["observed_step_ms"] > 0
    assert d["rsq_final"] > d["rsq_model"] > 0.5


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_bench_cfg4_n_ranks_plumbing(hip, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = _run_bench(world, "cfg4-mini", port)
    assert d["n_gpus"] == world and d["units_on_rank0"] == 12 // world and d["value"] > 0      # 4 tiles x 3 layers
    assert 0.5 < d["rsq_model_mean"] <= d["rsq_final_mean"]
    assert d["roofline"] is not None and d["unit_profile"]["whole_unit_ms_alone"] > 0


# ------------------------------------------------------------- reference-tiled Step 3 dealt over the ranks (HIP ops) --
def _tiled_tps_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import machisplin_amd as hip
    from machisplin_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    hip.init(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run = sharded.TiledTpsShardedMltps(_build(hip), dist, rank, world, NROW, NCOL, tile_edge=120)
        out = run.step()
        torch.cuda.synchronize()
        q.put((rank, out["final"].cpu().numpy(), out["rsq_model"], out["rsq_final"], out["tile_owner"]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_reference_tiled_step3_sharded_equals_the_one_call_surface(hip):
    """TiledTpsShardedMltps with the HIP ops: the Step-3 tiles (3 x 4 at tile edge 120) dealt over two ranks == one
    rank == mltps_predict's one-call tiled surface (mhs_tps_surface), bit for bit."""
    import torch
    import torch.multiprocessing as mp
    from machisplin_amd import sharded
    ops = _build(hip)
    single = sharded.TiledTpsShardedMltps(ops, None, 0, 1, NROW, NCOL, tile_edge=120).step()
    torch.cuda.synchronize()
    assert (ops.tps_tiles(120)["nRx"], ops.tps_tiles(120)["nCx"]) == (3, 4)
    ref = hip.mltps_predict(ops.stack, ops.X[:, -2:], ops.y, ops.models, ops.weights, ops.wt_total, tile_edge=120)
    assert ref["rsq_final"] == single["rsq_final"] and ref["rsq_model"] == single["rsq_model"]
    assert torch.equal(torch.nan_to_num(ref["final"]), torch.nan_to_num(single["final"]))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tiled_tps_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=500) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = single["final"].cpu().numpy()
    for rank, final, rsq_m, rsq_f, owner in results:
        assert np.array_equal(final, want, equal_nan=True)
        assert rsq_m == single["rsq_model"] and rsq_f == single["rsq_final"]
        assert sorted(set(owner)) == [0, 1] and len(owner) == 12


@pytest.mark.timeout(900)
def test_bench_two_ranks_reference_tiled_step3(hip):
    """`bench.py --gpus 2 --tps-mode tiled`: the Step-3 tiles dealt over the ranks, no serial fit, one all-gather."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = _run_bench(2, "cfg3-mini", port, extra=("--tps-mode", "tiled"))
    assert d["n_gpus"] == 2 and d["value"] > 0 and "reference-tiled" in d["config"]["tps_mode"]
    assert d["rsq_final"] > d["rsq_model"] > 0.5


def test_fit_beside_the_forest_changes_no_number(hip):
    """mhs_fit_reserve_cus (the forest launched with 32 compute units masked out, the GCV fit confined to them, the
    whole step on a stream of its own): the planes, lambda and both R^2 are those of the plain step, bit for bit."""
    import torch
    from machisplin_amd import models, sharded, synth
    side, n = 2304, 1200                     # 5.3e6 cells: above the 2^22-cell threshold of the masked launch
    geom = synth.grid(side, side)
    seed = synth.BASE_SEED + 9
    planes, nodata = synth.covariates(geom, 3, seed, dtype="f32")
    stack = hip.RasterStack(geom, planes, nodata)
    xy, rows, cols, uv = synth.stations(geom, n, seed)
    cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
    X = np.column_stack([cov, xy])
    resp = synth.response(X, uv, seed)
    params = synth.ensemble_params(X, resp, seed, n_gbm_trees=300, n_rf_trees=40)
    mods = [hip.models.from_param_dict(p) for p in params]
    _, weights, wt_total = hip.models.select_weights(synth.OPTX_WEIGHTS)
    outs = []
    for reserve in (0, 32):
        ops = sharded.HipOps(stack, xy, resp, mods, weights, wt_total)
        run = sharded.ShardedMltps(ops, None, 0, 1, side, side)
        run.fit_reserve_cus = reserve
        o = run.step()
        torch.cuda.synchronize()
        outs.append((o["final"].clone(), o["lambda"], o["rsq_model"], o["rsq_final"]))
        assert models.fit_reserve_cus(0) == 0                      # the step leaves the setting as it found it
    assert torch.equal(torch.nan_to_num(outs[0][0]), torch.nan_to_num(outs[1][0]))
    assert outs[0][1:] == outs[1][1:]
    # round 4: the number of units comes from a measurement, not from constants
    cal = run.calibrate_reservation(candidates=(0, 32), repeats=1)
    assert set(cal) == {0, 32} and run.fit_reserve_cus in (0, 32) and run.reservation_calibration["chosen"] == run.fit_reserve_cus
    assert models.fit_reserve_cus(0) == 0
    with pytest.raises(hip.MhsError):
        models.fit_reserve_cus(12)                                 # not a multiple of 8
    with pytest.raises(hip.MhsError):
        models.fit_reserve_cus(200)                                # more than half of the device
