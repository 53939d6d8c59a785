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
