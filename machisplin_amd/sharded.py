"""One-process-per-GPU sharding of machisplin.mltps Steps 2-5.

Cells are independent given the fitted parameters, so the grid is cut into contiguous
row bands, one per rank.  The only data-path exchanges are the ones the path really has:
  * rank 0 fits the thin-plate spline on the station residuals and BROADCASTS its
    coefficients (3 n + 8 doubles);
  * ONE ALL-GATHER of the ensemble row bands stitches the grid on every rank (RCCL over xGMI
    on the GPU box; gloo in the CPU tests).  It is started as soon as the band is enqueued and
    runs behind rank 0's fit; every rank then evaluates the spline on the whole grid itself
    (milliseconds with the far-field-interpolated sum) and adds it, so the result is the
    one-GPU result bit for bit and Step 5 never needs a second exchange.
The per-band arithmetic is injected (`ops`): bench.py passes the HIP implementation, the
CPU tests pass a numpy stand-in so the collective plumbing is exercised under gloo.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np

# Compute units the fitting rank keeps free of one long ensemble kernel (the forest) so that the spline can be fitted
# beside the grid kernels (mhs_fit_reserve_cus; 0 = off; the cost to the masked kernel comes in quanta of 32 units).
# Whether that pays, and with how many units, depends on how long the forest and the fit take on THIS box with THIS
# workload -- rounds 2 and 3 re-tuned three hand-set constants twice (32 -> 64 units; "up to 6 000 stations"; "bands of 6e7+
# cells") because the forest got shorter.  Round 4: no constants.  ShardedMltps.calibrate_reservation() times whole steps
# with each candidate and keeps the fastest; without that call nothing is reserved (MHS_FIT_RESERVE_CUS forces a value).
FIT_RESERVE_CANDIDATES = (0, 32, 64, 96)


# Bands are cut at multiples of this many grid rows: the coherent gbm kernel sums a cell's trees in an order that depends on
# the 16 x 16-cell tile the cell sits in (tiles anchored to the grid, csrc/ensemble.hip BAND_ALIGN), so bands of whole tiles
# reproduce the one-GPU planes bit for bit.
BAND_ALIGN = 16


def row_bands(nrow: int, world: int, rank0_share: float | None = None):
    """Contiguous row bands, one per rank.  With rank0_share = None the bands are equal
    (ceil(nrow / world) rows, the last ones may be short or empty).  Otherwise rank 0 -- which
    also carries the spline fit -- gets round(rank0_share * nrow) rows (possibly none) and the
    other ranks take ceil(rest / (world - 1)) rows each (the last ones may be short or empty); every cut is a multiple
    of BAND_ALIGN rows.
    Returns (rows of the largest band, [(r0, r1)] per rank).  Every band but rank 0's and the trailing
    ones is exactly `band` rows high, so with rank 0's rows parked at the END of its chunk the all-gather
    of equal chunks lands the whole grid in place (ShardedMltps: no stitching copy)."""
    if world == 1:
        return nrow, [(0, nrow)]
    even = -(-nrow // world)
    # whole 16-row tiles whenever a rank gets at least one (round-4 advisor finding: the old 4-row rule for toy grids cut
    # between the rows of a tile); grids with fewer rows than 16 per rank are below the coherent kernel's own threshold
    align = BAND_ALIGN if even >= BAND_ALIGN else 1
    up = lambda n: -(-n // align) * align
    if rank0_share is None:
        band = up(even)
        return band, [(min(r * band, nrow), min((r + 1) * band, nrow)) for r in range(world)]
    n0 = min(nrow, int(round(min(max(rank0_share, 0.0), 1.0) * nrow / align)) * align)
    rest, others = nrow - n0, world - 1
    h = up(-(-rest // others))
    if n0 > h:     # rank 0 must not be the tallest band (its rows sit at the end of its chunk)
        n0 = h = up(-(-nrow // world))
        return h, [(min(r * h, nrow), min((r + 1) * h, nrow)) for r in range(world)]
    bands = [(0, n0)] + [(min(n0 + k * h, nrow), min(n0 + (k + 1) * h, nrow)) for k in range(others)]
    return max(h, n0), bands


def balanced_rank0_share(world: int, cells_ms: float, fit_ms: float) -> float:
    """Share of the rows for rank 0 so that  fit + its band  takes as long as the other ranks' bands:
    s0 = 1/N - fit (N-1) / (N cells_ms), clamped to [0, 1/N]  (cells_ms = one GPU's time for ALL the
    cells, fit_ms = the stand-alone fit)."""
    if world <= 1:
        return 1.0
    s0 = 1.0 / world - fit_ms * (world - 1) / (world * max(cells_ms, 1e-9))
    return min(max(s0, 0.0), 1.0 / world)


def tps_msg_len(n_stations: int) -> int:
    """Doubles in the coefficient broadcast for a fit on at most `n_stations` station rows."""
    return 3 * int(n_stations) + 9


def pack_tps(knots, c, d, center, scale, lambda_, n_stations: int | None = None):
    """Flat float64 message for the coefficient broadcast: n, u[n], v[n], c[n], d[3], center[2],
    scale[2], lambda -- zero-padded to tps_msg_len(n_stations).  n is the number of UNIQUE knots of
    the fit: fields::Tps collapses replicated locations (two stations in one cell, V73:127-154 make
    knots cell centres), so n <= the number of station rows and travels in the message."""
    knots = np.asarray(knots, dtype=np.float64)
    n = knots.shape[0]
    msg = np.concatenate([[float(n)], knots[:, 0], knots[:, 1], np.asarray(c, dtype=np.float64),
                          np.asarray(d, dtype=np.float64), np.asarray(center, dtype=np.float64),
                          np.asarray(scale, dtype=np.float64), [float(lambda_)]])
    if n_stations is not None:
        if n > n_stations:
            raise ValueError("fit has more knots than station rows")
        msg = np.concatenate([msg, np.zeros(tps_msg_len(n_stations) - msg.size)])
    return msg


def unpack_tps(buf):
    buf = np.asarray(buf, dtype=np.float64)
    n = int(buf[0])
    b = buf[1:]
    return {"knots": np.column_stack([b[:n], b[n:2 * n]]), "c": b[2 * n:3 * n], "d": b[3 * n:3 * n + 3],
            "center": b[3 * n + 3:3 * n + 5], "scale": b[3 * n + 5:3 * n + 7], "lambda": float(b[3 * n + 7]), "n": n}


class ShardedMltps:
    """Steps 2-5 for one response layer over `world` ranks.

    ops must provide (tensors live on ops.device):
      ensemble_band(r0, r1, out)      pred.elev rows [r0, r1) written into `out`         (Step 2)
      station_residuals() -> (knots n x 2, res.FINAL n, resp n, rows n, cols n)          (Step 2)
      tps_fit(knots, resid) -> packed coefficients (pack_tps)                            (Step 3, rank 0)
      tps_band(packed, r0, r1, out)   final.TPS rows [r0, r1)                            (Step 3)
      add(a, b, out)                  out = a + b, NA if either is NA                    (Step 5)
      gather(plane, rows, cols) -> np.ndarray                                            (Step 5)
    """

    def __init__(self, ops, dist, rank: int, world: int, nrow: int, ncol: int, rank0_share: float | None = None):
        import torch
        self.ops, self.dist, self.rank, self.world = ops, dist, rank, world
        self.nrow, self.ncol = nrow, ncol
        self.band, self.bands = row_bands(nrow, world, rank0_share)
        self.r0, self.r1 = self.bands[rank]
        # Per-rank memory: the gather target (grid + at most one band of padding), the final plane, and -- for
        # N > 1 -- this rank's band as the gather's source.  Rank 0's rows sit at the END of its chunk and every
        # other band is `band` rows high, so the gathered chunks ARE the grid, in place, from row `self.lead` on.
        self.lead = self.band - (self.bands[0][1] - self.bands[0][0]) if world > 1 else 0
        kw = {"dtype": torch.float64, "device": ops.device}
        self.full = torch.zeros((self.band * world, ncol), **kw) if world > 1 else None   # all-gather target
        self.pred = torch.zeros((self.band, ncol), **kw)              # this rank's chunk: pred.elev
        self.tot = torch.zeros((self.band, ncol), **kw)               # ... final.TPS, then pred.elev + final.TPS
        self.torch = torch
        self.fit_reserve_cus = int(os.environ.get("MHS_FIT_RESERVE_CUS", 0))
        self.reservation_calibration = None

    def calibrate_reservation(self, candidates=FIT_RESERVE_CANDIDATES, repeats: int = 2):
        """Times whole steps with 0 / 32 / 64 / 96 compute units kept free for the fit on the fitting rank and keeps the
        fastest setting (COLLECTIVE at N > 1: every rank steps along, rank 0 decides for itself -- only its forest is masked).
        A setup call, outside any timed region; MHS_FIT_RESERVE_CUS overrides it.  Returns {units: step ms}."""
        import time
        torch = self.torch
        if "MHS_FIT_RESERVE_CUS" in os.environ or not hasattr(self.ops, "reserve"):
            return None
        timings = {}
        for cus in candidates:
            self.fit_reserve_cus = cus
            best = float("inf")
            for _ in range(repeats):
                if self.world > 1:
                    self.dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                self.step()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e3)
            timings[cus] = best
        if self.rank == 0:
            self.fit_reserve_cus = min(timings, key=lambda c: (timings[c], c))
        else:
            self.fit_reserve_cus = 0
        self.reservation_calibration = {"step_ms_by_reserved_units": timings, "chosen": self.fit_reserve_cus}
        return timings

    def step(self):
        # The whole step is enqueued on a stream of its own (ops.side_stream, non-blocking), never on the NULL stream:
        # the CU-masked streams of mhs_fit_reserve_cus are blocking streams, and ANY operation that reaches the NULL
        # stream between the forest's launch and the end of the fit (a torch.zeros is enough) becomes a barrier
        # between the two.  The caller's current stream takes over at the end.
        side = getattr(self.ops, "side_stream", None) or contextlib.nullcontext
        with side():
            out = self._step()
        join = getattr(self.ops, "join_side_stream", None)
        if join:
            join()
        return out

    def _step(self):
        ops, torch = self.ops, self.torch
        nb = self.r1 - self.r0
        off = self.lead if self.rank == 0 else 0          # rank 0's rows at the end of its chunk
        # The band kernels are only enqueued here (VALU / LDS bound, ~0.5 s at N = 1).  Step 2 at the stations (a
        # few thousand points, all the fit needs) follows on the library's own high-priority stream: a
        # prioritised small grid gets its slots within a millisecond or two of its launch, so the residuals cost
        # nothing on the critical path, and rank 0 fits the spline while every rank's band is still running
        # ... provided its small dependent kernels find workgroup slots beside grid-filling kernels: the fitting rank
        # launches the forest with fit_reserve_cus compute units masked out and confines the fit to them (the number comes
        # from calibrate_reservation's measurement, or is 0).
        reserve = getattr(ops, "reserve", None) if (self.rank == 0 and self.fit_reserve_cus > 0 and nb > 0) else None
        prev = reserve(self.fit_reserve_cus) if reserve else None
        try:
            if nb > 0:
                ops.ensemble_band(self.r0, self.r1, self.pred[off:off + nb])
            knots, resid, resp, rows, cols = ops.station_residuals()
            n = knots.shape[0]
            msg = torch.zeros(tps_msg_len(n), dtype=torch.float64, device=ops.device)
            if self.rank == 0:
                packed0 = np.ascontiguousarray(ops.tps_fit(knots, resid))
                msg[:packed0.size].copy_(torch.from_numpy(packed0))   # replicates collapsed: fewer knots than rows
        finally:
            if prev is not None:
                reserve(prev)
        if self.world > 1:
            self.dist.broadcast(msg, src=0)
        packed = msg.cpu().numpy()
        # Step 3 + the sum on THIS rank's rows only (round 6; rounds 1-5 evaluated the whole grid on every rank, a term that
        # did not shrink with N): the band is evaluated with the whole grid's plan (ops.tps_band: mhs_tps_predict_rows_dev),
        # so the stitched plane is the one-rank plane bit for bit -- the flow of the library's own driver (csrc/multi.hip).
        rows, cols = np.asarray(rows), np.asarray(cols)
        band_pred, band_tot = self.pred[off:off + nb], self.tot[off:off + nb]
        f_actual = np.zeros(n)
        if nb > 0:
            ops.tps_band(packed, self.r0, self.r1, band_tot)
            ops.add(band_pred, band_tot, band_tot)           # Step 5's sum, in place
            mine = (rows >= self.r0) & (rows < self.r1)
            if mine.any():
                f_actual[mine] = ops.gather(band_tot, rows[mine] - self.r0, cols[mine])
        if self.world > 1:                                   # every station's cell lives on exactly one rank
            t = torch.from_numpy(f_actual).to(ops.device)
            self.dist.all_reduce(t)
            f_actual = t.cpu().numpy()
        f_actual[(rows < 0) | (cols < 0)] = np.nan
        # Step 5 (V73:910-930): R^2 at the stations, select
        tss = float(np.sum((resp - resp.mean()) ** 2))
        rsq_model = 1.0 - float(np.sum(resid ** 2)) / tss
        rsq_final = 1.0 - float(np.sum((resp - f_actual) ** 2)) / tss
        src = self.tot if rsq_final > rsq_model else self.pred          # V73:925-930
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.full, src)            # the ONE collective: the output plane, stitched in place
            final = self.full[self.lead:self.lead + self.nrow]
        else:
            final = src[:self.nrow]
        return {"final": final, "rsq_model": rsq_model, "rsq_final": rsq_final, "lambda": unpack_tps(packed)["lambda"]}


def assign_tiles(costs, world: int):
    """Owner rank of every Step-3 tile: longest-processing-time greedy on the tile costs (stations x cells of the
    keep window) -- heaviest tile first onto the least loaded rank, ties to the lower rank.
    Deterministic, so every rank computes the same table."""
    order = sorted(range(len(costs)), key=lambda h: (-float(costs[h]), h))
    load, owner = [0.0] * world, [0] * len(costs)
    for h in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[h] = r
        load[r] += float(costs[h])
    return owner


class TiledTpsShardedMltps:
    """Steps 2-5 over `world` ranks with Step 3 the way the REFERENCE computes it above 1500 px (V73:636-897):
    ceil(nrow/1500) x ceil(ncol/1500) overlapping tiles, each with its own spline fit, mean mosaic, seam feathering.
    The tiles are independent given res.FINAL, so they are dealt to the ranks (assign_tiles) and there is no serial
    fit: every rank predicts its row band of the ensemble AND fits + evaluates its tiles, ONE all-gather moves both
    (chunk = band rows followed by the rank's tile slots), and every rank mosaics, feathers and sums.

    ops: ensemble_band, station_residuals, add, gather as for ShardedMltps, plus
      tps_tiles(tile_edge) -> dict(nRx, nCx, keep = [(r0, r1, c0, c1)], cost = [stations x cells])   (same on every rank)
      tps_tile(h, knots, resid, out)   tile h's keep-window plane (its own fit; zeros below 10 stations, V73:710-721)
      tps_mosaic(nRx, nCx, keep, tiles, out)   mean mosaic + feathering of the tile planes into the grid plane `out`
    """

    def __init__(self, ops, dist, rank: int, world: int, nrow: int, ncol: int, tile_edge: int = 1500):
        import torch
        self.ops, self.dist, self.rank, self.world = ops, dist, rank, world
        self.nrow, self.ncol = nrow, ncol
        self.band, self.bands = row_bands(nrow, world)
        self.r0, self.r1 = self.bands[rank]
        self.layout = ops.tps_tiles(tile_edge)
        keep = [tuple(int(v) for v in w) for w in self.layout["keep"]]
        self.keep = keep
        self.cells = [(w[1] - w[0]) * (w[3] - w[2]) for w in keep]
        self.owner = assign_tiles(self.layout["cost"], world)
        self.offset, fill = [0] * len(keep), [0] * world          # tile h's offset inside its owner's tile area
        for h in range(len(keep)):
            self.offset[h] = fill[self.owner[h]]
            fill[self.owner[h]] += self.cells[h]
        self.band_len = self.band * ncol
        self.chunk = self.band_len + max(fill + [1])
        kw = {"dtype": torch.float64, "device": ops.device}
        self.full = torch.zeros(world * self.chunk, **kw)
        self.mine = self.full[:self.chunk] if world == 1 else torch.zeros(self.chunk, **kw)
        self.total = torch.zeros((nrow, ncol), **kw)
        self.pred_full = None
        self.torch = torch
        # no reservation here by default: the tiles' evaluations are real work (773 ms per cfg3 step when confined to 32
        # units); off the NULL stream the small fits and evaluations find their slots beside the band (546 ms)
        self.fit_reserve_cus = int(os.environ.get("MHS_TILED_FIT_RESERVE_CUS", 0))

    def _tile_view(self, buf, base, h):
        r0, r1, c0, c1 = self.keep[h]
        o = base + self.band_len + self.offset[h]
        return buf[o:o + self.cells[h]].view(r1 - r0, c1 - c0)

    def step(self):
        # as ShardedMltps.step: the whole step on a stream of its own, never on the NULL stream
        side = getattr(self.ops, "side_stream", None) or contextlib.nullcontext
        with side():
            out = self._step()
        join = getattr(self.ops, "join_side_stream", None)
        if join:
            join()
        return out

    def _step(self):
        ops = self.ops
        nb = self.r1 - self.r0
        # this rank's tiles (small fits + their evaluations) run beside its band on the compute units the forest is
        # launched without (mhs_fit_reserve_cus), when the band is long enough for that to pay
        reserve = getattr(ops, "reserve", None) if (self.fit_reserve_cus > 0 and nb > 0) else None
        prev = reserve(self.fit_reserve_cus) if reserve else None
        try:
            if nb > 0:
                ops.ensemble_band(self.r0, self.r1, self.mine[:nb * self.ncol].view(nb, self.ncol))
            knots, resid, resp, rows, cols = ops.station_residuals()
            mine = [h for h in range(len(self.keep)) if self.owner[h] == self.rank]
            if hasattr(ops, "tps_tiles_batch"):      # the library fits a rank's tiles side by side on its lanes
                ops.tps_tiles_batch(mine, knots, resid, [self._tile_view(self.mine, 0, h) for h in mine])
            else:
                for h in mine:
                    ops.tps_tile(h, knots, resid, self._tile_view(self.mine, 0, h))
        finally:
            if prev is not None:
                reserve(prev)
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.full, self.mine)      # the one exchange: bands + tile planes
        tiles = [self._tile_view(self.full, self.owner[h] * self.chunk, h) for h in range(len(self.keep))]
        ops.tps_mosaic(self.layout["nRx"], self.layout["nCx"], self.keep, tiles, self.total)     # final.TPS
        bands = []
        for r, (a, b) in enumerate(self.bands):
            if b > a:
                pb = self.full[r * self.chunk:r * self.chunk + (b - a) * self.ncol].view(b - a, self.ncol)
                bands.append((a, b, pb))
        f_tps = ops.gather(self.total, rows, cols)          # final.TPS at the stations, before the sum goes in place
        for a, b, pb in bands:
            ops.add(pb, self.total[a:b], self.total[a:b])    # Step 5's sum, band by band, in place
        f_actual = ops.gather(self.total, rows, cols)
        tss = float(np.sum((resp - resp.mean()) ** 2))
        rsq_model = 1.0 - float(np.sum(resid ** 2)) / tss
        rsq_final = 1.0 - float(np.sum((resp - f_actual) ** 2)) / tss
        if rsq_final > rsq_model:
            final = self.total
        else:      # V73:925-930: the ensemble alone (assembled from the gathered bands; the rare branch)
            if self.pred_full is None:
                self.pred_full = self.torch.zeros_like(self.total)
            for a, b, pb in bands:
                self.pred_full[a:b].copy_(pb)
            final = self.pred_full
        return {"final": final, "rsq_model": rsq_model, "rsq_final": rsq_final, "tps_at_stations": f_tps,
                "tile_owner": list(self.owner)}


class HipOps:
    """The per-band arithmetic of ShardedMltps on one MI355X, through the C ABI (global-TPS
    mode: one fit on all stations, V73:748-753).  `timings` collects HIP-event durations (ms)
    of the device phases on the launch stream."""

    def __init__(self, stack, int_xy, resp, models, weights, wt_total, lambda_=None, gcv_mode="fields",
                 timed: bool = False, keep=None):
        import torch
        from . import _lib, mltps
        self.torch, self._lib = torch, _lib
        self.stack, self.models, self.weights, self.wt_total = stack, models, list(weights), float(wt_total)
        self.lambda_, self.gcv_mode = lambda_, gcv_mode
        self.device = stack.planes.device
        X, rows, cols = mltps.station_predictors(stack, int_xy)
        y = np.asarray(resp, dtype=np.float64)
        own = ~np.isnan(X).any(axis=1) & ~np.isnan(y)  # complete.cases (V73:154)
        keep = own if keep is None else (np.asarray(keep, dtype=bool) & own)   # table-wide mask: mltps.complete_cases
        self.X, self.rows, self.cols, self.y = X[keep], rows[keep], cols[keep], y[keep]
        self.n_stations = int(self.X.shape[0])
        self.timed = timed
        self.timings = {"ensemble_ms": [], "tps_eval_ms": [], "tps_fit_ms": [], "residuals_ms": []}
        self.last_fit = None
        self._pending = []

    def _timed(self, key, fn):
        """Bracket `fn`'s launches with HIP events on the stream they are enqueued on (torch's
        current stream); no host synchronisation here -- collect() reads the durations later."""
        if not self.timed:
            return fn()
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self._pending.append((key, e0, e1))
        return r

    def collect(self):
        """After a device synchronisation: move the recorded event pairs into `timings` (ms)."""
        for key, e0, e1 in self._pending:
            self.timings.setdefault(key, []).append(e0.elapsed_time(e1))
        self._pending = []

    def reserve(self, n_cus):
        """mhs_fit_reserve_cus; returns the previous setting, or None where the device refuses CU masks (the step then
        runs without the reservation)."""
        from . import _lib
        from .models import fit_reserve_cus
        if getattr(self, "_no_cu_mask", False):
            return None
        try:
            return fit_reserve_cus(n_cus)
        except _lib.MhsError:
            self._no_cu_mask = True
            return None

    @contextlib.contextmanager
    def side_stream(self):
        """The band kernels' own (non-blocking) stream, ordered after what the current stream holds."""
        torch = self.torch
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.device)
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._side):
            yield

    def join_side_stream(self):
        if getattr(self, "_side", None) is not None:
            self.torch.cuda.current_stream(self.device).wait_stream(self._side)

    def ensemble_band(self, r0, r1, out):
        from .models import ensemble_predict
        g = self.stack.geom
        self._timed("ensemble_ms", lambda: ensemble_predict(self.stack, self.models, self.weights, self.wt_total,
                                                            window=(r0, r1, 0, g.ncol), out=out))

    def station_residuals(self):
        import time
        from .mltps import ensemble_residuals
        t0 = time.perf_counter()
        res = ensemble_residuals(self.models, self.weights, self.wt_total, self.X, self.y)
        self.timings["residuals_ms"].append((time.perf_counter() - t0) * 1e3)
        return self.X[:, -2:], res, self.y, self.rows, self.cols

    def tps_fit(self, knots, resid):
        import time
        from .tps import Tps
        t0 = time.perf_counter()
        fit = Tps(knots, resid, lambda_=self.lambda_, gcv_mode=self.gcv_mode)
        self.timings["tps_fit_ms"].append((time.perf_counter() - t0) * 1e3)
        self.last_fit = fit
        return pack_tps(fit.knots, fit.c, fit.d, fit.center, fit.scale, fit.lambda_)

    def tps_band(self, packed, r0, r1, out):
        from .tps import Tps, interpolate
        d = unpack_tps(packed)
        fit = Tps.from_coef(d["knots"], d["c"], d["d"], d["lambda"], d["center"], d["scale"])
        g = self.stack.geom
        # rows [r0, r1) with the WHOLE grid's plan: the bands of several ranks stitch to the one-rank plane bit for bit
        self._timed("tps_eval_ms", lambda: interpolate(g, fit, rows=(r0, r1), out=out))
        self.last_eval_plan = fit.eval_plan()

    # ---- reference-tiled Step 3 (TiledTpsShardedMltps) --------------------------------------------------------------
    def tps_tiles(self, tile_edge):
        from . import tiles
        g = self.stack.geom
        nRx, nCx, fit, keep = tiles.step3_tile_windows(g, tile_edge)
        ok = ~np.isnan(self.X[:, 0])                       # stations on an NA cell of covariate 1 are dropped (V73:701-706)
        self._tile_sel, cost = [], []
        for h in range(nRx * nCx):
            fr0, fr1, fc0, fc1 = (int(v) for v in fit[h])
            sel = np.flatnonzero(ok & (self.rows >= fr0) & (self.rows < fr1) & (self.cols >= fc0) & (self.cols < fc1))
            self._tile_sel.append(sel)
            kr0, kr1, kc0, kc1 = (int(v) for v in keep[h])
            cost.append(float(sel.size) * (kr1 - kr0) * (kc1 - kc0) + float(sel.size) ** 3)
        self._tile_fit, self._tile_keep, self._tile_edge = fit, keep, tile_edge
        return {"nRx": int(nRx), "nCx": int(nCx), "keep": [tuple(int(v) for v in w) for w in keep], "cost": cost}

    def tps_tile(self, h, knots, resid, out):
        from .tps import Tps, interpolate
        sel = self._tile_sel[h]
        if sel.size < 10:                                  # V73:710-721: the tile is all zeros
            out.zero_()
            return
        fr0, fr1, fc0, fc1 = (int(v) for v in self._tile_fit[h])
        kr0, kr1, kc0, kc1 = (int(v) for v in self._tile_keep[h])
        fit = Tps(knots[sel], resid[sel], lambda_=self.lambda_, gcv_mode=self.gcv_mode)
        gf = self.stack.geom.window(fr0, fr1, fc0, fc1)     # terra::rast(rb): geometry of the fit raster
        interpolate(gf, fit, window=(kr0 - fr0, kr1 - fr0, kc0 - fc0, kc1 - fc0), out=out)

    def tps_tiles_batch(self, hs, knots, resid, outs):
        """This rank's tiles in ONE library call (mhs_tps_tiles_dev: up to 8 fits side by side)."""
        import ctypes as C
        import time
        if not hs:
            return
        t0 = time.perf_counter()
        g = self.stack.geom.c_struct()
        xyf = np.asfortranarray(np.asarray(knots, dtype=np.float64))
        res = np.ascontiguousarray(resid, dtype=np.float64)
        cov = np.ascontiguousarray(self.X[:, 0], dtype=np.float64)
        ids = np.ascontiguousarray(hs, dtype=np.int64)
        ptrs = (C.c_void_p * len(hs))(*[o.data_ptr() for o in outs])
        mode = {"fields": self._lib.GCV_FIELDS, "converged": self._lib.GCV_CONVERGED}[self.gcv_mode]
        self._lib.check(self._lib.lib().mhs_tps_tiles_dev(C.byref(g), xyf.ctypes.data, res.ctypes.data, res.shape[0], cov.ctypes.data,
                                                          int(self._tile_edge), float("nan") if self.lambda_ is None else float(self.lambda_),
                                                          mode, ids.ctypes.data, len(hs), ptrs))
        self.timings.setdefault("tps_tiles_ms", []).append((time.perf_counter() - t0) * 1e3)

    def tps_mosaic(self, nRx, nCx, keep, tiles_, out):
        from . import tiles
        tiles.mosaic_feather(self.stack.geom, nRx, nCx, np.asarray(keep, dtype=np.int64), list(tiles_), merge_mode=False, out=out)

    def add(self, a, b, out):
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        assert a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
        self._lib.check(self._lib.lib().mhs_scale_add_dev(a.data_ptr(), 1.0, b.data_ptr(), out.data_ptr(), a.numel(), st))

    def gather(self, plane, rows, cols):
        from . import tiles
        return tiles.extract(plane, rows, cols)


# ======================================================================================================
# machisplin.tiles.* sharding (BASELINE.json configs[3]; README.md:157-215): machisplin.tiles.create cuts the
# study area into out.nrow x out.ncol user tiles (V73:1165-1256), every tile is an INDEPENDENT machisplin.mltps
# run per response layer (its own stations, its own Step-3 tiles and spline fits, its own R^2 selection), and
# machisplin.tiles.merge feathers the per-tile finals back together (V73:1392-1548).  The shard unit is one
# (tile, layer) run -- no halo, no exchange while it runs.  ONE all-gather moves every unit's final plane (and its
# two R^2 values) to every rank; the merge of layer l then runs on rank l mod N (or on every rank).
# ======================================================================================================
def unit_owner(tile: int, layer: int, n_tiles: int, world: int):
    """(rank, slot) of the (tile, layer) unit: units are numbered layer-major, u = layer * n_tiles + tile, and
    dealt round-robin -- u mod N is the rank, u div N its slot in that rank's chunk of the gather.  With as many
    ranks as tiles every rank keeps ONE tile's covariates resident (tile t -> GPU t mod N, all layers); with more
    ranks than tiles the layers of a tile are shared out as well."""
    u = layer * n_tiles + tile
    return u % world, u // world


class TileShardedMltps:
    """Every response layer of a tiled run over `world` ranks.

    ops must provide (tensors live on ops.device):
      tile_shapes -> [(rows, cols)] of the user tiles, in machisplin.tiles.create order (row-major from the SW)
      n_layers
      tile_layer(tile, layer, out) -> (rsq_model, rsq_final)   the whole mltps run of one tile and layer, final
                                                               plane written into `out` (rows x cols, contiguous)
      merge(layer, planes) -> tensor                           machisplin.tiles.merge of one layer's tile planes
    `merge_on`: "owner" = layer l is merged on rank l mod N only (the rank that would write its GeoTIFF, V73:998);
    "all" = every rank merges every layer.
    """

    def __init__(self, ops, dist, rank: int, world: int, merge_on: str = "owner"):
        import torch
        if merge_on not in ("owner", "all"):
            raise ValueError("merge_on must be 'owner' or 'all'")
        self.ops, self.dist, self.rank, self.world, self.merge_on = ops, dist, rank, world, merge_on
        self.shapes = [(int(h), int(w)) for h, w in ops.tile_shapes]
        self.n_tiles, self.n_layers = len(self.shapes), int(ops.n_layers)
        n_units = self.n_tiles * self.n_layers
        self.slots = -(-n_units // world)
        self.slot_len = max(h * w for h, w in self.shapes) + 2      # the plane + (rsq_model, rsq_final)
        kw = {"dtype": torch.float64, "device": ops.device}
        self.full = torch.zeros((world * self.slots, self.slot_len), **kw)     # gather target: every unit
        self.mine = self.full[rank * self.slots:(rank + 1) * self.slots] if world == 1 else \
            torch.zeros((self.slots, self.slot_len), **kw)
        self.torch = torch

    def my_units(self):
        return [(t, l) for l in range(self.n_layers) for t in range(self.n_tiles)
                if unit_owner(t, l, self.n_tiles, self.world)[0] == self.rank]

    def _plane(self, buf, slot, tile):
        h, w = self.shapes[tile]
        return buf[slot, :h * w].view(h, w)

    def step(self):
        torch = self.torch
        # one scope per step: a tile's response layers share its stations, so the Step-3 tiles' reductions are built by
        # the first layer this rank runs on the tile and reused by its other layers (ops.reduction_cache; bit-identical
        # fits) -- and nothing is carried from one step to the next
        scope = getattr(self.ops, "reduction_cache", None) or contextlib.nullcontext
        with scope():
            for t, l in self.my_units():
                _, slot = unit_owner(t, l, self.n_tiles, self.world)
                rsq = self.ops.tile_layer(t, l, self._plane(self.mine, slot, t))
                self.mine[slot, -2:] = torch.tensor([float(rsq[0]), float(rsq[1])], dtype=torch.float64).to(self.mine.device)
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.full, self.mine)      # the one exchange of the path
        stats = self.full[:, -2:].cpu().numpy()
        out = {"layers": {}, "rsq_model": np.full((self.n_layers, self.n_tiles), np.nan),
               "rsq_final": np.full((self.n_layers, self.n_tiles), np.nan)}
        for l in range(self.n_layers):
            rows = []
            for t in range(self.n_tiles):
                r, slot = unit_owner(t, l, self.n_tiles, self.world)
                rows.append(r * self.slots + slot)
                out["rsq_model"][l, t], out["rsq_final"][l, t] = stats[rows[-1]]
            if self.merge_on == "all" or l % self.world == self.rank:
                out["layers"][l] = self.ops.merge(l, [self._plane(self.full, rows[t], t) for t in range(self.n_tiles)])
        return out


class HipTileOps:
    """TileShardedMltps' per-unit arithmetic on one MI355X through the C ABI: `tiles` is
    machisplin_amd.tiles.tiles_create's result for the study area, `stack_for_tile(t)` returns the RasterStack of
    tile t's window (cropped covariates -- each rank only ever asks for the tiles it owns), `int_values` the
    station table (long, lat, one response column per layer) and `fitted[t][l]` the members / weights / wt_total
    of tile t, layer l (model fitting is out of scope: V73:176-436 stays in R)."""

    def __init__(self, geom, tiles, stack_for_tile, int_values, fitted, tile_edge=1500, lambda_=None, gcv_mode="fields",
                 tps: bool = True):
        import torch
        from . import _lib
        self.torch = torch
        self.device = torch.device("cuda", _lib.init())
        self.geom, self.tiles, self.stack_for_tile = geom, tiles, stack_for_tile
        self.int_values = np.asarray(int_values, dtype=np.float64)
        self.fitted, self.tile_edge, self.lambda_, self.gcv_mode, self.tps = fitted, tile_edge, lambda_, gcv_mode, tps
        self.tile_shapes = [(int(w[1] - w[0]), int(w[3] - w[2])) for w in tiles["win"]]
        self.n_layers = self.int_values.shape[1] - 2
        self._stacks, self._keep = {}, {}
        self.unit_ms = {}

    def _stack(self, t):
        if t not in self._stacks:
            from .mltps import complete_cases
            self._stacks[t] = self.stack_for_tile(t)
            sel = self.tiles["dat"][t]
            self._keep[t] = complete_cases(self._stacks[t], self.int_values[sel])   # once per tile, over every column
        return self._stacks[t]

    def tile_layer(self, t, l, out):
        import time
        from .mltps import mltps_predict
        stack = self._stack(t)
        sel = self.tiles["dat"][t]
        f = self.fitted[t][l]
        t0 = time.perf_counter()
        res = mltps_predict(stack, self.int_values[sel, :2], self.int_values[sel, 2 + l], f["models"], f["weights"],
                            f["wt_total"], tps=self.tps, tile_edge=self.tile_edge, lambda_=self.lambda_,
                            gcv_mode=self.gcv_mode, keep=self._keep[t])
        out.copy_(res["final"])
        self.unit_ms[(t, l)] = (time.perf_counter() - t0) * 1e3
        return res["rsq_model"], res.get("rsq_final", float("nan"))

    def reduction_cache(self):
        from .tps import reduction_cache
        return reduction_cache()

    def merge(self, l, planes):
        from . import tiles as tl
        return tl.tiles_merge(self.geom, self.tiles["win"], [p.contiguous() for p in planes], in_ncol=self.tiles["nC"],
                              in_nrow=self.tiles["nR"])
