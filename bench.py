#!/usr/bin/env python
"""bench.py -- MACHISPLIN hot path on MI355X.

One "step" = one pass of machisplin.mltps Steps 2-5 (V73:442-930) over one synthetic input
set already resident in HBM: six-member ensemble prediction of every grid cell, station
residuals, thin-plate-spline fit on the residuals (GCV lambda), TPS evaluation of every
cell, the final sum and the R^2 selection [+ coefficient broadcast and ONE all-gather of
the row bands when N > 1].  Model fitting (Step 1) is out of scope.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (algorithmic work per
launch from SURVEY.md section 8d divided by its HIP-event duration on the launch stream);
`cpu_baseline` is the oracle's C restatement timed on this box's host cores on a bounded
sample of the same workload.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64: vector == MFMA peak (v_mfma_f64_16x16x4 measured 75.3 TF)
HBM_PEAK_GBS = 8000.0
FP32_PEAK_TFLOPS = 157.3  # FP32 vector == FP32 MFMA peak
LDS_PEAK_GBS = 256 * 128 * 2.4  # 256 CUs x 128 B/clk x 2.4 GHz (round-1/2 view of the LDS rate; kept for comparison)
VALU_ISSUE_PEAK_G = 1024 * 2.4 / 4.0   # G wave-instructions/s: 1024 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz
LDS_CYCLE_PEAK_G = 256 * 2.4           # G LDS-array cycles/s: 256 CUs at 2.4 GHz (MI355X guide: ds_read_b64 = ds_read_b32 = 2 cycles)


PMC_FILES = ("r06_8d_members_pmc_derived.json", "r05_8d_members_pmc_derived.json")
MEMBER_KERNEL_SOURCES = ("ensemble.hip", "forest.hip", "ensemble_int.h", "devmath.h", "gbm_rt_loop.inc", "rf_walk_loop4x.inc", "rf_walk_loop4xo.inc",
                         "rf_walk_loop5x.inc", "rf_walk_loop5xo.inc")


def member_kernel_source_hash():
    """sha256 over the sources the member kernels are built from: PMC counters are only quoted for THIS build of them."""
    import hashlib
    h = hashlib.sha256()
    for name in MEMBER_KERNEL_SOURCES:
        with open(os.path.join(ROOT, "machisplin_amd", "csrc", name), "rb") as f:
            h.update(name.encode()); h.update(f.read())
    return h.hexdigest()[:16]


def pmc_derived():
    """profiles/r06_8d_members_pmc_derived.json (tools/r06_collect.sh memberspmc = tools/r04_members_pmc.sh + tools/r04_pmc_derive.py;
    an earlier round's file if its hash still matches): per-unit instruction counts and pipe utilisations of the member kernels from rocprofv3 --pmc passes
    of a torch-free driver over cfg3's own 10 000 x 10 000 SURVEY-8d planes.  Every figure the roofline rows take from it is
    a property of THOSE rasters and of that run, not of the timed steps; the rows say so ("pmc_source")."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if d.get("member_kernel_source_hash") != member_kernel_source_hash():
                continue      # counters of another build of the member kernels: not quoted (round-5 verdict item 6)
            out = {("gbm" if "gbm" in k else "rf" if "rf_" in k else "small" if "small" in k else "svr"): v for k, v in d.items() if isinstance(v, dict)}
            out["_source"] = "profiles/%s (%s planes, %s x %s cells; member kernel sources %s)" % (
                name, d.get("rasters", "8d"), d.get("grid", ["?", "?"])[0], d.get("grid", ["?", "?"])[1], d["member_kernel_source_hash"])
            return out
        except (OSError, ValueError):
            continue
    return {}

def forest_traffic(n_nodes, cells, layers):
    """The forest's traffic past the L2 as the PMC passes measured it (FETCH_SIZE doubled + WRITE_SIZE, per cell, on cfg3's own
    10 000 x 10 000 8d planes) against the algorithmic bytes (the covariate planes once, the plane written once, the forest once),
    for the default kernel and for the subtree-staging kernel (MHS_RF_KERNEL=sub) -- round-4 verdict item 8: a first-class field."""
    alg = 4.0 * layers + 8.0 + 8.0 * n_nodes / cells
    out = {"algorithmic_bytes_per_cell": alg, "what": "float32 covariate planes read once + float64 plane written once + 8-byte node records once"}
    for key, names in (("default_kernel", ("r06_8d_members_pmc_derived.json", "r05_8d_members_pmc_derived.json")),
                       ("subtree_kernel", ("r05_8d_rfsub_members_pmc_derived.json",))):
        try:
            name = next(n for n in names if os.path.exists(os.path.join(ROOT, "profiles", n)))
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            k = next(k for k in d if isinstance(d[k], dict) and "rf_" in k)
            b = d[k]["hbm_bytes_per_cell_fetch_x2_plus_write"]
            out[key] = {"kernel": k, "measured_bytes_per_cell": b, "ratio_to_algorithmic": b / alg, "source": "profiles/" + name}
        except (OSError, ValueError, StopIteration, KeyError):
            out[key] = None
    return out


WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the north-star target is quoted on
    "cfg3": dict(stations=5000, side=10000, layers=3, gbm_trees=10000, rf_trees=500, ensemble=True,
                 name="cfg3: 5000 stations, 3 covariates, 10000x10000 grid, 6-model ensemble + TPS residual correction"),
    # BASELINE.json configs[1]
    "cfg2": dict(stations=2000, side=2000, layers=3, gbm_trees=0, rf_trees=0, ensemble=False,
                 name="cfg2: 2000 stations, 2000x2000 grid, TPS only"),
    # BASELINE.json configs[4] (its 8-GPU shape; also runs on one GPU: 3.2 GB Gram matrix, 8e12 TPS pairs)
    "cfg5": dict(stations=20000, side=20000, layers=5, gbm_trees=10000, rf_trees=500, ensemble=True,
                 name="cfg5: 20000 stations, 5 covariates, 20000x20000 grid, 6-model ensemble + TPS residual correction"),
    # BASELINE.json configs[3]: 12 response layers, smooth members only (g, n, m, v -- V73:366-392), the study area cut
    # into 2 x 2 user tiles by machisplin.tiles.create(feather.d = 50); every (tile, layer) is an independent mltps run
    # (reference-tiled Step 3 inside: 4 x 4 TPS tiles per user tile), merged by machisplin.tiles.merge
    "cfg4": dict(stations=5000, side=10000, layers=3, resp_layers=12, tiles=(2, 2), feather_d=50, ensemble=True, tiled=True,
                 name="cfg4: 5000 stations, 12 response layers, 10000x10000 grid, smooth.outputs.only, 2x2 machisplin.tiles.* "
                      "tiles (feather.d 50), tiles.merge"),
    "cfg4-mini": dict(stations=1200, side=1200, layers=3, resp_layers=3, tiles=(2, 2), feather_d=20, ensemble=True, tiled=True,
                      tile_edge=400, name="cfg4-mini (debug only)"),
    # a small version of cfg3 for quick checks (NOT a bench line)
    "cfg3-mini": dict(stations=1000, side=1500, layers=3, gbm_trees=500, rf_trees=50, ensemble=True,
                      name="cfg3-mini (debug only)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tps-mode", default="global", choices=["global", "tiled"],
                    help="Step 3 of the row-band workloads: one global fit (the north-star primitive, default) or the reference's "
                         "own ceil(n/1500)^2 overlapping tiles dealt over the ranks (no serial fit)")
    ap.add_argument("--in-library", action="store_true",
                    help="drive the --gpus devices from ONE process through the library's own multi-device entry points "
                         "(mhs_init_devices + mhs_mltps_grid_multi_dev / mhs_tiles_units_multi: the path the single-threaded R host "
                         "takes) instead of one process per GPU under torch.distributed; with fewer physical devices than --gpus "
                         "the slots share devices (plumbing only, flagged in the line)")
    return ap.parse_args()


class Workload:
    def __init__(self, cfg, mhs, torch, dist, rank, world, tps_mode="global"):
        from machisplin_amd import sharded, synth
        self.cfg, self.mhs, self.torch = cfg, mhs, torch
        self.rank, self.world, self.tps_mode = rank, world, tps_mode
        side, n = cfg["side"], cfg["stations"]
        self.geom = synth.grid(side, side)
        seed = synth.BASE_SEED + 3
        planes, nodata = synth.covariates(self.geom, cfg["layers"], seed, dtype="f32")
        self.stack = mhs.RasterStack(self.geom, planes, nodata)
        self.xy, rows, cols, uv = synth.stations(self.geom, n, seed)
        cov = planes[:, torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()].cpu().numpy().astype(np.float64).T
        X = np.column_stack([cov, self.xy])
        if cfg["ensemble"]:
            self.resp = synth.response(X, uv, seed)
            self.params = synth.ensemble_params(X, self.resp, seed, n_gbm_trees=cfg["gbm_trees"], n_rf_trees=cfg["rf_trees"])
            _, self.weights, self.wt_total = mhs.models.select_weights(synth.OPTX_WEIGHTS)
        else:  # TPS only: a zero "ensemble" (lm with zero coefficients, weight 1) so resp IS the residual
            self.resp = synth.tps_residual(uv, seed)
            self.params = [{"kind": "lm", "coef": np.zeros(cfg["layers"] + 3)}]
            self.weights, self.wt_total = [1.0], 1.0
        self.models = [mhs.models.from_param_dict(p) for p in self.params]
        self.X = X
        self.ops = sharded.HipOps(self.stack, self.xy, self.resp, self.models, self.weights, self.wt_total, timed=True)
        if tps_mode == "tiled":
            self.run = sharded.TiledTpsShardedMltps(PerModelOps(self.ops), dist, rank, world, side, side, tile_edge=1500)
        else:
            self.run = sharded.ShardedMltps(PerModelOps(self.ops), dist, rank, world, side, side)
        self.cells = side * side
        self.last = None
        self.rank0_share = None
        self.reservation = None
        if world == 1 and tps_mode == "global" and cfg["ensemble"]:
            # setup, untimed: how many compute units (if any) the forest leaves to the spline fit -- measured, not a constant
            self.run.calibrate_reservation()
            self.reservation = self.run.reservation_calibration
            for v in self.ops.timings.values():
                v.clear()
        if world > 1 and tps_mode == "global":
            # Load balance (setup, untimed): rank 0 also carries the spline fit, so it gets fewer rows.  One
            # calibration pass with equal bands gives this GPU's time for all cells and the stand-alone fit time;
            # rank 0 decides the share and broadcasts it so that every rank builds the same bands.
            self.run.step()
            self.collect()
            share = torch.zeros(1, dtype=torch.float64, device="cuda")
            if rank == 0:
                tm = self.ops.timings
                band_ms = sum(v[-1] for k, v in tm.items() if k.startswith("model_"))   # the TPS is evaluated on the whole grid by every rank
                cells_ms = band_ms * side / max(1, self.run.r1 - self.run.r0)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                mhs.Tps(self.ops.X[:, -2:], self.run.ops.station_residuals()[1])
                fit_ms = (time.perf_counter() - t1) * 1e3
                share[0] = sharded.balanced_rank0_share(world, cells_ms, fit_ms)
            dist.broadcast(share, src=0)
            self.rank0_share = float(share.item())
            for v in self.ops.timings.values():
                v.clear()
            self.run = sharded.ShardedMltps(PerModelOps(self.ops), dist, rank, world, side, side,
                                            rank0_share=self.rank0_share)
            self.run.calibrate_reservation()      # collective; rank 0 keeps what it measured fastest
            self.reservation = self.run.reservation_calibration
            for v in self.ops.timings.values():
                v.clear()

    def step(self):
        self.last = self.run.step()

    def collect(self):
        self.torch.cuda.synchronize()
        self.ops.collect()

    # ---- algorithmic work per launch of the heavy kernels ----------------
    def kernel_table(self):
        ops = self.ops
        band_cells = (self.run.r1 - self.run.r0) * self.geom.ncol
        n = ops.X.shape[0]
        rows = []
        pmc = pmc_derived()

        def mean_ms(key):
            v = ops.timings.get(key, [])
            v = v[-max(1, len(v) // 2):]
            return float(np.mean(v)) if v else None

        ms = mean_ms("tps_eval_ms")
        if ms:
            ens_band_cells, band_cells = band_cells, self.geom.nrow * self.geom.ncol   # the spline covers the whole grid on every rank
            tc, tr, node_pairs, cell_pairs = getattr(ops, "last_eval_plan", (0, 0, 0, 0))
            if tc:   # far-field-interpolated path: kernel evaluations actually performed + 32 FMA/cell of interpolation
                fl = 8.0 * (node_pairs + cell_pairs) + band_cells * 70.0
                rows.append({"kernel": "tps_ff_nodes_kernel+tps_ff_cells_kernel", "bound": "fp64-valu", "launch_ms": ms,
                             "achieved": fl / ms / 1e9, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "work": "far-field-interpolated sum, tiles %d x %d cells: 8 flop per evaluated (point, knot) pair "
                                     "(%.3g pairs at tile nodes + %.3g at cells, vs %.3g for the direct sum) + 70 flop/cell of "
                                     "Chebyshev interpolation; FP64 VALU-bound" % (tc, tr, node_pairs, cell_pairs, band_cells * float(n)),
                             "direct_sum_equivalent_tflops": band_cells * (8.0 * n + 6.0) / ms / 1e9})
            else:
                fl = band_cells * (8.0 * n + 6.0)
                rows.append({"kernel": "tps_eval_grid_kernel", "bound": "fp64-valu", "launch_ms": ms, "achieved": fl / ms / 1e9,
                             "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "work": "8N+6 flop/cell, log = 1 flop; FP64 VALU/"
                             "transcendental-bound (FP64 MFMA shares the DP pipe, same peak)"})
        if mean_ms("tps_eval_ms"):
            band_cells = ens_band_cells
        fused = [kk for kk in ops.timings if kk.startswith("model_") and "+" in kk]
        for fk in fused:
            ms = mean_ms(fk)
            if ms:
                by = band_cells * (4.0 * self.cfg["layers"] + 16.0)
                pm = pmc.get("small", {})
                # ten logistic units per cell (nnet, V73:468) + 15 hinges + the linear model: compute, not traffic, bounds it
                ipc = pm.get("valu_per_cell_unit", 420.0)
                rows.append({"kernel": "small_members_kernel (%s)" % fk[6:-3], "bound": "valu-issue", "launch_ms": ms,
                             "achieved": band_cells / 64.0 * ipc / ms / 1e6, "peak": VALU_ISSUE_PEAK_G, "unit": "Gwave-instr/s",
                             "work": "gam + nnet + earth in one pass: %.0f VALU wave-instructions per cell / 64 lanes (%s); 20 B/cell of HBM "
                                     "traffic (C fp32 planes read once, the fp64 plane read-modify-written once)" % (
                                         ipc, "SQ_INSTS_VALU, " + pmc["_source"] if "valu_per_cell_unit" in pm else "counted in the source, no PMC pass"),
                             "pmc": {kk: pm[kk] for kk in ("valu_issue_utilisation",) if kk in pm},
                             "hbm_view": {"achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS}})
        for prm in self.params:
            k = prm["kind"]
            ms = mean_ms("model_%s_ms" % k)
            if not ms:
                continue
            if k == "svr":
                nsv, p = prm["sv"].shape
                fl = band_cells * nsv * (3.0 * p + 2.0)
                pm = pmc.get("svr", {})
                rowtile = self.geom.ncol >= 0.93 * (-(-self.geom.ncol // 192) * 192)      # launch_svr's choice (ensemble.hip)
                ipp = pm.get("valu_per_cell_unit", 14.4 if rowtile else 15.07)      # measured (SQ_INSTS_VALU); the defaults are round 3's counts
                rows.append({"kernel": "svr_rt_kernel" if rowtile else "svr_kernel", "bound": "fp64-valu", "launch_ms": ms, "achieved": fl / ms / 1e9,
                             "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "work": "(3p+2) flop per (cell, SV), exp = 1 flop",
                             "issue_view": {"achieved": band_cells * nsv / 64.0 * ipp / ms / 1e6, "peak": VALU_ISSUE_PEAK_G, "unit": "Gwave-instr/s",
                                            "frac": band_cells * nsv / 64.0 * ipp / ms / 1e6 / VALU_ISSUE_PEAK_G,
                                            "work": "%.2f VALU wave-instructions per (cell, SV) / 64 lanes (SQ_INSTS_VALU, %s), "
                                                    "12 of them FP64" % (ipp, pmc.get("_source", "round-3 count"))},
                             "pmc": {kk: pm[kk] for kk in ("valu_issue_utilisation", "lds_array_busy") if kk in pm}})
            elif k == "gbm":
                # Grids: gbm_coherent_kernel (a probe on the device prices it against the tree-order row-tile kernel).  Both are
                # bound by the rate at which a SIMD issues VALU instructions (one wave64 instruction per 4 cycles); achieved =
                # VALU wave-instructions per second, the count per (cell, tree) from the SQ_INSTS_VALU pass on the torch-free
                # driver over the same rasters (tree-order kernel: 6.4; coherent: ~1 on cfg3's rasters -- it depends on them).
                nt = len(prm["tree_offsets"]) - 1
                grid_ok = prm["p"] <= 8 and self.geom.nrow >= 8
                pm = pmc.get("gbm", {}) if grid_ok else {}
                probe = None
                try:
                    import ctypes as C
                    from machisplin_amd import _lib
                    cst, cnt = C.c_int64(0), C.c_int64(0)
                    mod = self.models[[q["kind"] for q in self.params].index("gbm")]
                    _lib.check(_lib.lib().mhs_gbm_probe_last(mod._h, C.byref(cst), C.byref(cnt)))
                    if cnt.value:
                        probe = {"estimated_cost_vs_tree_order_kernel": 0.12 + cst.value / (100.0 * cnt.value), "coherent_kernel_ran": cst.value < 83 * cnt.value,
                                 "sample": "%d (tile, tree) pairs of the window classified on the device before the launch" % cnt.value}
                except Exception:
                    pass
                coherent = grid_ok and (probe is None or probe["coherent_kernel_ran"])
                ipt = pm.get("valu_per_cell_unit", 1.0 if coherent else 6.4)
                inst = band_cells * nt / 64.0 * ipt
                rows.append({"kernel": "gbm_coherent_kernel" if coherent else "gbm_lutreg_rt_kernel", "bound": "valu-issue", "launch_ms": ms,
                             "achieved": inst / ms / 1e6, "peak": VALU_ISSUE_PEAK_G, "unit": "Gwave-instr/s",
                             "work": "%.2f VALU wave-instructions per (cell, tree) / 64 lanes (SQ_INSTS_VALU pass, "
                                     "%s -- a property of those rasters): per wave of 64 x 4 cells a tree whose splits fall the same way for every cell is "
                                     "summed once (lane = tree), one or two straddling splits cost a clamp-add and a multiply-add per cell; "
                                     "reference walk = %.0f node visits/cell, %.3g visits/s.  The instruction mix is mostly FP32 (2-cycle issue), so "
                                     "`frac` is the PMC pass's measured VALU utilisation when there is one; the count-based figure is in issue_view" % (
                                         ipt, pmc.get("_source", "no PMC file"), self.mean_visits[k], self.mean_visits[k] * band_cells / (ms * 1e-3)),
                             "issue_view": {"achieved": inst / ms / 1e6, "peak": VALU_ISSUE_PEAK_G, "unit": "Gwave-instr/s", "frac": inst / ms / 1e6 / VALU_ISSUE_PEAK_G,
                                            "work": "count x rate against one wave64 instruction per 4 cycles (an upper reading for FP32-heavy code)"},
                             "frac_override": pm.get("valu_issue_utilisation"),
                             "node_visits_per_s": self.mean_visits[k] * band_cells / (ms * 1e-3),
                             "probe": probe,
                             "pmc": {kk: pm[kk] for kk in ("valu_issue_utilisation", "salu_per_valu", "lds_array_busy") if kk in pm}})
            elif k == "rf":
                # LDS-bound walk: per lane, tree and level one ds_read_b64 (node) + one ds_read_b32 (rank key), 2 LDS-array
                # cycles each per wave when conflict-free (MI355X guide, LDS table); every lane descends each tree's full depth
                by = band_cells * (4.0 * self.cfg["layers"] + 16.0)
                full = float(self.rf_level_sum())
                prm = next(p for p in self.params if p["kind"] == "rf")
                big = int(np.diff(prm["tree_offsets"]).max()) > 4095      # launch_forest's choice (forest.hip)
                pm = pmc.get("rf", {}) if not big else {}
                # a wave leaves a tree when all its walks sit at terminal nodes: the levels really walked come from the
                # SQ_INSTS_LDS pass on the same rasters (two LDS instructions per walk and level), scaled to this forest's depth
                walked_share = pm["levels_walked_per_cell"] / pm["levels_full_depth_per_cell"] if "levels_walked_per_cell" in pm else 1.0
                levels = full * min(1.0, walked_share)
                cyc = band_cells * levels / 64.0 * 4.0
                tb = not big and int(np.diff(prm["tree_offsets"]).max()) * 8 <= 25600      # rf_walk_ld_config (forest.hip)
                rows.append({"kernel": "rf_walk_cbs_kernel" if big else "rf_walk_ld_kernel" if tb else "rf_walk_db_kernel", "bound": "lds", "launch_ms": ms,
                             "achieved": cyc / ms / 1e6, "peak": LDS_CYCLE_PEAK_G, "unit": "G LDS-cycles/s",
                             "frac_is_upper_reading": bool(big),
                             "work": ("trees beyond 4 095 nodes: the block-subtree kernel (round 6) -- no PMC pass on it, so the figure prices every tree "
                                      "at its FULL depth, which the kernel does not walk (tools/r06_forest_stats.py on a forest of this shape and the 8d "
                                      "planes: 11 levels a tree skipped above the wave's entry, 2 - 4 walked below it): an upper reading, not a utilisation.  "
                                      if big else "") +
                                     "4 conflict-free LDS-array cycles per wave, tree level WALKED and walk (ds_read_b64 node + ds_read_b32 key): "
                                     "%.0f levels/cell walked (a wave of neighbouring cells starts a tree where its cells part ways and leaves it at its deepest leaf; "
                                     "share of the full depth from SQ_INSTS_LDS / 2 in %s -- measured on those rasters, not in this run) "
                                     "of %d levels/cell of full tree depth; %.0f node visits/cell on the reference's walk, %.3g visits/s" % (
                                         levels, pmc.get("_source", "no PMC file: full depth assumed"), full, self.mean_visits[k], self.mean_visits[k] * band_cells / (ms * 1e-3)),
                             "node_visits_per_s": self.mean_visits[k] * band_cells / (ms * 1e-3),
                             "pmc": {kk: pm[kk] for kk in ("lds_array_busy", "lds_bank_conflict_share_of_lds_cycles", "lds_cycles_per_lds_instruction",
                                                           "lds_cmd_fifo_full_share", "valu_issue_utilisation") if kk in pm},
                             "hbm_view": {"achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS},
                             "lds_bytes_view": {"achieved": band_cells * levels * 12.0 / ms / 1e6, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                                                "frac": band_cells * levels * 12.0 / ms / 1e6 / LDS_PEAK_GBS,
                                                "work": "rounds 1-2: 12 B of LDS reads per lane and level against 128 B/clk/CU"}})
            else:
                by = band_cells * (4.0 * self.cfg["layers"] + 16.0)
                rows.append({"kernel": "%s_kernel" % k, "bound": "hbm", "launch_ms": ms, "achieved": by / ms / 1e6,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "work": "read C fp32 planes + read-modify-write fp64 out"})
        pmc_t = {}
        try:  # bytes per cell measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH doubled as
            # the gfx950 guide prescribes) on tools/pmc_probe; committed under profiles/
            pmc_file = next(p for p in (os.path.join(ROOT, "profiles", n) for n in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"))
                            if os.path.exists(p))
            with open(pmc_file) as f:
                for kn, d in json.load(f)["kernels"].items():
                    pmc_t[kn.split("::")[-1].split("<")[0]] = d["fetch_bytes_per_cell_x2_corrected"] + d["write_bytes_per_cell"]
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        for kind, kn in (("gbm", "gbm_lutreg_rt_kernel"), ("gbm", "gbm_coherent_kernel"), ("rf", "rf_walk_db_kernel"), ("rf", "rf_walk_tb_kernel"), ("rf", "rf_walk_ld_kernel"), ("svr", "svr_kernel"), ("svr", "svr_rt_kernel")):
            if "hbm_bytes_per_cell_fetch_x2_plus_write" in pmc.get(kind, {}):      # round 3's passes (fresh output plane: no RMW read)
                pmc_t[kn] = pmc[kind]["hbm_bytes_per_cell_fetch_x2_plus_write"]
        for r in rows:
            r["frac"] = r["achieved"] / r["peak"]
            if r.get("frac_override") is not None:      # gbm: the measured pipe utilisation leads, the count-based figure stays in issue_view
                r["frac"] = r.pop("frac_override")
            else:
                r.pop("frac_override", None)
            r["pmc_source"] = pmc.get("_source")
            parts = [pmc_t.get(kn.split("<")[0]) for kn in r["kernel"].split("+")]
            r["traffic"] = sum(parts) * band_cells if all(x is not None for x in parts) else None
        return rows

    def f64_boundary(self):
        """The ensemble pass with float64 planes: (a) resident in HBM through the _dev entry point, (b) through the
        host-pointer entry point the R shim binds (covariates and result cross PCIe inside the call)."""
        import ctypes as C
        from machisplin_amd import _lib
        torch, mhs = self.torch, self.mhs
        g = self.geom
        # without mhs_fit_reserve_cus (a setting of the Python drivers, not of the R shim): its CU-masked stream is a
        # BLOCKING stream, and the runtime's pageable copies then wait for the forest instead of running under it
        reserved = self.ops.reserve(0)
        try:
            return self._f64_boundary(torch, mhs, g)
        finally:
            if reserved:
                self.ops.reserve(reserved)

    def _f64_boundary(self, torch, mhs, g):
        import ctypes as C
        from machisplin_amd import _lib
        # on a stream of its own, as the timed steps run: launches on the NULL stream synchronise with every blocking stream
        # (the CU-masked streams of the reservation are blocking ones, whether the reservation is in use or not)
        t32, t64 = [], []
        with self.ops.side_stream():
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                a = mhs.ensemble_predict(self.stack, self.models, self.weights, self.wt_total)
                torch.cuda.synchronize(); t32.append((time.perf_counter() - t0) * 1e3)
            stack64 = mhs.RasterStack(g, self.stack.planes.to(torch.float64), self.stack.nodata)
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                b = mhs.ensemble_predict(stack64, self.models, self.weights, self.wt_total)
                torch.cuda.synchronize(); t64.append((time.perf_counter() - t0) * 1e3)
        self.ops.join_side_stream()
        same = bool(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)))
        host = np.ascontiguousarray(stack64.planes.cpu().numpy())
        del stack64, b
        out = np.empty((g.nrow, g.ncol))
        hs = (C.c_void_p * len(self.models))(*[m._h for m in self.models])
        ws = (C.c_double * len(self.models))(*[float(w) for w in self.weights])
        st = _lib.Stack(host.ctypes.data, host.shape[0], _lib.F64, g.nrow * g.ncol, g.ncol, float("nan"))
        gs = g.c_struct()
        th = []
        for _ in range(2):
            t0 = time.perf_counter()
            _lib.check(_lib.lib().mhs_ensemble_predict(hs, ws, len(self.models), float(self.wt_total), C.byref(gs), C.byref(st),
                                                       0, g.nrow, 0, g.ncol, out.ctypes.data))
            th.append((time.perf_counter() - t0) * 1e3)
        same_host = bool(np.array_equal(np.nan_to_num(out), np.nan_to_num(a.cpu().numpy())))
        nbytes = host.nbytes + out.nbytes
        return {"f32_resident_ensemble_ms": min(t32), "f64_resident_ensemble_ms": min(t64),
                "f64_over_f32": min(t64) / min(t32), "f64_planes_equal_f32_planes_bitwise": same,
                "host_abi_f64_ms": min(th), "host_abi_bytes_over_pcie": nbytes,
                "host_abi_equals_resident_bitwise": same_host,
                "note": "f64 planes = the f32 planes widened (same values), what terra holds in RAM; host ABI = "
                        "mhs_ensemble_predict with pageable host buffers, as mhsr_ensemble_predict calls it"}

    def raster_sensitivity(self, side=4000):
        """How much of the heavy members' speed is a property of the synthetic rasters (round-3 verdict, item 2).  The three
        tree / kernel members are timed, outside the timed loop, on the NW `side` x `side` window of this workload's grid
        (LONG / LAT unchanged) holding (a) the SURVEY 8d planes the steps run on, (b) the reference's OWN rasters -- the bundled
        TWI and slope overviews (tests/golden/cfg1_extdata.npz, /root/reference/README.md:84-96, inst/extdata/*.aux.xml;
        1238 x 1632 INT2S, mirrored into a mosaic; alt.tif is a missing blob, its stand-in is made of the same two rasters) --, (c) the 8d planes
        plus white noise of 1 / 10 / 100 % of each covariate's range.  ms per 1e8 cells; for the forest also round 2's walk
        (no prefix, full depth, far walks, two buffers), for ksvm also the lane-per-cell kernel, for gbm the probe's verdict."""
        import ctypes as C
        from machisplin_amd import _lib, synth
        torch, mhs = self.torch, self.mhs
        if not self.cfg["ensemble"] or self.geom.nrow < side or self.geom.ncol < side or self.cfg["layers"] != 3:
            return None
        g = synth.grid(side, side)
        base = self.stack.planes[:, :side, :side].contiguous()
        variants = [("survey_8d_planes", base, "the planes of the timed steps (six sinusoids of 0.5-6 cycles across the 10 000-cell side)")]
        fixture = os.path.join(ROOT, "tests", "golden", "cfg1_extdata.npz")
        if os.path.exists(fixture):
            d = np.load(fixture)
            def mosaic(a):
                a = a.astype(np.float32)
                a[a == -32768] = np.nan
                ny, nx = -(-side // a.shape[0]), -(-side // a.shape[1])
                rows = []
                for iy in range(ny):
                    t = a[::-1] if iy & 1 else a
                    rows.append(np.concatenate([t[:, ::-1] if ix & 1 else t for ix in range(nx)], axis=1))
                return np.ascontiguousarray(np.concatenate(rows, axis=0)[:side, :side])
            real = base.clone()
            real[1] = torch.from_numpy(mosaic(d["slope"])).cuda()
            real[2] = torch.from_numpy(mosaic(d["TWI"])).cuda()
            # alt.tif is a missing blob: its stand-in is built from the bundled rasters themselves (round-5 verdict item 5: a
            # synthetic sinusoid here let a fitted gbm, 75 % of whose importance sits on alt, exploit the generator's
            # coherence) -- the TWI overview rotated by 180 degrees and shifted by half a tile (decorrelated from plane 2, same
            # spatial spectrum), minus a tenth of the slope, rescaled linearly to alt's own range (alt.tif.aux.xml: 76 .. 4668)
            twi, slp = d["TWI"].astype(np.float32), d["slope"].astype(np.float32)
            twi[twi == -32768] = np.nan; slp[slp == -32768] = np.nan
            sub = np.roll(twi[::-1, ::-1], (twi.shape[0] // 2, twi.shape[1] // 2), axis=(0, 1)) - 0.1 * np.roll(slp, twi.shape[1] // 3, axis=1)
            lo_, hi_ = np.nanmin(sub), np.nanmax(sub)
            a_lo, a_hi = synth.COV_RANGES[0]
            sub = (sub - lo_) / (hi_ - lo_) * (a_hi - a_lo) + a_lo
            sub[np.isnan(sub)] = -32768
            real[0] = torch.from_numpy(mosaic(sub)).cuda()
            variants.append(("bundled_twi_slope_overviews", real, "the reference's bundled TWI.tif / slope.tif overviews (1238 x 1632 INT2S, NoData -> NA), "
                             "mirrored into a %d x %d mosaic; alt (alt.tif is not in the repository): a stand-in made of the SAME rasters -- the TWI "
                             "overview rotated by 180 degrees and shifted by half a tile minus a tenth of the slope, rescaled to alt's range" % (side, side)))
        gen = torch.Generator(device="cuda")
        gen.manual_seed(7)
        for frac in (0.01, 0.1, 1.0):
            noisy = base.clone()
            for k in range(base.shape[0]):
                lo, hi = synth.COV_RANGES[k]
                noisy[k] += (torch.rand((side, side), device="cuda", generator=gen) - 0.5) * (frac * (hi - lo))
            variants.append(("survey_8d_plus_%g%%_white_noise" % (100 * frac), noisy, "uniform noise of %g %% of the covariate's range on every cell" % (100 * frac)))
        kinds = [p["kind"] for p in self.params]
        out = torch.empty((side, side), dtype=torch.float64, device="cuda")
        scale = 1e8 / (side * side)

        def timed(stack, model, env=()):
            env = dict(e if isinstance(e, tuple) else (e, "1") for e in env)
            for e, v in env.items():
                os.environ[e] = v
            try:
                mhs.predict(stack, model, out=out)
                torch.cuda.synchronize()
                best = 1e30
                for _ in range(2):
                    t1 = time.perf_counter()
                    mhs.predict(stack, model, out=out)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t1) * 1e3)
            finally:
                for e in env:
                    del os.environ[e]
            return best * scale

        rows = []
        small_ms = sum(float(np.mean(v[-max(1, len(v) // 2):])) for k, v in self.ops.timings.items()
                       if k.startswith("model_") and "+" in k and v) * 1e8 / self.cells
        spline_ms = (float(np.mean(self.ops.timings["tps_eval_ms"][-2:])) if self.ops.timings.get("tps_eval_ms") else 0.0) * 1e8 / self.cells
        for name, planes, what in variants:
            stack = mhs.RasterStack(g, planes, float("nan"))
            row = {"rasters": name, "what": what}
            for kind, key in (("gbm", "gbm"), ("rf", "forest"), ("svr", "ksvm")):
                model = self.models[kinds.index(kind)]
                row[key + "_ms_per_1e8_cells"] = timed(stack, model)
                if kind == "gbm":
                    cst, cnt = C.c_int64(0), C.c_int64(0)
                    _lib.check(_lib.lib().mhs_gbm_probe_last(model._h, C.byref(cst), C.byref(cnt)))
                    if cnt.value:
                        row["gbm_probe"] = {"estimated_cost_vs_tree_order_kernel": 0.12 + cst.value / (100.0 * cnt.value),
                                            "coherent_kernel_ran": bool(cst.value < 83 * cnt.value)}
                    row["gbm_tree_order_kernel_ms_per_1e8_cells"] = timed(stack, model, ("MHS_GBM_NO_COHERENT",))
                elif kind == "rf":
                    # the plain walk: double-buffered kernel, every tree from the root to its full depth (round 2's, minus its cell order)
                    row["forest_plain_walk_ms_per_1e8_cells"] = timed(stack, model, (("MHS_RF_KERNEL", "db"), "MHS_RF_PLAIN"))
                else:
                    row["ksvm_lane_per_cell_kernel_ms_per_1e8_cells"] = timed(stack, model, ("MHS_SVR_NO_ROWTILE",))
            heavy = row["gbm_ms_per_1e8_cells"] + row["forest_ms_per_1e8_cells"] + row["ksvm_ms_per_1e8_cells"]
            row["ensemble_plus_spline_mcells_per_s_estimate"] = 1e8 / ((heavy + small_ms + spline_ms) * 1e-3) / 1e6
            rows.append(row)
            del stack
        # (d) FITTED tree structures (round-4 verdict item 4): synth.gbm_params / rf_params grow random structures with
        # stagewise-fitted leaves; a real gbm.step / randomForest model concentrates its splits on the predictors that explain the
        # response, which changes the coherent kernel's hit rate and the forest's prefix depth.  scikit-learn's trainers on the same
        # station table (GradientBoostingRegressor: max_leaf_nodes = 6 as V73:493's depth-5 trees, learning rate 0.001, as many trees
        # as the workload; RandomForestRegressor: 500 trees, min_samples_split = 6 for randomForest's nodesize 5, max_features = p / 3),
        # exported to the flat layouts (tests/modelgen.py), timed on the 8d planes and on the reference's rasters.
        fitted = None
        if not os.environ.get("MHS_BENCH_SKIP_FITTED"):
            try:
                t1 = time.perf_counter()
                from sklearn.ensemble import GradientBoostingRegressor, RandomForestRegressor
                from tests import modelgen
                p_ = self.X.shape[1]
                gbr = GradientBoostingRegressor(n_estimators=self.cfg["gbm_trees"], max_leaf_nodes=6, max_depth=None, learning_rate=0.001,
                                                subsample=0.5, random_state=1).fit(self.X, self.resp)
                rfr = RandomForestRegressor(n_estimators=self.cfg["rf_trees"], min_samples_split=6, max_features=max(p_ // 3, 1), n_jobs=8,
                                            random_state=1).fit(self.X, self.resp)
                mg = mhs.models.from_param_dict(modelgen.gbm_from_sklearn(gbr, p_))
                mr = mhs.models.from_param_dict(modelgen.rf_from_sklearn(rfr, p_))
                imp_g = [float(v) for v in gbr.feature_importances_]
                fit_s = time.perf_counter() - t1
                frows = []
                for name, planes, what in variants[:2]:
                    stack = mhs.RasterStack(g, planes, float("nan"))
                    r = {"rasters": name, "gbm_ms_per_1e8_cells": timed(stack, mg), "forest_ms_per_1e8_cells": timed(stack, mr)}
                    cst, cnt = C.c_int64(0), C.c_int64(0)
                    _lib.check(_lib.lib().mhs_gbm_probe_last(mg._h, C.byref(cst), C.byref(cnt)))
                    if cnt.value:
                        r["gbm_probe"] = {"estimated_cost_vs_tree_order_kernel": 0.12 + cst.value / (100.0 * cnt.value),
                                          "coherent_kernel_ran": bool(cst.value < 83 * cnt.value)}
                    r["gbm_tree_order_kernel_ms_per_1e8_cells"] = timed(stack, mg, ("MHS_GBM_NO_COHERENT",))
                    ks = next(rw["ksvm_ms_per_1e8_cells"] for rw in rows if rw["rasters"] == name)
                    r["ensemble_plus_spline_mcells_per_s_estimate"] = 1e8 / ((r["gbm_ms_per_1e8_cells"] + r["forest_ms_per_1e8_cells"] + ks + small_ms + spline_ms) * 1e-3) / 1e6
                    frows.append(r)
                    del stack
                fitted = {"trainers": "scikit-learn %s: GradientBoostingRegressor(n_estimators=%d, max_leaf_nodes=6, learning_rate=0.001, subsample=0.5), "
                                      "RandomForestRegressor(n_estimators=%d, min_samples_split=6, max_features=p/3) on the workload's %d stations; fitted in %.0f s on the host"
                                      % (__import__("sklearn").__version__, self.cfg["gbm_trees"], self.cfg["rf_trees"], self.X.shape[0], fit_s),
                          "gbm_feature_importances": imp_g, "rf_nodes_per_tree": float(np.mean([e.tree_.node_count for e in rfr.estimators_])),
                          "variants": frows}
                del mg, mr
            except Exception as e:      # the line is still a line without this block
                fitted = {"error": repr(e)}
        return {"fitted_models": fitted,
                "window": "NW %d x %d cells of the workload's grid, float32 planes resident in HBM; ms per 1e8 cells = time x %.4g" % (side, side, scale),
                "estimate": "the three heavy members here + this run's fused small members (%.1f ms) and spline evaluation (%.1f ms) per 1e8 cells; the forest "
                            "un-masked (no compute units reserved for the fit)" % (small_ms, spline_ms),
                "variants": rows}

    def phase_table(self):
        """COLLECTIVE (every rank calls it): per-rank phase times [band, spline evaluation, fit inside the step,
        station residuals, rows] gathered on every rank."""
        import torch.distributed as dist
        torch = self.torch
        tm = self.ops.timings
        mean = lambda v: float(np.mean(v[-max(1, len(v) // 2):])) if v else 0.0
        band_ms = sum(mean(v) for k, v in tm.items() if k.startswith("model_"))
        mine = torch.tensor([band_ms, mean(tm.get("tps_eval_ms", [])), mean(tm.get("tps_fit_ms", [])),
                             mean(tm.get("residuals_ms", [])), float(self.run.r1 - self.run.r0)], dtype=torch.float64, device="cuda")
        allv = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allv, mine)
        return torch.stack(allv).cpu().numpy()

    def model_check(self, allv, fit_ms, step_ms):
        """Predicted step = max over ranks of (band [+ rank 0's fit]) + the spline on the rank's own rows + the one all-gather of
        the output plane + Step 5 (round 6's flow), against the observed step."""
        gather_bytes = self.run.band * self.geom.ncol * 8 * (self.world - 1)
        gather_ms = gather_bytes / 153e9 * 1e3 / max(1, min(7, self.world - 1))   # direct mesh: one xGMI link per peer
        pred0 = allv[0, 0] + fit_ms
        pred = max(pred0, float(allv[1:, 0].max())) + float(allv[:, 1].max()) + gather_ms + 3.0
        return {"band_ms_per_rank": allv[:, 0].tolist(), "rows_per_rank": allv[:, 4].astype(int).tolist(),
                "tps_eval_ms_per_rank": allv[:, 1].tolist(), "fit_ms_in_step_rank0": float(allv[0, 2]),
                "fit_ms_standalone": fit_ms, "rank0_share": self.rank0_share,
                "gather_ms_model": gather_ms, "predicted_step_ms": pred, "observed_step_ms": step_ms}

    def rf_level_sum(self):
        """sum over the forest's trees of the tree depth (levels every lane descends in rf_walk_kernel)."""
        if getattr(self, "_rf_levels", None) is None:
            prm = next(p for p in self.params if p["kind"] == "rf")
            off, total = prm["tree_offsets"], 0
            for t in range(len(off) - 1):
                o, cnt = int(off[t]), int(off[t + 1] - off[t])
                L, R, st = prm["left"][o:o + cnt] - 1, prm["right"][o:o + cnt] - 1, prm["status"][o:o + cnt]
                d = np.zeros(cnt, dtype=np.int64)
                for kk in np.flatnonzero(st != -1):     # children are numbered after their parent
                    d[L[kk]] = d[R[kk]] = d[kk] + 1
                total += int(d.max())
            self._rf_levels = total
        return self._rf_levels

    def measure_mean_visits(self):
        """mean node visits per cell of the tree members (host walk over the station sample)."""
        from machisplin_amd import synth
        self.mean_visits = {}
        for prm in self.params:
            if prm["kind"] in ("gbm", "rf"):
                self.mean_visits[prm["kind"]] = synth.mean_tree_visits(prm, self.X[:256])

    def cpu_baseline(self, ms_per_step=None, direct_sum_ms=None, farfield_ms=None):
        """kind 'port' (no R on the box -- probed below): the oracle's C restatement
        of the six predict methods and of predict.Krig's direct radial sum, OpenMP over cells, plus the numpy
        QR + eigen + GCV fit, on a bounded row band extrapolated linearly to the grid.  Threads = what one socket
        offers, capped by the cgroup CPU quota; the same band on ONE core (how the reference itself runs,
        V73:117) and the reference-tiled Step 3 (V73:656-753: ceil(n/1500)^2 tile fits + evaluations) beside it."""
        import shutil
        import subprocess
        from oracle import cbind, ensemble as oe, tiles as ot, tps as otps
        host = host_info()
        threads = host["threads_used"]
        g = self.geom
        X, y = self.ops.X, self.ops.y
        res = None
        for p, w in zip(self.params, self.weights):
            rk = (y - cbind.predict(p, X, threads)) * w
            res = rk if res is None else res + rk
        res = res / self.wt_total
        t0 = time.perf_counter()
        m = otps.fit(X[:, -2:], res)
        t_fit = time.perf_counter() - t0
        hostp = None

        def band(rows, nthreads, keep=False):
            nonlocal hostp
            cov = self.stack.planes[:, :rows].cpu().numpy().astype(np.float64)
            xs, ys = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol, 0, rows)
            Xg = oe.stack_predictors(cov, (xs, ys))
            t0 = time.perf_counter()
            pred = cbind.ensemble(self.params, self.weights, self.wt_total, Xg, nthreads)
            t_ens = time.perf_counter() - t0
            tps = cbind.tps_eval_grid(m, g.xmin, g.ymax, g.xres, g.yres, 0, rows, 0, g.ncol, threads=nthreads)
            t_all = time.perf_counter() - t0
            if keep:
                hostp = pred.reshape(rows, g.ncol) + tps
            return t_all, t_ens

        probe = max(1, min(g.nrow, 64 * 10000 // g.ncol // 8))
        t_probe, _ = band(probe, threads)
        rows = int(min(g.nrow, max(probe, probe * 10.0 / max(t_probe, 1e-3))))
        t_band, t_band_ens = band(rows, threads, keep=True)
        t_cells = t_band * (g.nrow / rows)
        # the CPU sample doubles as a full-size spot check of the GPU result
        gpu = self.last["final"][:rows].cpu().numpy()
        err = float(np.nanmax(np.abs(gpu - hostp)) / np.nanmax(np.abs(hostp)))
        # one core: the reference's own mode (n.cores is forced to 1, V73:117)
        rows1 = max(1, int(rows * 6.0 / max(t_band * threads, 1e-3)))
        t_band1, _ = band(rows1, 1)
        t_cells1 = t_band1 * (g.nrow / rows1)
        out = {"value": self.cells / (t_fit + t_cells) / 1e6, "unit": "Mcells/s", "cores": threads, "kind": "port",
               "sample": f"C restatement (OpenMP, {threads} threads) of the 6-member ensemble + direct-sum TPS evaluation on {rows} of "
                         f"{g.nrow} rows ({t_band:.1f} s, extrapolated linearly to the grid: {t_cells:.0f} s) + numpy QR/eigen/GCV fit of "
                         f"{X.shape[0]} stations ({t_fit:.1f} s); global TPS mode, as the GPU step",
               "host": host,
               "fit_s": t_fit, "cells_s_full_grid": t_cells, "gpu_vs_cpu_sample_max_rel_err": err,
               "one_core": {"value": self.cells / (t_fit + t_cells1) / 1e6, "unit": "Mcells/s", "cores": 1,
                            "sample": f"{rows1} rows in {t_band1:.1f} s on one thread, extrapolated: {t_cells1:.0f} s (+ the same fit)"}}
        # reference-tiled Step 3 (what V73 really does above 1500 px): every tile's fit, and the evaluation of one
        # row of tiles extrapolated to all rows of tiles; Step 2 is the band figure above
        og = ot.Geom(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)
        nRx, nCx, fw, kw = ot.step3_windows(og, 1500)
        if nRx * nCx > 1:
            cov1 = np.zeros((1, 1))   # stations_in_window only needs "not NA": the synthetic planes have no NA at stations
            knots = X[:, -2:]
            rowsS = np.array([og.row_from_y(v) for v in knots[:, 1]])
            colsS = np.array([og.col_from_x(v) for v in knots[:, 0]])
            t0 = time.perf_counter()
            fits = []
            for h in range(nRx * nCx):
                r0, r1, c0, c1 = fw[h]
                sel = np.flatnonzero((rowsS >= r0) & (rowsS < r1) & (colsS >= c0) & (colsS < c1))
                fits.append(otps.fit(knots[sel], res[sel]) if sel.size >= 10 else None)
            t_tfit = time.perf_counter() - t0
            t0 = time.perf_counter()
            for h in range(nCx):       # the southern row of tiles
                if fits[h] is None:
                    continue
                gf = ot.window_geom(og, fw[h])
                wk = (kw[h][0] - fw[h][0], kw[h][1] - fw[h][0], kw[h][2] - fw[h][2], kw[h][3] - fw[h][2])
                cbind.tps_eval_grid(fits[h], gf.xmin, gf.ymax, gf.xres, gf.yres, *wk, threads=threads)
            t_teval = (time.perf_counter() - t0) * nRx
            t_ens_full = t_band_ens * (g.nrow / rows)
            out["reference_tiled"] = {"value": self.cells / (t_tfit + t_teval + t_ens_full) / 1e6, "unit": "Mcells/s", "cores": threads,
                                      "tiles": [int(nRx), int(nCx)],
                                      "sample": f"{nRx * nCx} tile fits in full ({t_tfit:.1f} s) + evaluation of one row of {nCx} tiles x {nRx} "
                                                f"({t_teval:.1f} s) + the ensemble band above ({t_ens_full:.0f} s); mosaic / feather not timed"}
        # the reference itself, if the box had R
        rs = shutil.which("Rscript")
        if rs:
            try:
                pr = subprocess.run([rs, "-e", "library(fields);library(terra);cat(as.character(packageVersion('fields')))"],
                                    capture_output=True, text=True, timeout=60)
                out["rscript"] = {"path": rs, "rc": pr.returncode, "out": (pr.stdout + pr.stderr)[-200:]}
            except Exception as e:  # noqa: BLE001
                out["rscript"] = {"path": rs, "error": str(e)}
        else:
            out["rscript"] = "not installed (which Rscript: empty): the reference itself cannot be timed on this box"
        if ms_per_step and direct_sum_ms and farfield_ms:
            like = ms_per_step - farfield_ms + direct_sum_ms    # the GPU step with predict.Krig's own direct sum
            out["gpu_step_ms_with_direct_sum_tps"] = like
            out["gpu_over_cpu_like_for_like"] = (t_fit + t_cells) * 1e3 / like
        if ms_per_step:
            out["gpu_over_cpu"] = (t_fit + t_cells) * 1e3 / ms_per_step
            out["gpu_over_one_core"] = (t_fit + t_cells1) * 1e3 / ms_per_step
        return out


def host_info():
    """CPU model, sockets, cores per socket, visible CPUs, cgroup CPU quota, and the thread count the CPU baseline
    uses: min(cores of ONE socket, quota) -- the 'single-socket CPU wall-clock' of the north-star target."""
    info = {"model": None, "sockets": None, "cores_per_socket": None, "nproc": os.cpu_count(), "cgroup_cpu_quota": None}
    try:
        phys, cores, model = set(), {}, None
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [t.strip() for t in line.split(":", 1)]
                cur[k] = v
            elif cur:
                phys.add(cur.get("physical id", "0")); cores[cur.get("physical id", "0")] = int(cur.get("cpu cores", "1"))
                model = cur.get("model name", model); cur = {}
        if cur:
            phys.add(cur.get("physical id", "0")); cores[cur.get("physical id", "0")] = int(cur.get("cpu cores", "1"))
            model = cur.get("model name", model)
        info.update({"model": model, "sockets": len(phys), "cores_per_socket": max(cores.values()) if cores else None})
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    info["cgroup_cpu_quota"] = quota
    try:
        info["sched_affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["sched_affinity"] = None
    t = info["cores_per_socket"] or info["nproc"] or 1
    if quota:
        t = min(t, max(1, int(quota)))
    if info["sched_affinity"]:
        t = min(t, info["sched_affinity"])
    info["threads_used"] = int(max(1, t))
    return info


class PerModelOps:
    """HipOps with the Step-2 raster loop issued member by member (same launches, same order as
    mhs_ensemble_predict_dev) so that each member's kernel gets its own HIP-event timing."""

    def __init__(self, ops):
        self.o = ops
        self.device = ops.device

    def __getattr__(self, name):
        return getattr(self.o, name)

    def ensemble_band(self, r0, r1, out):
        o = self.o
        g = o.stack.geom
        from machisplin_amd.models import members_predict
        kinds = {"Gbm": "gbm", "Gam": "lm", "Nnet": "nnet", "Earth": "earth", "RandomForest": "rf", "Ksvm": "svr"}
        small = ("lm", "nnet", "earth")
        k, n = 0, len(o.models)
        # mhs_fit_reserve_cus as ONE ensemble call applies it: only the forest (else ksvm, else gbm) is masked
        names = [kinds[type(m).__name__] for m in o.models]
        masked = next((w for w in ("rf", "svr", "gbm") if w in names), None)
        reserve_cus = o.reserve(0)
        while k < n:    # the library's own grouping: a run of gam / nnet / earth members is ONE fused launch
            e = k + 1
            name = kinds[type(o.models[k]).__name__]
            if name in small:
                want = list(small[small.index(name) + 1:])
                while e < n and kinds[type(o.models[e]).__name__] in want:
                    want = want[want.index(kinds[type(o.models[e]).__name__]) + 1:]
                    e += 1
            key = "model_%s_ms" % ("+".join(kinds[type(m).__name__] for m in o.models[k:e]))
            o.timings.setdefault(key, [])
            ms, ws, first = o.models[k:e], o.weights[k:e], k == 0
            o.reserve(reserve_cus if name == masked else 0)
            o._timed(key, lambda: members_predict(o.stack, ms, ws, window=(r0, r1, 0, g.ncol), accumulate=not first, out=out))
            k = e
        o.reserve(reserve_cus)
        st = o.torch.cuda.current_stream(o.device).cuda_stream
        o._lib.check(o._lib.lib().mhs_scale_add_dev(out.data_ptr(), o.wt_total, None, out.data_ptr(), out.numel(), st))


class TileWorkload:
    """BASELINE.json configs[3]: one step = every (tile, layer) unit of a machisplin.tiles.* run -- the whole mltps
    Steps 2-5 of the tile for that response layer, smooth members only, reference-tiled Step 3 inside -- plus ONE
    all-gather of the units' final planes and machisplin.tiles.merge of every layer on its owner rank."""

    def __init__(self, cfg, mhs, torch, dist, rank, world):
        from machisplin_amd import sharded, synth
        self.cfg, self.mhs, self.torch, self.rank, self.world = cfg, mhs, torch, rank, world
        side, n, L = cfg["side"], cfg["stations"], cfg["resp_layers"]
        self.geom = g = synth.grid(side, side)
        seed = synth.BASE_SEED + 4
        xy, rows, cols, uv = synth.stations(g, n, seed)
        self.tiles = mhs.tiles.tiles_create(g, xy, out_ncol=cfg["tiles"][1], out_nrow=cfg["tiles"][0], feather_d=cfg["feather_d"])
        nt = len(self.tiles["dat"])
        mine = sorted({t for t in range(nt) for l in range(L) if sharded.unit_owner(t, l, nt, world)[0] == rank})
        # every rank generates only the windows of the tiles it owns (the generator is a function of the absolute
        # cell indices: synth.covariates(window=...)), i.e. it "reads" only its tiles' crops of the rasters
        self._stacks = {}
        for t in mine:
            r0, r1, c0, c1 = (int(v) for v in self.tiles["win"][t])
            planes, nodata = synth.covariates(g, cfg["layers"], seed, dtype="f32", window=(r0, r1, c0, c1))
            self._stacks[t] = mhs.RasterStack(self.tiles["geom"][t], planes, nodata)
        # the station table: 12 BIO-like response layers = the cfg3 response, shifted / rescaled / re-noised per layer
        full_cov = synth.covariates_at(g, cfg["layers"], seed, rows, cols)
        X = np.column_stack([full_cov, xy])
        base = synth.response(X, uv, seed)
        rng = np.random.default_rng(seed + 99)
        resp = np.column_stack([(1.0 + 0.1 * l) * base + 3.0 * np.sin((2 + l) * uv[:, 0]) + 0.5 * rng.standard_normal(n) for l in range(L)])
        self.int_values = np.column_stack([xy, resp])
        _, wts, tot = mhs.models.select_weights([0.22, 0.12, 0.18, 0.41], labels="gnmv")      # smooth.outputs.only (V73:366-392)
        fitted = {}
        self.unit_params = {}
        for t in mine:
            sel = self.tiles["dat"][t]
            fitted[t] = {}
            for l in range(L):
                if sharded.unit_owner(t, l, nt, world)[0] != rank:
                    continue
                params = synth.ensemble_params(X[sel], resp[sel, l], seed + 7 * l + t, which="gnmv")
                fitted[t][l] = {"models": [mhs.models.from_param_dict(p) for p in params], "weights": wts, "wt_total": tot}
                if not self.unit_params:      # the first unit's host-side parameters: unit_profile / cpu_baseline
                    sv = next(p for p in params if p["kind"] == "svr")
                    self.unit_params[(t, l)] = {"params": params, "weights": wts, "wt_total": tot, "svr_shape": tuple(sv["sv"].shape)}
        self.ops = sharded.HipTileOps(g, self.tiles, lambda t: self._stacks[t], self.int_values, fitted,
                                      tile_edge=cfg.get("tile_edge", 1500))
        self.run = sharded.TileShardedMltps(self.ops, dist, rank, world, merge_on="owner")
        self.cells = side * side * L
        self.last = None

    def step(self):
        self.last = self.run.step()

    def collect(self):
        self.torch.cuda.synchronize()

    # ---- outside the timed region: what ONE (tile, layer) unit is made of, its dominant kernel and the CPU figure ----
    def unit_profile(self):
        """Phases of the unit (tile t, layer l) this rank owns first, each one fenced by device synchronisations: the
        Step-2 member launches (the library's own grouping: gam + nnet + earth fused, ksvm), the station residuals,
        Step 3 + 4 as the reference computes them (its own tiles, fits at their GCV lambdas, mosaic, feathering), Step 5."""
        import torch
        from machisplin_amd import mltps as ml, models as mm, tiles as tl
        mhs = self.mhs
        t, l = self.run.my_units()[0]
        ops = self.ops
        stack = ops._stack(t)
        g = stack.geom
        sel = self.tiles["dat"][t]
        f = ops.fitted[t][l]
        cells = g.nrow * g.ncol
        X, rows, cols = ml.station_predictors(stack, self.int_values[sel, :2])
        keep = ops._keep[t] & ~np.isnan(X).any(axis=1)
        X, rows, cols = X[keep], rows[keep], cols[keep]
        y = self.int_values[sel, 2 + l][keep]

        def timed(fn, reps=2):
            best, out = 1e30, None
            for _ in range(reps):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = fn()
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
            return best, out

        out = torch.empty((g.nrow, g.ncol), dtype=torch.float64, device="cuda")
        kinds = [type(m).__name__ for m in f["models"]]
        small = [i for i, k in enumerate(kinds) if k in ("Gam", "Nnet", "Earth")]
        sv = [i for i, k in enumerate(kinds) if k == "Ksvm"]
        rows_tab = []
        if small:
            ms, _ = timed(lambda: mm.members_predict(stack, [f["models"][i] for i in small], [f["weights"][i] for i in small], out=out))
            by = cells * (4.0 * self.cfg["layers"] + 8.0)
            # ten logistic units per cell (nnet) + the hinges + the linear model: compute, not traffic, bounds it -- the same label
            # and instruction count as in the cfg3 line (884 VALU wave-instructions per cell in the SQ_INSTS_VALU pass of
            # profiles/r04_8d_members_pmc_derived.json, five predictors; this workload's units have the same member shapes)
            ipc = 884.0
            rows_tab.append({"kernel": "small_members_kernel (gam+nnet+earth)", "bound": "valu-issue", "launch_ms": ms,
                             "achieved": cells / 64.0 * ipc / ms / 1e6, "peak": VALU_ISSUE_PEAK_G, "unit": "Gwave-instr/s",
                             "frac": cells / 64.0 * ipc / ms / 1e6 / VALU_ISSUE_PEAK_G,
                             "work": "gam + nnet + earth in one pass: %.0f VALU wave-instructions per cell / 64 lanes (profiles/r04_8d_members_pmc_derived.json); "
                                     "read C fp32 planes once + write the fp64 plane once" % ipc,
                             "hbm_view": {"achieved": by / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / ms / 1e6 / HBM_PEAK_GBS}})
        if sv:
            m = f["models"][sv[0]]
            ms, _ = timed(lambda: mm.members_predict(stack, [m], [f["weights"][sv[0]]], accumulate=True, out=out))
            nsv, p = self.unit_params[(t, l)]["svr_shape"]
            fl = cells * nsv * (3.0 * p + 2.0)
            pm = pmc_derived().get("svr", {})
            ipp = pm.get("valu_per_cell_unit", 15.07)
            rows_tab.append({"kernel": "svr_kernel", "bound": "fp64-valu", "launch_ms": ms, "achieved": fl / ms / 1e9, "peak": FP64_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": fl / ms / 1e9 / FP64_PEAK_TFLOPS,
                             "work": "(3p+2) flop per (cell, SV), exp = 1 flop; %d support vectors, %d cells" % (nsv, cells),
                             "issue_view": {"achieved": cells * nsv / 64.0 * ipp / ms / 1e6, "peak": VALU_ISSUE_PEAK_G, "unit": "Gwave-instr/s",
                                            "frac": cells * nsv / 64.0 * ipp / ms / 1e6 / VALU_ISSUE_PEAK_G},
                             "traffic": pm.get("hbm_bytes_per_cell_fetch_x2_plus_write", 0) * cells or None})
        ms_res, res = timed(lambda: ml.ensemble_residuals(f["models"], f["weights"], f["wt_total"], X, y))
        from machisplin_amd.tps import reduction_cache
        with reduction_cache():
            ms_tps, _ = timed(lambda: mhs.tps_residual_surface(g, X[:, -2:], res, cov1_at_stations=X[:, 0], tile_edge=self.cfg.get("tile_edge", 1500)), reps=1)
            ms_tps2, _ = timed(lambda: mhs.tps_residual_surface(g, X[:, -2:], res, cov1_at_stations=X[:, 0], tile_edge=self.cfg.get("tile_edge", 1500)), reps=1)
        nRx, nCx = tl.step3_tile_windows(g, self.cfg.get("tile_edge", 1500))[:2]
        ms_unit, _ = timed(lambda: ops.tile_layer(t, l, out), reps=1)
        return {"unit": [int(t), int(l)], "tile_cells": int(cells), "stations": int(y.size), "kernels": rows_tab,
                "station_residuals_ms": ms_res, "step3_4_ms_first_layer_of_the_tile": ms_tps, "step3_4_ms_later_layers_reductions_cached": ms_tps2,
                "step3_tiles": [int(nRx), int(nCx)], "whole_unit_ms_alone": ms_unit}

    def cpu_baseline(self, ms_per_step):
        """kind 'port': one (tile, layer) unit on the host -- the oracle's C restatement (OpenMP) of the four smooth members
        over a bounded band of the tile's rows, extrapolated to the tile, + the unit's Step-3 tile fits (numpy QR / eigen /
        GCV) in full + the direct-sum evaluation of one row of its Step-3 tiles, extrapolated -- times the number of units."""
        from oracle import cbind, ensemble as oe, tiles as ot, tps as otps
        from machisplin_amd import mltps as ml
        host = host_info()
        threads = host["threads_used"]
        t, l = self.run.my_units()[0]
        stack = self.ops._stack(t)
        g = stack.geom
        sel = self.tiles["dat"][t]
        params, wts, tot = self.unit_params[(t, l)]["params"], self.unit_params[(t, l)]["weights"], self.unit_params[(t, l)]["wt_total"]
        X, rows, cols = ml.station_predictors(stack, self.int_values[sel, :2])
        keep = self.ops._keep[t] & ~np.isnan(X).any(axis=1)
        X, y = X[keep], self.int_values[sel, 2 + l][keep]
        res = None
        for p, w in zip(params, wts):
            rk = (y - cbind.predict(p, X, threads)) * w
            res = rk if res is None else res + rk
        res = res / tot

        def band(nrows):
            cov = stack.planes[:, :nrows].cpu().numpy().astype(np.float64)
            xs, ys = otps.cell_centres(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol, 0, nrows)
            Xg = oe.stack_predictors(cov, (xs, ys))
            t0 = time.perf_counter()
            cbind.ensemble(params, wts, tot, Xg, threads)
            return time.perf_counter() - t0

        probe = max(1, min(g.nrow, 32))
        tp = band(probe)
        nrows = int(min(g.nrow, max(probe, probe * 8.0 / max(tp, 1e-3))))
        t_ens = band(nrows) * (g.nrow / nrows)
        og = ot.Geom(g.xmin, g.ymax, g.xres, g.yres, g.nrow, g.ncol)
        nRx, nCx, fw, kw = ot.step3_windows(og, self.cfg.get("tile_edge", 1500))
        knots = X[:, -2:]
        rS = np.array([og.row_from_y(v) for v in knots[:, 1]]); cS = np.array([og.col_from_x(v) for v in knots[:, 0]])
        t0 = time.perf_counter()
        fits = []
        for h in range(nRx * nCx):
            r0, r1, c0, c1 = fw[h]
            s_ = np.flatnonzero((rS >= r0) & (rS < r1) & (cS >= c0) & (cS < c1))
            fits.append(otps.fit(knots[s_], res[s_]) if s_.size >= 10 else None)
        t_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        for h in range(nCx):
            if fits[h] is None:
                continue
            gf = ot.window_geom(og, fw[h])
            wk = (kw[h][0] - fw[h][0], kw[h][1] - fw[h][0], kw[h][2] - fw[h][2], kw[h][3] - fw[h][2])
            cbind.tps_eval_grid(fits[h], gf.xmin, gf.ymax, gf.xres, gf.yres, *wk, threads=threads)
        t_eval = (time.perf_counter() - t0) * nRx
        n_units = self.run.n_tiles * self.run.n_layers
        t_unit = t_ens + t_fit + t_eval
        return {"value": self.cells / (t_unit * n_units) / 1e6, "unit": "Mcells/s (cells x response layers)", "cores": threads, "kind": "port",
                "sample": f"one (tile, layer) unit of {n_units}: C restatement (OpenMP, {threads} threads) of the gam/nnet/earth/ksvm ensemble on "
                          f"{nrows} of {g.nrow} rows of the tile (extrapolated: {t_ens:.0f} s) + its {nRx * nCx} Step-3 tile fits in full, numpy "
                          f"({t_fit:.1f} s) + direct-sum evaluation of one row of {nCx} Step-3 tiles x {nRx} ({t_eval:.0f} s); mosaic, feathering and "
                          f"tiles.merge not timed; all units assumed alike",
                "host": host, "unit_s": t_unit, "gpu_over_cpu": t_unit * n_units * 1e3 / ms_per_step}


def emit_line(obj):
    """The ONE JSON line, as the last thing on stdout: C libraries loaded into the process (RCCL prints a version banner through
    C stdio when its first communicator is made) are flushed first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001 -- a platform without fflush in the global namespace loses nothing
        pass
    sys.stdout.flush()
    print(json.dumps(obj), flush=True)


def main_in_library(args, require_distinct=False):
    """`--in-library`: the same step driven by ONE host process over N device slots (csrc/multi.hip).  Launched plainly
    (`python bench.py --gpus N --in-library`) or under torch.distributed.run as the driver launches bench.py -- then rank 0
    drives all N devices and the other ranks only stand at the barriers (they never touch a GPU)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        if rank != 0:
            dist.barrier()          # start of the timed region on rank 0
            dist.barrier()          # its end
            dist.destroy_process_group()
            return
    import machisplin_amd as mhs
    from machisplin_amd import multi, synth
    N = args.gpus
    ndev = torch.cuda.device_count()
    if require_distinct and ndev < N:
        sys.stderr.write("bench.py: --gpus %d without a launcher drives %d devices from one process, but this host has %d; "
                         "refusing to report an N-GPU line from fewer devices (use --in-library for the aliased-slot plumbing run)\n"
                         % (N, N, ndev))
        sys.exit(3)
    ids = [k % max(ndev, 1) for k in range(N)]
    torch.cuda.set_device(0)
    multi.init_devices(N, ids)
    cfg = WORKLOADS[args.workload]
    line = {"metric": "grid Mcells/s (ensemble+TPS predict) + TPS-solve GFLOP/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    driver = {"host": "one process, one host thread per device slot (mhs_init_devices)", "device_ids": ids,
              "n_physical_gpus": len(set(ids)), "slots_share_devices": len(set(ids)) < N}
    if cfg.get("tiled"):
        # BASELINE configs[3]: machisplin.tiles.create -> mltps per (tile, layer) -> machisplin.tiles.merge in ONE library call,
        # host planes in, host planes out (what the R shim hands over): PCIe is inside the call
        side, n, L = cfg["side"], cfg["stations"], cfg["resp_layers"]
        g = synth.grid(side, side)
        seed = synth.BASE_SEED + 4
        xy, rows, cols, uv = synth.stations(g, n, seed)
        tl = mhs.tiles.tiles_create(g, xy, out_ncol=cfg["tiles"][1], out_nrow=cfg["tiles"][0], feather_d=cfg["feather_d"])
        planes, nodata = synth.covariates(g, cfg["layers"], seed, dtype="f32")
        host = planes.cpu().numpy()
        del planes
        torch.cuda.empty_cache()
        full_cov = synth.covariates_at(g, cfg["layers"], seed, rows, cols)
        X = np.column_stack([full_cov, xy])
        base = synth.response(X, uv, seed)
        rng = np.random.default_rng(seed + 99)
        resp = np.column_stack([(1.0 + 0.1 * l) * base + 3.0 * np.sin((2 + l) * uv[:, 0]) + 0.5 * rng.standard_normal(n) for l in range(L)])
        _, wts, tot = mhs.models.select_weights([0.22, 0.12, 0.18, 0.41], labels="gnmv")
        nt = len(tl["dat"])
        units = [[None] * nt for _ in range(L)]
        for t in range(nt):
            sel = tl["dat"][t]
            gt = tl["geom"][t]
            tr, tc = mhs.tiles.cells_from_xy(gt, xy[sel])
            Xt = np.column_stack([full_cov[sel], gt.x_from_col(tc), gt.y_from_row(tr)])       # the TILE raster's cell centres (V73:127-154)
            ok = (tr >= 0) & ~np.isnan(Xt).any(axis=1) & ~np.isnan(resp[sel]).any(axis=1)
            for l in range(L):
                params = synth.ensemble_params(X[sel], resp[sel, l], seed + 7 * l + t, which="gnmv")
                units[l][t] = {"models": [mhs.models.from_param_dict(p) for p in params], "weights": wts, "wt_total": tot,
                               "X": Xt[ok], "resp": resp[sel, l][ok]}
        kw = dict(tile_edge=cfg.get("tile_edge", 1500))
        merged = [np.zeros((side, side)) for _ in range(L)]      # the caller's planes, reused: 9.6 GB of fresh numpy arrays per call cost
                                                                 # ~0.4 s to map, fault in and unmap again
        run = lambda: multi.tiles_units_multi(g, host, nodata, cfg["tiles"][1], cfg["tiles"][0], cfg["feather_d"], units, L, out=merged, **kw)
        cells = side * side * L
        line["unit"] = "Mcells/s (cells x response layers)"
        config = {"workload": cfg["name"], "stations": n, "grid": [side, side], "response_layers": L, "user_tiles": list(cfg["tiles"]),
                  "parallelism": "in-library: (tile, layer) units round-robin over %d device slot(s), a layer's tiles to its owner over xGMI, "
                                 "tiles.merge there (mhs_tiles_units_multi)" % N,
                  "boundary": "host planes in, merged host planes out: PCIe inside the timed region (float32 covariates up, 12 float64 planes down)"}
    else:
        wl = Workload(cfg, mhs, torch, None, 0, 1, args.tps_mode)
        host = wl.stack.planes.cpu().numpy()
        g, cells = wl.geom, wl.cells
        tile_edge = 1500 if args.tps_mode == "tiled" else None
        # the bands' balance (setup, untimed): one step with equal bands tells what share slot 0 -- which also fits -- should take
        ms = multi.MultiStack(g, host, wl.stack.nodata)
        info = ms.step(wl.models, wl.weights, wl.wt_total, wl.X, wl.resp, tile_edge=tile_edge, gather=True)
        share = info["suggested_slot0_share"]
        if N > 1 and share == share:
            ms.free()
            ms = multi.MultiStack(g, host, wl.stack.nodata, slot0_share=share)
        run = lambda: ms.step(wl.models, wl.weights, wl.wt_total, wl.X, wl.resp, tile_edge=tile_edge, gather=True)
        line["unit"] = "Mcells/s"
        config = {"workload": cfg["name"], "stations": cfg["stations"], "grid": [cfg["side"], cfg["side"]],
                  "covariates": "%d x float32 planes, band k resident in slot k's HBM" % cfg["layers"],
                  "members": [p["kind"] for p in wl.params], "gbm_trees": cfg["gbm_trees"], "rf_trees": cfg["rf_trees"],
                  "tps_mode": args.tps_mode, "slot0_row_share": None if share != share else share,
                  "parallelism": "in-library: rowband%d over device slots, coefficients handed over in host memory, 1 all-gather "
                                 "(mhs_mltps_grid_multi_dev, gather = 1)" % N}
    for _ in range(args.warmup):
        last = run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    line.update({"value": cells * args.steps / dt / 1e6, "ms_per_step": dt / args.steps * 1e3, "config": config})
    if cfg.get("tiled"):
        outs, rsq, uinfo = last
        driver.update(uinfo)
        line.update({"rsq_model_mean": float(np.nanmean(rsq[:, :, 0])), "rsq_final_mean": float(np.nanmean(rsq[:, :, 1])),
                     "roofline": None, "cpu_baseline": None})
    else:
        driver.update({k: last[k] for k in ("bands", "band_ms", "tiles_ms", "tiles_owned", "tiles_pulled_bytes", "fit_ms", "step_ms", "collective",
                                             "suggested_slot0_share")})
        driver["collective_ranks"] = N
        try:
            driver["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            driver["rccl_version"] = None
        if N > 1 and len(set(ids)) == N and last["collective"] != "rccl-all-gather":
            # N distinct devices must stitch the plane with the ONE RCCL all-gather the north star names: a silent fall-back to
            # peer copies would measure another collective
            sys.stderr.write("bench.py: %d distinct devices but the step's collective was '%s', not the RCCL all-gather\n" % (N, last["collective"]))
            sys.exit(4)
        line.update({"lambda": last["lambda"], "rsq_model": last["rsq_model"], "rsq_final": last["rsq_final"]})
        # the dominant kernel's roofline: the slots launch the kernels of the one-device step on their bands; the member-by-member
        # HIP-event table is taken from that step on slot 0's device over the whole grid (outside the timed region)
        for _ in range(2):
            wl.step()
        wl.collect()
        wl.measure_mean_visits()
        table = wl.kernel_table()
        dom = max(table, key=lambda r: r["launch_ms"]) if table else None
        line["roofline"] = ({k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "work", "pmc",
                                                  "issue_view") if k in dom} if dom else None)
        if line["roofline"] is not None:
            line["roofline"]["measured_on"] = "one device, whole grid: the same kernel every slot launches on its band"
        line["cpu_baseline"] = None if (args.no_cpu_baseline or N > 1) else wl.cpu_baseline(line["ms_per_step"])
        # the in-library planes against the one-process-per-GPU driver's arithmetic on one device (bit for bit)
        one = wl.last["final"].cpu().numpy()
        line["equals_one_device_plane"] = bool(np.array_equal(ms.download(), one, equal_nan=True))
        ms.free()
        # what the R shim's ONE call costs with PCIe inside it (mhs_mltps_grid_multi: host planes in, host plane out): the band
        # of every slot travels in sub-bands under its own first kernels; never `value`
        hc = []
        got = np.empty((g.nrow, g.ncol))          # the caller's plane, reused: a fresh 800 MB numpy array per call costs ~30 ms to unmap
        for rep in range(4):
            got.fill(0.0)
            t0 = time.perf_counter()
            _, hinfo = multi.mltps_grid_multi(g, host, wl.stack.nodata, wl.models, wl.weights, wl.wt_total, wl.X, wl.resp,
                                              tile_edge=tile_edge, slot0_share=None if share != share else share, out=got)
            hc.append((time.perf_counter() - t0) * 1e3)
        driver["host_call"] = {"ms": min(hc), "calls_ms": hc, "copies_issued_ms": hinfo["upload_ms"], "download_ms": hinfo["download_ms"],
                               "step_ms": hinfo["step_ms"], "planes": "%d x %s up, 1 x float64 down" % (cfg["layers"], host.dtype),
                               "equals_one_device_plane": bool(np.array_equal(got, one, equal_nan=True))}
    line["in_library"] = driver
    emit_line(line)
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.in_library:
        return main_in_library(args)
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # `python bench.py --gpus N` with no launcher: the N devices are driven from THIS process through the library's own
        # multi-device path (mhs_init_devices: the path the single-threaded R host takes, one RCCL all-gather over xGMI) --
        # never a one-GPU line under an N-GPU flag.  N distinct devices are required; anything else exits non-zero.
        return main_in_library(args, require_distinct=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MHS_BENCH_ONE_GPU"):   # plumbing tests: every rank on GPU 0
        local = 0
    import torch
    import torch.distributed as dist
    import machisplin_amd as mhs

    if local >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d wants GPU %d but this node shows %d device(s); one rank per GPU is the contract\n"
                         % (rank, local, torch.cuda.device_count()))
        sys.exit(3)
    torch.cuda.set_device(local)
    mhs.init(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MHS_BENCH_BACKEND", "nccl")   # "gloo" lets two ranks share one GPU (plumbing tests only)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE is %d\n" % (args.gpus, world))
        sys.exit(2)

    cfg = WORKLOADS[args.workload]
    wl = TileWorkload(cfg, mhs, torch, dist, rank, world) if cfg.get("tiled") else Workload(cfg, mhs, torch, dist, rank, world, args.tps_mode)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    fence()
    wl.collect()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    fence()
    dt = time.perf_counter() - t0
    wl.collect()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if cfg.get("tiled"):
        if rank == 0:
            out = wl.last
            unit = wl.ops.unit_ms
            res = {
                "metric": "grid Mcells/s (ensemble+TPS predict) + TPS-solve GFLOP/s",
                "value": wl.cells * args.steps / dt / 1e6, "unit": "Mcells/s (cells x response layers)",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": cfg["name"], "stations": cfg["stations"], "grid": [cfg["side"], cfg["side"]],
                           "response_layers": cfg["resp_layers"], "members": ["lm", "nnet", "earth", "svr"],
                           "user_tiles": list(cfg["tiles"]), "feather_d": cfg["feather_d"],
                           "tile_shapes": [list(sh) for sh in wl.run.shapes],
                           "stations_per_tile": [int(len(d)) for d in wl.tiles["dat"]],
                           "tps_mode": "reference-tiled inside every user tile (tile edge %d, V73:656-753), GCV lambda" % cfg.get("tile_edge", 1500),
                           "parallelism": "(tile, layer) units round-robin over %d rank(s) + 1 all-gather + tiles.merge on the layer's owner" % world},
                "units_on_rank0": len(unit), "unit_ms_mean_rank0": float(np.mean(list(unit.values()))) if unit else None,
                "unit_ms_max_rank0": float(np.max(list(unit.values()))) if unit else None,
                "rsq_model_mean": float(np.nanmean(out["rsq_model"])), "rsq_final_mean": float(np.nanmean(out["rsq_final"])),
                "roofline": None, "cpu_baseline": None,
            }
            prof = wl.unit_profile()
            dom = max(prof["kernels"], key=lambda r: r["launch_ms"]) if prof["kernels"] else None
            res["unit_profile"] = prof
            res["roofline"] = ({k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "work", "issue_view")
                                if k in dom} if dom else None)
            if res["roofline"] is not None:
                res["roofline"].setdefault("traffic", None)
            if not args.no_cpu_baseline and world == 1:
                res["cpu_baseline"] = wl.cpu_baseline(res["ms_per_step"])
            emit_line(res)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    phase_table = wl.phase_table() if world > 1 else None     # collective: outside the rank-0 block
    # N > 1, global mode: the same Steps 2-5 with Step 3 the way the reference computes it above 1 500 px (tiles dealt over the
    # ranks, no serial fit) for the record, so that one SCALE run shows both modes (collective; outside the timed region)
    tiled_mode = None
    if world > 1 and args.tps_mode == "global" and cfg["ensemble"] and not cfg.get("tiled") and not os.environ.get("MHS_BENCH_SKIP_TILED"):
        from machisplin_amd import sharded
        run2 = sharded.TiledTpsShardedMltps(PerModelOps(wl.ops), dist, rank, world, cfg["side"], cfg["side"], tile_edge=1500)
        run2.step()
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            run2.step()
        fence()
        t2 = torch.tensor([(time.perf_counter() - t1) / 3.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        tiled_mode = {"tps_mode": "reference-tiled Step 3 (V73:636-897), tiles dealt over the ranks, one all-gather", "steps": 3,
                      "ms_per_step": float(t2.item()) * 1e3, "mcells_per_s": wl.cells / float(t2.item()) / 1e6}
        del run2
    if rank == 0:
        wl.measure_mean_visits()
        table = wl.kernel_table()
        dom = max(table, key=lambda r: r["launch_ms"]) if table else None
        tm = wl.ops.timings
        fit_overlapped_ms = float(np.mean(tm["tps_fit_ms"][-max(1, len(tm["tps_fit_ms"]) // 2):])) if tm["tps_fit_ms"] else None
        # inside a step the fit runs BESIDE the ensemble kernels (it is starved by them and its wall time
        # is not a kernel property), so the TPS-solve rate is taken from a stand-alone fit after the run
        # for the record (outside the timed region), FIRST among the records -- the later ones (direct-sum spline evaluation,
        # stand-alone fits) leave the device clocked lower for a while, which a profiled run showed as +25 % on every member
        # kernel of this leg: the boundary as the R shim reaches it -- float64 planes (terra holds doubles in RAM,
        # integration/r/src/machisplin_shim.c passes MHS_F64) resident in HBM, and the host-pointer entry point
        # mhs_ensemble_predict exactly as mhsr_ensemble_predict calls it (PCIe included)
        # (round 6: these two stand-alone records come FIRST, right behind the timed steps -- the legs below leave the device
        # clocked lower for a while, which is how one call read 41 ms in one line and 105 ms in another in round 5)
        knots, resid = wl.ops.X[:, -2:], wl.run.ops.station_residuals()[1]
        fit_ms = 1e30
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            mhs.Tps(knots, resid)
            fit_ms = min(fit_ms, (time.perf_counter() - t1) * 1e3)
        # for the record (outside the timed region): the same residual surface the way the REFERENCE computes it at
        # this size -- ceil(n/1500)^2 overlapping tiles with their own fits, mean mosaic, seam feathering (V73:656-895)
        tiled_ms = 1e30
        for _ in range(3):       # the first call grows the library's arenas (round-4 verdict: one cold call was reported)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            mhs.tps_residual_surface(wl.geom, knots, resid, cov1_at_stations=wl.ops.X[:, 0], tile_edge=1500)
            torch.cuda.synchronize()
            tiled_ms = min(tiled_ms, (time.perf_counter() - t1) * 1e3)
        nRx, nCx = mhs.tiles.step3_tile_windows(wl.geom, 1500)[:2]
        info = {"nRx": int(nRx), "nCx": int(nCx)}
        f64_boundary = None
        if world == 1 and cfg["ensemble"] and wl.cells <= 2 * 10 ** 8 and not os.environ.get("MHS_BENCH_SKIP_F64"):
            f64_boundary = wl.f64_boundary()
        raster_sensitivity = None
        if world == 1 and cfg["ensemble"] and not os.environ.get("MHS_BENCH_SKIP_SENSITIVITY"):
            reserved = wl.ops.reserve(0)
            try:
                raster_sensitivity = wl.raster_sensitivity()
            finally:
                if reserved:
                    wl.ops.reserve(reserved)
        # for the record (outside the timed region): the spline of the last step evaluated by the direct sum
        # (predict.Krig's own loop) next to the far-field-interpolated sum the timed steps use
        eval_check = None
        if world == 1 and wl.ops.last_fit is not None:
            fit = wl.ops.last_fit
            far = mhs.interpolate(wl.geom, fit)
            plan = fit.eval_plan()
            mhs.eval_mode(mhs.EVAL_DIRECT)
            try:
                direct = torch.empty_like(far)
                mhs.interpolate(wl.geom, fit, out=direct)           # warm-up
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                mhs.interpolate(wl.geom, fit, out=direct)
                torch.cuda.synchronize()
                direct_ms = (time.perf_counter() - t1) * 1e3
            finally:
                mhs.eval_mode(mhs.EVAL_AUTO)
            # the yardstick for both sums' rounding is S = sum_j |c_j phi_j| (the terms cancel by orders of magnitude in
            # a fitted spline); S on 2 000 sampled cells, on the host
            rng = np.random.default_rng(0)
            rr, cc = rng.integers(0, wl.geom.nrow, 2000), rng.integers(0, wl.geom.ncol, 2000)
            u = (wl.geom.x_from_col(cc) - fit.center[0]) / fit.scale[0]
            v = (wl.geom.y_from_row(rr) - fit.center[1]) / fit.scale[1]
            S = np.zeros(2000)
            for j0 in range(0, fit.n, 500):
                d2 = (u[:, None] - fit.knots[j0:j0 + 500, 0]) ** 2 + (v[:, None] - fit.knots[j0:j0 + 500, 1]) ** 2
                with np.errstate(divide="ignore", invalid="ignore"):
                    S += (np.abs(fit.c[j0:j0 + 500]) * np.abs(np.where(d2 > 0, d2 * np.log(d2), 0.0))).sum(axis=1) * (0.5 / (8 * np.pi))
            diff = float((far - direct).abs().max())
            eval_check = {"timed_path": "far-field-interpolated, tiles %d x %d" % plan[:2] if plan[0] else "direct sum",
                          "direct_sum_ms": direct_ms,
                          "max_abs_diff_over_max_abs": diff / float(direct.abs().max()),
                          "max_abs_diff_over_sum_abs_terms": diff / float(S.max())}
            del far, direct
        # N > 1: what the step should cost from its parts (this rank's band, the stand-alone fit, the gather) against
        # what it did cost -- so that an 8-GPU line can be read without a profiler
        model_check = None
        if world > 1:
            model_check = wl.model_check(phase_table, fit_ms, dt / args.steps * 1e3)
        # N = 1: what the row-band driver should do on 2 / 4 / 8 GPUs, from this run's parts and model_check's own formula --
        # a prediction for the first real SCALE run to be held against (8-GPU boxes are the driver's, not the builder's)
        projected = None
        if world == 1 and table and args.tps_mode == "tiled":
            # Reference-tiled Step 3 (V73:656-747) has NO serial fit: every rank predicts its band and fits + evaluates its
            # share of the tiles (LPT-dealt on stations x cells), ONE all-gather moves bands and tile planes, every rank mosaics.
            # Model: step(N) = max(ens / N, tiles(N)) + gather(N) + mosaic + Step 5, with
            #   tiles(N)  = the stand-alone time of all tiles x the heaviest rank's cost share (round 6: a rank's tiles are ONE fit
            #               launch + one pair of evaluation launches, a workgroup per spline -- no 8-at-a-time floor any more);
            #   gather(N) = one rank's chunk (its band + its tile planes, 8 B per cell) over one 153 GB/s xGMI link per peer;
            #   mosaic + Step 5 measured here, stand-alone.
            from machisplin_amd import sharded
            run = wl.run
            ens = sum(r["launch_ms"] for r in table if not r["kernel"].startswith("tps_"))
            step1 = dt / args.steps * 1e3
            costs = [float(c) for c in run.layout["cost"]]
            tiles_view = [run._tile_view(run.full, run.owner[h] * run.chunk, h) for h in range(len(run.keep))]
            mosaic_ms = 1e30
            for _ in range(3):      # best of three: the first call of this path loads its kernels
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                run.ops.tps_mosaic(run.layout["nRx"], run.layout["nCx"], run.keep, tiles_view, run.total)
                torch.cuda.synchronize()
                mosaic_ms = min(mosaic_ms, (time.perf_counter() - t1) * 1e3)
            tiles_alone = max(tiled_ms - mosaic_ms, 0.5 * tiled_ms)      # (the stand-alone mosaic also looks for the seams' boxes: never more than half)
            tiles_in_step = float(np.mean(tm["tps_tiles_ms"][-max(1, len(tm["tps_tiles_ms"]) // 2):])) if tm.get("tps_tiles_ms") else None
            tail1 = max(0.0, step1 - max(ens, tiles_in_step or 0.0))      # mosaic, the band sums, Step 5 as this run had them
            projected = {"model": "step(N) = max(ens / N, tiles(N)) + gather(N) + tail; tiles(N) = stand-alone tile time x the heaviest rank's LPT cost share; "
                                  "gather(N) = (band + tile planes of one rank) x 8 B over one 153 GB/s xGMI link per peer; tail = this run's step "
                                  "minus max(ens, tiles in the step) = mosaic + feathering + band sums + Step 5.  No serial fit in this mode.",
                         "ensemble_ms_n1": ens, "tiles_ms_alone": tiles_alone, "tiles_ms_in_step_n1": tiles_in_step, "mosaic_ms": mosaic_ms,
                         "tail_ms": tail1, "step_ms_n1": step1, "n_tiles": len(costs)}
            tile_cells = float(sum(run.cells))
            for N in (2, 4, 8):
                owner = sharded.assign_tiles(costs, N)
                load = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(N)]
                share = max(load) / sum(costs)
                tiles_n = tiles_alone * share
                gather = (wl.cells + tile_cells) * 8.0 / N / 153e9 * 1e3
                step = max(ens / N, tiles_n) + gather + tail1
                projected["n%d" % N] = {"band_ms": ens / N, "tiles_ms": tiles_n, "heaviest_rank_cost_share": max(load) / sum(costs), "gather_ms": gather,
                                        "step_ms": step, "mcells_per_s": wl.cells / step / 1e3, "speedup_over_n1": step1 / step}
        elif world == 1 and table:
            ens = sum(r["launch_ms"] for r in table if not r["kernel"].startswith("tps_"))
            spl = sum(r["launch_ms"] for r in table if r["kernel"].startswith("tps_"))
            step1 = dt / args.steps * 1e3
            projected = {"model": "rank 0: band x + the serial fit; ranks 1..N-1: band (ens - x) / (N - 1); x = max(0, (ens - (N - 1) fit) / N); then every "
                                  "rank evaluates the spline on ITS rows (spline / N; round 6 -- the whole grid on every rank before), sums, and ONE "
                                  "all-gather stitches the selected plane (a band's bytes from each peer over its own 153 GB/s xGMI link), plus Step 5 "
                                  "and the small all-reduce (~3 ms); fit unconfined (no CU reservation at N > 1)",
                         "ensemble_ms_n1": ens, "fit_ms": fit_ms, "spline_eval_ms": spl, "step_ms_n1": step1}
            for N in (2, 4, 8):
                x = max(0.0, (ens - (N - 1) * fit_ms) / N)
                others = (ens - x) / (N - 1)
                band_bytes = wl.cells * 8.0 * (others / ens)
                gather = band_bytes / 153e9 * 1e3
                step = max(x + fit_ms, others) + spl / N + gather + 3.0
                projected["n%d" % N] = {"rank0_band_ms": x, "other_band_ms": others, "spline_band_ms": spl / N, "gather_ms": gather, "step_ms": step,
                                        "mcells_per_s": wl.cells / step / 1e3, "speedup_over_n1": step1 / step}
        m = wl.ops.X.shape[0] - 3
        if cfg["ensemble"] and wl.cfg["stations"] >= 2000:
            # the synthetic members are fitted, not random: the ensemble explains the response and the spline improves on it
            assert 0.5 < wl.last["rsq_model"] < wl.last["rsq_final"], (wl.last["rsq_model"], wl.last["rsq_final"])
        res = {
            "metric": "grid Mcells/s (ensemble+TPS predict) + TPS-solve GFLOP/s",
            "value": wl.cells * args.steps / dt / 1e6,
            "unit": "Mcells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "stations": cfg["stations"], "grid": [cfg["side"], cfg["side"]],
                       "covariates": "%d x float32 planes resident in HBM" % cfg["layers"],
                       "members": [p["kind"] for p in wl.params], "gbm_trees": cfg["gbm_trees"], "rf_trees": cfg["rf_trees"],
                       "tps_mode": ("global (one fit on all stations, GCV lambda, V73:748-753)" if args.tps_mode == "global" else
                                    "reference-tiled (ceil(n/1500)^2 overlapping tiles with their own GCV fits, mosaic, feathering, V73:636-897), tiles dealt over the ranks"),
                       "parallelism": ("rowband%d + bcast(coef) + 1 all-gather" if args.tps_mode == "global" else
                                       "rowband%d + Step-3 tiles dealt over the ranks + 1 all-gather") % world,
                       "rank0_row_share": wl.rank0_share, "fit_reservation": wl.reservation},
            "roofline": ({k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "work", "pmc",
                                               "ops_view", "issue_view") if k in dom} if dom else None),  # None only if rank 0 was given no rows at all
            "kernels": table,
            "tps_fit_ms": fit_ms, "tps_fit_ms_overlapped_with_ensemble": fit_overlapped_ms,
            "tps_solve_gflops": 4.0 * m ** 3 / 3.0 / (fit_ms * 1e-3) / 1e9,
            "tps_solve_flop_model": "4/3 (n-3)^3: Householder reduction of Q2'KQ2 to band form (GCV path), whole mhs_tps_fit call",
            "reference_tiled_tps_ms": tiled_ms, "reference_tiled_tps_tiles": [info.get("nRx"), info.get("nCx")],
            "reference_tiled_tps_ms_how": "best of 3 warmed calls of mhs_tps_surface_dev, taken right behind the timed steps (before the float64 / "
                                          "sensitivity legs); Steps 3 + 4: station selection, every tile's fit in one launch, batched evaluation, "
                                          "fused mosaic + feather",
            "tps_eval_check": eval_check,
            "forest_traffic": (forest_traffic(int(next(p["tree_offsets"][-1] for p in wl.params if p["kind"] == "rf")), 1e8, cfg["layers"])
                               if any(p["kind"] == "rf" for p in wl.params) else None),
            "f64_boundary": f64_boundary, "raster_sensitivity": raster_sensitivity, "model_check": model_check, "projected": projected,
            "tiled_mode_same_run": tiled_mode,
            "lambda": wl.last.get("lambda"), "rsq_model": wl.last["rsq_model"], "rsq_final": wl.last["rsq_final"],
        }
        if not args.no_cpu_baseline and world == 1:   # timed on rank 0 at N = 1 only
            ff = next((r["launch_ms"] for r in table if r["kernel"].startswith("tps_ff")), None)
            res["cpu_baseline"] = wl.cpu_baseline(res["ms_per_step"], eval_check["direct_sum_ms"] if eval_check else None, ff)
        emit_line(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
