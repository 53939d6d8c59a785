// Step 3 + Step 4 of machisplin.mltps in ONE call (V73:636-897): tile grid, per-tile
// fields::Tps on the stations of the fit box, evaluation on the keep window, mean mosaic,
// seam feathering, overlay.  This is the entry point a .Call() shim binds so that the R
// side replaces the whole block by a single call; the Python mirror composes the same
// steps from the finer-grained entry points (machisplin_amd/mltps.py).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "common.h"
#include "tps_batch.h"

using namespace mhs;

constexpr int64_t TILE_LANES = 8;   // tiles fitted side by side (mhs_tps_surface)

// Fit and evaluate a set of Step-3 tiles: job k = tile tile_ids[k] (or tile k), its keep-window plane (rows x cols of
// the window, contiguous) written to out_ptrs[k].  Returns after everything has finished.
//
// Round 6: every tile with 8..256 distinct stations -- all of them at the reference's tile size (105-250
// per tile) -- goes through ONE fit launch (tps_batch.hip: a workgroup per spline) and ONE pair of evaluation launches
// (tps_eval.hip: a window per spline, knots and polynomial part read where the fit left them): three kernels and two
// copies for the whole set, where rounds 1-5 drove a chain of ~10 launches and 4 host round trips per tile from eight
// host threads (41 ms for cfg3's 49 tiles; the launches were the time).  Tiles the batch cannot hold (more than 256
// stations: tile_edge far above the reference's 1500) still take that route, on the lanes, beside the batch.
// MHS_TILES_BATCH=0 sends every tile down the lanes (the parity tests compare the two).
static int run_tiles(const mhs_grid *g, const double *xy, const double *resid, int64_t n, const double *cov1_at_stations,
                     const std::vector<int64_t> &fit, const std::vector<int64_t> &keep, const std::vector<int64_t> &rows,
                     const std::vector<int64_t> &cols, double lambda, int gcv_mode, const int64_t *tile_ids, int64_t njobs,
                     double *const *out_ptrs) {
    if (njobs <= 0) return MHS_OK;
    const bool timing = getenv("MHS_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[run_tiles] %-36s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const char *benv = getenv("MHS_TILES_BATCH");
    const bool use_batch = !(benv && benv[0] == '0');
    // ---- the tiles' stations (terra::extract(rb[[1]], Full.cords) + complete.cases, V73:701-706)
    struct Job {
        int64_t h = 0, m = 0;
        std::vector<double> txy, sr;
        int route = 0;      // 0 zeros, 1 batch, 2 lane
        int bjob = -1, rc = MHS_OK;
        std::string err;
        EvalPlanHandle *plan = nullptr;
        std::vector<int> perm;
    };
    std::vector<Job> jobs((size_t)njobs);
    std::lock_guard<std::mutex> batch_lock(batch_mutex());
    FitLane *Lb = nullptr;
    if (int rc = batch_lane(&Lb)) return rc;
    const hipStream_t sb = Lb->s;
    SmallBatch B;
    EvalBatch *EB = eval_batch_create();
    struct EbGuard { EvalBatch *p; ~EbGuard() { eval_batch_destroy(p); } } eb_guard{EB};
    std::vector<TpsPrep> preps((size_t)njobs);
    std::vector<int64_t> lane_jobs;
    // Everything the host does per tile before the launches -- its stations (an O(n) scan), Krig's replicate collapse, the
    // QR of [1 u v], the evaluation plan's counting sort -- is independent of the other tiles: a few host threads share
    // the tiles (2.1 ms on one thread for cfg3's 49 tiles, as long as the kernels they feed).
    auto prepare_one = [&](int64_t job) {
        Job &J = jobs[(size_t)job];
        J.h = tile_ids ? tile_ids[job] : job;
        const int64_t *f = &fit[(size_t)J.h * 4], *k = &keep[(size_t)J.h * 4];
        const int64_t kc = k[3] - k[2];
        std::vector<double> sx, sy;
        for (int64_t i = 0; i < n; ++i) {
            if (rows[i] < f[0] || rows[i] >= f[1] || cols[i] < f[2] || cols[i] >= f[3]) continue;
            if (cov1_at_stations && std::isnan(cov1_at_stations[i])) continue;
            if (std::isnan(resid[i])) continue;
            sx.push_back(xy[i]); sy.push_back(xy[n + i]); J.sr.push_back(resid[i]);
        }
        J.m = (int64_t)J.sr.size();
        if (J.m < 10) { J.route = 0; return; }      // V73:710-721: the tile is all zeros
        J.txy.resize((size_t)2 * J.m);
        for (int64_t i = 0; i < J.m; ++i) { J.txy[(size_t)i] = sx[(size_t)i]; J.txy[(size_t)(J.m + i)] = sy[(size_t)i]; }
        J.route = 2;
        if (use_batch && J.m <= 4 * SB_NMAX) {      // (replicates can only shrink the count)
            TpsPrep &P = preps[(size_t)job];
            J.rc = tps_prepare(J.txy.data(), J.sr.data(), J.m, P);
            if (J.rc) { J.err = mhs_last_error(); return; }
            if (P.n >= SB_NMIN && P.n <= SB_NMAX) {
                // terra::interpolate(terra::rast(rb), tps): cell centres of the FIT raster (V73:726), the keep window of it
                mhs_grid gf = *g;
                gf.xmin = g->xmin + (double)f[2] * g->xres;
                gf.ymax = g->ymax - (double)f[0] * g->yres;
                gf.nrow = f[1] - f[0]; gf.ncol = f[3] - f[2];
                J.plan = eval_batch_plan(P.uv.data(), (int)P.n, P.center, P.scale, &gf, k[0] - f[0], k[1] - f[0], k[2] - f[2],
                                         k[3] - f[2], out_ptrs[job], kc, J.perm, &J.rc);
                if (J.rc) { J.err = mhs_last_error(); return; }
                J.route = 1;
            }
        }
    };
    {
        const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(njobs / 8, 6), cpu_budget()));
        std::atomic<int64_t> next{0};
        auto work = [&]() { for (;;) { const int64_t job = next.fetch_add(1); if (job >= njobs) break; prepare_one(job); } };
        std::vector<std::thread> threads;
        for (int q = 1; q < nthreads; ++q) threads.emplace_back(work);
        work();
        for (std::thread &th : threads) th.join();
    }
    for (int64_t job = 0; job < njobs; ++job) {      // in job order: errors, the zero tiles, the batch's entries
        Job &J = jobs[(size_t)job];
        if (J.rc) {
            for (Job &Q : jobs) if (Q.plan) { eval_plan_drop(Q.plan); Q.plan = nullptr; }
            set_error("%s", J.err.c_str());
            return J.rc;
        }
        const int64_t *k = &keep[(size_t)J.h * 4];
        if (J.route == 0) MHS_HIP(hipMemsetAsync(out_ptrs[job], 0, sizeof(double) * (size_t)((k[1] - k[0]) * (k[3] - k[2])), sb));
        else if (J.route == 1) {
            eval_batch_commit(EB, J.plan, B.count, B.knot_total);
            J.plan = nullptr;
            J.bjob = small_batch_add(B, preps[(size_t)job], lambda, gcv_mode, J.perm.data());
        } else lane_jobs.push_back(job);
    }
    lap("stations, preparation, plans (host)");
    // ---- the batch: fit launch, then the two evaluation launches behind it on the same stream
    char *extra = nullptr;
    if (int rc = small_batch_launch(B, *Lb, sb, eval_batch_device_bytes(EB), &extra)) return rc;
    if (B.count > 0)
        if (int rc = eval_batch_launch(EB, extra, B.knots_dev, B.res_dev, sb)) return rc;
    lap("uploads + launches");
    // ---- the other tiles: chains of small, latency-bound kernels, several side by side, each on its own lane (two streams +
    // work arena) driven by its own host thread
    int rc = MHS_OK;
    std::string err_msg;
    std::vector<mhs_tps *> handles(lane_jobs.size(), nullptr);
    std::vector<FitLane *> lanes;
    if (!lane_jobs.empty()) {
        const int nlanes = (int)std::min<int64_t>((int64_t)lane_jobs.size(), TILE_LANES);
        lanes.resize((size_t)nlanes);
        for (int l = 0; l < nlanes; ++l)
            if (int rc2 = fit_lane(1 + l, &lanes[(size_t)l])) return rc2;
        std::atomic<int64_t> next{0};
        std::atomic<int> first_rc{MHS_OK};
        std::mutex err_mu;
        const int slot = current_slot();
        auto worker = [&](int lane_id) {
            FitLane &L = *lanes[(size_t)lane_id];
            SlotBind bind(slot);                // the slot and HIP's current device are per host thread
            // mhs_fit_reserve_cus active: the tiles' evaluations stay, like their fits, on the reserved compute units
            const hipStream_t ls = (ctx().reserved_cus > 0 && L.ms) ? L.ms : L.s;
            for (;;) {
                const int64_t q = next.fetch_add(1);
                if (q >= (int64_t)lane_jobs.size() || first_rc.load() != MHS_OK) break;
                const int64_t job = lane_jobs[(size_t)q];
                Job &J = jobs[(size_t)job];
                const int64_t *f = &fit[(size_t)J.h * 4], *k = &keep[(size_t)J.h * 4];
                const int64_t kc = k[3] - k[2];
                mhs_tps *t = nullptr;
                int rc2 = tps_fit_lane(L, J.txy.data(), J.sr.data(), J.m, lambda, gcv_mode, J.m < 1500 ? 2 : 0, &t);
                if (!rc2) {
                    handles[(size_t)q] = t;   // freed after the last tile: hipFree synchronises the device
                    mhs_grid gf = *g;
                    gf.xmin = g->xmin + (double)f[2] * g->xres;
                    gf.ymax = g->ymax - (double)f[0] * g->yres;
                    gf.nrow = f[1] - f[0]; gf.ncol = f[3] - f[2];
                    rc2 = mhs_tps_predict_grid_dev(t, &gf, k[0] - f[0], k[1] - f[0], k[2] - f[2], k[3] - f[2], out_ptrs[job], kc, ls);
                }
                if (rc2) {
                    int expected = MHS_OK;
                    if (first_rc.compare_exchange_strong(expected, rc2)) {
                        std::lock_guard<std::mutex> lk(err_mu);
                        err_msg = mhs_last_error();   // thread-local in the worker: carry it to the caller
                    }
                    break;
                }
            }
        };
        {
            std::vector<std::thread> threads;
            for (int l = 1; l < nlanes; ++l) threads.emplace_back(worker, l);
            worker(0);
            for (std::thread &th : threads) th.join();
        }
        rc = first_rc.load();
        for (FitLane *L : lanes) {
            if (hipStreamSynchronize(L->s) != hipSuccess && !rc) rc = MHS_ERR_HIP;
            if (L->ms && hipStreamSynchronize(L->ms) != hipSuccess && !rc) rc = MHS_ERR_HIP;
        }
    }
    // ---- the batch's verdicts
    std::vector<SmallResult> res;
    if (int rc2 = small_batch_results(B, sb, res, nullptr)) { if (!rc) rc = rc2; }
    else if (hipStreamSynchronize(sb) != hipSuccess) { if (!rc) rc = MHS_ERR_HIP; }
    for (mhs_tps *t : handles) tps_free_quiet(t);      // every lane is idle: no wait, the blocks go back to the pool
    lap("wait for the kernels (+ lanes)");
    if (rc) { if (!err_msg.empty()) set_error("%s", err_msg.c_str()); return rc; }
    for (const SmallResult &r : res)
        if (r.status != 0.0) { set_error("mhs_tps_fit: GCV search failed"); return MHS_ERR_NUMERIC; }
    if (getenv("MHS_TIMING") && !res.empty()) {
        double t[8] = {0};
        for (const SmallResult &r : res) for (int q = 0; q < 8; ++q) t[q] += r.t_us[q] / (double)res.size();
        fprintf(stderr, "[run_tiles] %d tiles batched, %d on the lanes; mean us per batched fit: gram %.1f projection %.1f tridiagonalisation %.1f "
                "eigenvalues %.1f bracket+grid %.1f golden section %.1f solve+back-transform %.1f\n", (int)res.size(), (int)lane_jobs.size(),
                t[6], t[0], t[1], t[2], t[3], t[4], t[5]);
    }
    return MHS_OK;
}

extern "C" int mhs_tps_surface_dev(const mhs_grid *g, const double *xy, const double *resid, int64_t n,
                                   const double *cov1_at_stations, int64_t tile_edge, double lambda,
                                   int gcv_mode, double *out_dev, int64_t ld, int64_t *tiles_out,
                                   void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(g && xy && resid && out_dev && n > 0 && ld >= g->ncol, "bad arguments");
    int64_t nRx = 1, nCx = 1;
    if (tile_edge > 0)
        if (int rc = mhs_step3_tile_windows(g, tile_edge, 0.2, 0.025, &nRx, &nCx, nullptr, nullptr, 0)) return rc;
    const int64_t nt = nRx * nCx;
    if (tiles_out) { tiles_out[0] = nRx; tiles_out[1] = nCx; }
    hipStream_t s = pick_stream(stream);
    if (nt == 1) {  // V73:748-753
        mhs_tps *t = nullptr;
        if (int rc = mhs_tps_fit(xy, resid, n, lambda, gcv_mode, &t)) return rc;
        int rc = mhs_tps_predict_grid_dev(t, g, 0, g->nrow, 0, g->ncol, out_dev, ld, s);
        if (!rc) rc = (hipStreamSynchronize(s) == hipSuccess) ? MHS_OK : MHS_ERR_HIP;
        mhs_tps_free(t);
        return rc;
    }
    std::vector<int64_t> fit((size_t)nt * 4), keep((size_t)nt * 4), rows((size_t)n), cols((size_t)n);
    if (int rc = mhs_step3_tile_windows(g, tile_edge, 0.2, 0.025, &nRx, &nCx, fit.data(), keep.data(), nt)) return rc;
    if (int rc = mhs_cells_from_xy(g, xy, n, rows.data(), cols.data())) return rc;
    const bool timing = getenv("MHS_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mhs_tps_surface] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // the tiles' keep windows (~1.1 x the grid in total) live in one grow-only scratch buffer of the library
    struct TileBuf { double *p; };
    std::vector<TileBuf> bufs((size_t)nt);
    std::vector<const double *> ptrs((size_t)nt);
    {
        size_t total = 0;
        for (int64_t h = 0; h < nt; ++h) {
            const int64_t *k = &keep[(size_t)h * 4];
            total += ((size_t)((k[1] - k[0]) * (k[3] - k[2])) + 31) & ~(size_t)31;
        }
        Context &c = ctx();
        if (total > c.surface_arena_cap) {
            if (c.surface_arena) { (void)hipDeviceSynchronize(); (void)hipFree(c.surface_arena); c.surface_arena = nullptr; c.surface_arena_cap = 0; }
            MHS_HIP(hipMalloc((void **)&c.surface_arena, total * sizeof(double)));
            c.surface_arena_cap = total;
        }
        size_t off = 0;
        for (int64_t h = 0; h < nt; ++h) {
            const int64_t *k = &keep[(size_t)h * 4];
            bufs[h].p = c.surface_arena + off;
            ptrs[h] = bufs[h].p;
            off += ((size_t)((k[1] - k[0]) * (k[3] - k[2])) + 31) & ~(size_t)31;
        }
    }
    lap("windows + tile buffers");
    std::vector<double *> outs((size_t)nt);
    for (int64_t h = 0; h < nt; ++h) outs[(size_t)h] = bufs[(size_t)h].p;
    int rc = run_tiles(g, xy, resid, n, cov1_at_stations, fit, keep, rows, cols, lambda, gcv_mode, nullptr, nt, outs.data());
    lap("tile fits + evaluation");
    if (rc) return rc;
    rc = mosaic_feather_impl(g, nRx, nCx, keep.data(), ptrs.data(), 0, out_dev, ld, nullptr, s, true);      // spline planes: no NA
    if (timing) { (void)hipStreamSynchronize(s); lap("mosaic + feather"); }
    return rc;
}

extern "C" int mhs_tps_surface(const mhs_grid *g, const double *xy, const double *resid, int64_t n,
                               const double *cov1_at_stations, int64_t tile_edge, double lambda, int gcv_mode,
                               double *out_host, int64_t *tiles_out) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(g && out_host && g->nrow > 0 && g->ncol > 0, "bad arguments");
    // the plane comes from the library's persistent arena (no hipMalloc / hipFree per call)
    std::lock_guard<std::mutex> lk(pipe_mutex());
    if (int rc = host_pipe(sizeof(double) * (size_t)(g->nrow * g->ncol))) return rc;
    double *out = (double *)ctx().pipe_arena;
    hipStream_t s = ctx().pipe_comp;
    if (int rc = mhs_tps_surface_dev(g, xy, resid, n, cov1_at_stations, tile_edge, lambda, gcv_mode, out, g->ncol,
                                     tiles_out, s)) return rc;
    MHS_HIP(hipMemcpyAsync(out_host, out, sizeof(double) * (size_t)(g->nrow * g->ncol), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

extern "C" int mhs_tps_tiles_dev(const mhs_grid *g, const double *xy, const double *resid, int64_t n,
                                 const double *cov1_at_stations, int64_t tile_edge, double lambda, int gcv_mode,
                                 const int64_t *tile_ids, int64_t n_ids, double *const *out_dev_ptrs) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(g && xy && resid && n > 0 && tile_edge > 0 && n_ids >= 0 && (n_ids == 0 || (tile_ids && out_dev_ptrs)), "bad arguments");
    int64_t nRx = 1, nCx = 1;
    if (int rc = mhs_step3_tile_windows(g, tile_edge, 0.2, 0.025, &nRx, &nCx, nullptr, nullptr, 0)) return rc;
    const int64_t nt = nRx * nCx;
    for (int64_t k = 0; k < n_ids; ++k) MHS_REQUIRE(tile_ids[k] >= 0 && tile_ids[k] < nt && out_dev_ptrs[k], "tile id out of range or NULL output");
    std::vector<int64_t> fit((size_t)nt * 4), keep((size_t)nt * 4), rows((size_t)n), cols((size_t)n);
    if (int rc = mhs_step3_tile_windows(g, tile_edge, 0.2, 0.025, &nRx, &nCx, fit.data(), keep.data(), nt)) return rc;
    if (int rc = mhs_cells_from_xy(g, xy, n, rows.data(), cols.data())) return rc;
    return run_tiles(g, xy, resid, n, cov1_at_stations, fit, keep, rows, cols, lambda, gcv_mode, tile_ids, n_ids, out_dev_ptrs);
}
