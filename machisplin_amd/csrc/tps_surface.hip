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

using namespace mhs;

constexpr int64_t TILE_LANES = 8;   // tiles fitted side by side (mhs_tps_surface)

// Fit and evaluate a set of Step-3 tiles side by side: job k = tile tile_ids[k] (or tile k), its keep-window plane
// (rows x cols of the window, contiguous) written to out_ptrs[k].  Returns after every lane has finished.
static int run_tiles(const mhs_grid *g, const double *xy, const double *resid, int64_t n, const double *cov1_at_stations,
                     const std::vector<int64_t> &fit, const std::vector<int64_t> &keep, const std::vector<int64_t> &rows,
                     const std::vector<int64_t> &cols, double lambda, int gcv_mode, const int64_t *tile_ids, int64_t njobs,
                     double *const *out_ptrs) {
    if (njobs <= 0) return MHS_OK;
    // The tiles' fits are chains of small, latency-bound kernels: several of them run side by side, each on
    // its own lane (two streams + work arena) driven by its own host thread.
    const int64_t want_lanes = TILE_LANES;
    const int nlanes = (int)std::min<int64_t>(njobs, want_lanes);
    std::vector<FitLane *> lanes((size_t)nlanes);
    for (int l = 0; l < nlanes; ++l)
        if (int rc = fit_lane(1 + l, &lanes[(size_t)l])) return rc;
    std::vector<mhs_tps *> handles((size_t)njobs, nullptr);
    std::atomic<int64_t> next{0};
    std::atomic<int> first_rc{MHS_OK};
    std::mutex err_mu;
    std::string err_msg;
    const int slot = current_slot();
    auto worker = [&](int lane_id) {
        FitLane &L = *lanes[(size_t)lane_id];
        SlotBind bind(slot);                // the slot and HIP's current device are per host thread
        // mhs_fit_reserve_cus active: the tiles' evaluations stay, like their fits, on the reserved compute units
        const hipStream_t ls = (ctx().reserved_cus > 0 && L.ms) ? L.ms : L.s;
        std::vector<double> sx, sy, sr, txy;
        for (;;) {
            const int64_t job = next.fetch_add(1);
            if (job >= njobs || first_rc.load() != MHS_OK) break;
            const int64_t h = tile_ids ? tile_ids[job] : job;
            double *dst = out_ptrs[job];
            const int64_t *f = &fit[(size_t)h * 4], *k = &keep[(size_t)h * 4];
            const int64_t kr = k[1] - k[0], kc = k[3] - k[2];
            sx.clear(); sy.clear(); sr.clear();
            for (int64_t i = 0; i < n; ++i) {  // terra::extract(rb[[1]], Full.cords) + complete.cases (V73:701-706)
                if (rows[i] < f[0] || rows[i] >= f[1] || cols[i] < f[2] || cols[i] >= f[3]) continue;
                if (cov1_at_stations && std::isnan(cov1_at_stations[i])) continue;
                if (std::isnan(resid[i])) continue;
                sx.push_back(xy[i]); sy.push_back(xy[n + i]); sr.push_back(resid[i]);
            }
            const int64_t m = (int64_t)sr.size();
            int rc = MHS_OK;
            if (m < 10) {  // V73:710-721: the tile is all zeros
                if (hipMemsetAsync(dst, 0, sizeof(double) * (size_t)(kr * kc), ls) != hipSuccess) rc = MHS_ERR_HIP;
            } else {
                txy.resize((size_t)2 * m);
                for (int64_t i = 0; i < m; ++i) { txy[i] = sx[i]; txy[m + i] = sy[i]; }
                mhs_tps *t = nullptr;
                rc = tps_fit_lane(L, txy.data(), sr.data(), m, lambda, gcv_mode, m < 1500 ? 2 : 0, &t);
                if (!rc) {
                    handles[(size_t)job] = t;   // freed after the last tile: hipFree synchronises the device
                    // terra::interpolate(terra::rast(rb), tps): cell centres of the FIT raster (V73:726)
                    mhs_grid gf = *g;
                    gf.xmin = g->xmin + (double)f[2] * g->xres;
                    gf.ymax = g->ymax - (double)f[0] * g->yres;
                    gf.nrow = f[1] - f[0]; gf.ncol = f[3] - f[2];
                    rc = mhs_tps_predict_grid_dev(t, &gf, k[0] - f[0], k[1] - f[0], k[2] - f[2], k[3] - f[2], dst, kc, ls);
                }
            }
            if (rc) {
                int expected = MHS_OK;
                if (first_rc.compare_exchange_strong(expected, rc)) {
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
    int rc = first_rc.load();
    for (FitLane *L : lanes) {
        if (hipStreamSynchronize(L->s) != hipSuccess && !rc) rc = MHS_ERR_HIP;
        if (L->ms && hipStreamSynchronize(L->ms) != hipSuccess && !rc) rc = MHS_ERR_HIP;
    }
    for (mhs_tps *t : handles) tps_free_quiet(t);      // every lane is idle: no wait, the blocks go back to the pool
    if (rc && !err_msg.empty()) set_error("%s", err_msg.c_str());
    return rc;
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
    rc = mhs_mosaic_feather_dev(g, nRx, nCx, keep.data(), ptrs.data(), 0, out_dev, ld, nullptr, s);
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
