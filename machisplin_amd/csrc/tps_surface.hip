// Step 3 + Step 4 of machisplin.mltps in ONE call (V73:636-897): tile grid, per-tile
// fields::Tps on the stations of the fit box, evaluation on the keep window, mean mosaic,
// seam feathering, overlay.  This is the entry point a .Call() shim binds so that the R
// side replaces the whole block by a single call; the Python mirror composes the same
// steps from the finer-grained entry points (machisplin_amd/mltps.py).
#include <cmath>
#include <vector>
#include "common.h"

using namespace mhs;

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
    std::vector<DevBuf<double>> bufs((size_t)nt);
    std::vector<const double *> ptrs((size_t)nt);
    std::vector<double> sx, sy, sr;
    int rc = MHS_OK;
    for (int64_t h = 0; h < nt && !rc; ++h) {
        const int64_t *f = &fit[(size_t)h * 4], *k = &keep[(size_t)h * 4];
        const int64_t kr = k[1] - k[0], kc = k[3] - k[2];
        if (hipError_t e = bufs[h].alloc((size_t)(kr * kc)); e != hipSuccess) return hip_fail(e, "alloc tile", __FILE__, __LINE__);
        ptrs[h] = bufs[h].p;
        sx.clear(); sy.clear(); sr.clear();
        for (int64_t i = 0; i < n; ++i) {  // terra::extract(rb[[1]], Full.cords) + complete.cases (V73:701-706)
            if (rows[i] < f[0] || rows[i] >= f[1] || cols[i] < f[2] || cols[i] >= f[3]) continue;
            if (cov1_at_stations && std::isnan(cov1_at_stations[i])) continue;
            if (std::isnan(resid[i])) continue;
            sx.push_back(xy[i]); sy.push_back(xy[n + i]); sr.push_back(resid[i]);
        }
        const int64_t m = (int64_t)sr.size();
        if (m < 10) {  // V73:710-721: the tile is all zeros
            if (hipMemsetAsync(bufs[h].p, 0, sizeof(double) * (size_t)(kr * kc), s) != hipSuccess) return MHS_ERR_HIP;
            continue;
        }
        std::vector<double> txy((size_t)2 * m);
        for (int64_t i = 0; i < m; ++i) { txy[i] = sx[i]; txy[m + i] = sy[i]; }
        mhs_tps *t = nullptr;
        rc = mhs_tps_fit(txy.data(), sr.data(), m, lambda, gcv_mode, &t);
        if (rc) break;
        // terra::interpolate(terra::rast(rb), tps): cell centres of the FIT raster (V73:726)
        mhs_grid gf = *g;
        gf.xmin = g->xmin + (double)f[2] * g->xres;
        gf.ymax = g->ymax - (double)f[0] * g->yres;
        gf.nrow = f[1] - f[0]; gf.ncol = f[3] - f[2];
        rc = mhs_tps_predict_grid_dev(t, &gf, k[0] - f[0], k[1] - f[0], k[2] - f[2], k[3] - f[2], bufs[h].p, kc, s);
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = MHS_ERR_HIP;  // knots are freed with the handle
        mhs_tps_free(t);
    }
    if (rc) return rc;
    return mhs_mosaic_feather_dev(g, nRx, nCx, keep.data(), ptrs.data(), 0, out_dev, ld, nullptr, s);
}

extern "C" int mhs_tps_surface(const mhs_grid *g, const double *xy, const double *resid, int64_t n,
                               const double *cov1_at_stations, int64_t tile_edge, double lambda, int gcv_mode,
                               double *out_host, int64_t *tiles_out) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(g && out_host && g->nrow > 0 && g->ncol > 0, "bad arguments");
    DevBuf<double> out;
    MHS_HIP(out.alloc((size_t)(g->nrow * g->ncol)));
    if (int rc = mhs_tps_surface_dev(g, xy, resid, n, cov1_at_stations, tile_edge, lambda, gcv_mode, out.p, g->ncol,
                                     tiles_out, ctx().stream)) return rc;
    MHS_HIP(hipMemcpy(out_host, out.p, sizeof(double) * (size_t)(g->nrow * g->ncol), hipMemcpyDeviceToHost));
    return MHS_OK;
}
