// Thin-plate-spline evaluation on gfx950: predict.Krig over every cell centre of a
// raster window (terra::interpolate, V73:726,753) and over arbitrary points.
//
//   f(x,y) = d0 + d1 u + d2 v + sum_j c_j (1/8pi) 0.5 log(r2) r2 ,  r2 = |(u,v)-(u_j,v_j)|^2
//
// Regime: FP64 vector-ALU bound (8 N flop per cell against 8 B written), not HBM
// bound.  Layout of the work:
//   * a wave owns 64 consecutive columns x ROWS consecutive rows; lane = column, so
//     dx and dx^2 are shared by the lane's ROWS cells and the stores are coalesced;
//   * knots are wave-uniform, so they arrive through the scalar cache (s_load) and
//     cost no VALU or LDS bandwidth;
//   * log(r2) is computed in-kernel from a 1024-entry {1/c, log c} table staged in
//     LDS: r2 = 2^e m, log r2 = e ln2 + log c_i + log1p(m/c_i - 1), |m/c_i - 1| <= 2^-11,
//     cubic log1p => absolute error < 2e-14 (ocml's log costs ~70 FP64 instructions).
//     Neighbouring lanes see neighbouring r2, so table reads mostly broadcast.
#include <cmath>
#include <cstring>
#include "common.h"
#include "devmath.h"

namespace mhs {

constexpr int EVAL_ROWS = 4;       // rows per lane
constexpr int EVAL_WAVES = 4;      // waves per block, stacked along rows
constexpr int EVAL_TILE_ROWS = EVAL_ROWS * EVAL_WAVES;

struct EvalGeom {
    double xmin, ymax, xres, yres;  // grid affine
    double cx, cy, sx, sy;          // fields transform (x.center, x.scale)
    double d0, d1, d2;
    int64_t r0, c0;                 // window origin in the grid
    int nr, nc;                     // window size
    int64_t ld;                     // output leading dimension
};

__global__ __launch_bounds__(64 * EVAL_WAVES) void tps_eval_grid_kernel(
    const Knot *__restrict__ knots, int n, const double2 *__restrict__ gtab, EvalGeom g,
    double *__restrict__ out) {
    __shared__ double2 tab[LOG_TAB_N];
    stage_log_table(tab, gtab);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int row0 = blockIdx.y * EVAL_TILE_ROWS + wave * EVAL_ROWS;
    if (row0 >= g.nr) return;

    // cell centre -> scaled coordinates, same operation order as the host/oracle
    const double x = g.xmin + ((double)(g.c0 + col) + 0.5) * g.xres;
    const double u = (x - g.cx) / g.sx;
    double v[EVAL_ROWS], acc[EVAL_ROWS];
#pragma unroll
    for (int k = 0; k < EVAL_ROWS; ++k) {
        const double y = g.ymax - ((double)(g.r0 + row0 + k) + 0.5) * g.yres;
        v[k] = (y - g.cy) / g.sy;
        acc[k] = 0.0;
    }

#pragma unroll 2
    for (int j = 0; j < n; ++j) {
        const Knot kn = knots[j];
        const double dx = u - kn.u;
        const double dx2 = dx * dx;
#pragma unroll
        for (int k = 0; k < EVAL_ROWS; ++k) {
            const double dy = v[k] - kn.v;
            const double dd = fma(dy, dy, dx2);
            acc[k] = fma(kn.cw, r2logr2(dd, tab), acc[k]);
        }
    }

    if (col < g.nc) {
#pragma unroll
        for (int k = 0; k < EVAL_ROWS; ++k)
            if (row0 + k < g.nr)
                out[(int64_t)(row0 + k) * g.ld + col] = g.d0 + g.d1 * u + g.d2 * v[k] + acc[k];
    }
}

__global__ __launch_bounds__(256) void tps_eval_points_kernel(
    const Knot *__restrict__ knots, int n, const double2 *__restrict__ gtab, EvalGeom g,
    const double *__restrict__ px, const double *__restrict__ py, int64_t npts,
    double *__restrict__ out) {
    __shared__ double2 tab[LOG_TAB_N];
    stage_log_table(tab, gtab);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ii = i < npts ? i : npts - 1;
    const double u = (px[ii] - g.cx) / g.sx;
    const double v = (py[ii] - g.cy) / g.sy;
    double acc = 0.0;
    for (int j = 0; j < n; ++j) {
        const Knot kn = knots[j];
        const double dx = u - kn.u, dy = v - kn.v;
        const double dd = fma(dy, dy, dx * dx);
        acc = fma(kn.cw, r2logr2(dd, tab), acc);
    }
    if (i < npts) out[i] = g.d0 + g.d1 * u + g.d2 * v + acc;
}

int upload_knots(mhs_tps *t) {
    std::vector<Knot> h((size_t)t->n);
    const double k = 0.5 / (8.0 * M_PI);
    for (int64_t j = 0; j < t->n; ++j) {
        h[j].u = t->knots_uv[j];
        h[j].v = t->knots_uv[t->n + j];
        h[j].cw = t->c[j] * k;
        h[j].pad = 0.0;
    }
    if (t->knots_dev) { (void)hipFree(t->knots_dev); t->knots_dev = nullptr; }
    MHS_HIP(hipMalloc((void **)&t->knots_dev, sizeof(Knot) * (size_t)(t->n ? t->n : 1)));
    MHS_HIP(hipMemcpy(t->knots_dev, h.data(), sizeof(Knot) * (size_t)t->n, hipMemcpyHostToDevice));
    return MHS_OK;
}

static EvalGeom make_geom(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0,
                          int64_t c1, int64_t ld) {
    EvalGeom e;
    if (g) { e.xmin = g->xmin; e.ymax = g->ymax; e.xres = g->xres; e.yres = g->yres; }
    else { e.xmin = e.ymax = 0; e.xres = e.yres = 1; }
    e.cx = t->center[0]; e.cy = t->center[1]; e.sx = t->scale[0]; e.sy = t->scale[1];
    e.d0 = t->d[0]; e.d1 = t->d[1]; e.d2 = t->d[2];
    e.r0 = r0; e.c0 = c0; e.nr = (int)(r1 - r0); e.nc = (int)(c1 - c0); e.ld = ld;
    return e;
}

}  // namespace mhs

using namespace mhs;

extern "C" {

int mhs_tps_from_coef(const double *knots_uv, const double *c, const double *d3, int64_t n,
                      double lambda, const double *center2, const double *scale2, mhs_tps **out) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(knots_uv && c && d3 && center2 && scale2 && out, "NULL argument");
    MHS_REQUIRE(n >= 1 && n < (1LL << 31), "n out of range");
    MHS_REQUIRE(scale2[0] > 0 && scale2[1] > 0, "scale must be positive");
    mhs_tps *t = new mhs_tps();
    t->n = n;
    t->lambda = lambda;
    t->eff_df = t->gcv = NAN;
    t->c.assign(c, c + n);
    t->knots_uv.assign(knots_uv, knots_uv + 2 * n);
    memcpy(t->d, d3, sizeof(t->d));
    memcpy(t->center, center2, sizeof(t->center));
    memcpy(t->scale, scale2, sizeof(t->scale));
    if (int rc = upload_knots(t)) { mhs_tps_free(t); return rc; }
    *out = t;
    return MHS_OK;
}

int mhs_tps_size(const mhs_tps *t, int64_t *n) {
    MHS_REQUIRE(t && n, "NULL argument");
    *n = t->n;
    return MHS_OK;
}

int mhs_tps_get(const mhs_tps *t, double *c, double *d3, double *knots_uv, double *lambda,
                double *center2, double *scale2, double *eff_df, double *gcv) {
    MHS_REQUIRE(t != nullptr, "NULL handle");
    if (c) memcpy(c, t->c.data(), sizeof(double) * (size_t)t->n);
    if (d3) memcpy(d3, t->d, sizeof(t->d));
    if (knots_uv) memcpy(knots_uv, t->knots_uv.data(), sizeof(double) * 2 * (size_t)t->n);
    if (lambda) *lambda = t->lambda;
    if (center2) memcpy(center2, t->center, sizeof(t->center));
    if (scale2) memcpy(scale2, t->scale, sizeof(t->scale));
    if (eff_df) *eff_df = t->eff_df;
    if (gcv) *gcv = t->gcv;
    return MHS_OK;
}

int mhs_tps_free(mhs_tps *t) {
    if (!t) return MHS_OK;
    if (t->knots_dev) (void)hipFree(t->knots_dev);
    delete t;
    return MHS_OK;
}

int mhs_tps_predict_grid_dev(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1,
                             int64_t c0, int64_t c1, double *out_dev, int64_t ld, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(t && g && out_dev, "NULL argument");
    MHS_REQUIRE(g->nrow > 0 && g->ncol > 0 && g->xres > 0 && g->yres > 0, "bad grid geometry");
    MHS_REQUIRE(0 <= r0 && r0 <= r1 && r1 <= g->nrow && 0 <= c0 && c0 <= c1 && c1 <= g->ncol,
                "window outside the grid");
    MHS_REQUIRE(ld >= c1 - c0, "ld smaller than the window width");
    MHS_REQUIRE(r1 - r0 < (1LL << 30) && c1 - c0 < (1LL << 30), "window too large");
    if (r1 == r0 || c1 == c0) return MHS_OK;
    const EvalGeom e = make_geom(t, g, r0, r1, c0, c1, ld);
    dim3 grid((unsigned)((e.nc + 63) / 64), (unsigned)((e.nr + EVAL_TILE_ROWS - 1) / EVAL_TILE_ROWS));
    // gridDim.y is limited to 65535: 16 rows per block covers > 1e6 rows
    MHS_REQUIRE(grid.y <= 65535u, "too many rows for one launch");
    hipLaunchKernelGGL(tps_eval_grid_kernel, grid, dim3(64 * EVAL_WAVES), 0, pick_stream(stream),
                       t->knots_dev, (int)t->n, ctx().log_tab, e, out_dev);
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

int mhs_tps_predict_grid(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0,
                         int64_t c1, double *out_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(t && g && out_host, "NULL argument");
    MHS_REQUIRE(0 <= r0 && r0 <= r1 && 0 <= c0 && c0 <= c1, "bad window");
    const int64_t nr = r1 - r0, nc = c1 - c0;
    if (nr == 0 || nc == 0) return MHS_OK;
    DevBuf<double> buf;
    MHS_HIP(buf.alloc((size_t)(nr * nc)));
    if (int rc = mhs_tps_predict_grid_dev(t, g, r0, r1, c0, c1, buf.p, nc, nullptr)) return rc;
    MHS_HIP(hipMemcpyAsync(out_host, buf.p, sizeof(double) * (size_t)(nr * nc), hipMemcpyDeviceToHost,
                           ctx().stream));
    MHS_HIP(hipStreamSynchronize(ctx().stream));
    return MHS_OK;
}

int mhs_tps_predict_points(const mhs_tps *t, const double *xy, int64_t n, double *out_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(t && out_host && (xy || n == 0), "NULL argument");
    MHS_REQUIRE(n >= 0, "negative n");
    if (n == 0) return MHS_OK;
    DevBuf<double> dxy, dout;
    MHS_HIP(dxy.alloc((size_t)(2 * n)));
    MHS_HIP(dout.alloc((size_t)n));
    hipStream_t s = ctx().stream;
    MHS_HIP(hipMemcpyAsync(dxy.p, xy, sizeof(double) * 2 * (size_t)n, hipMemcpyHostToDevice, s));
    const EvalGeom e = make_geom(t, nullptr, 0, 0, 0, 0, 0);
    hipLaunchKernelGGL(tps_eval_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       t->knots_dev, (int)t->n, ctx().log_tab, e, dxy.p, dxy.p + n, n, dout.p);
    MHS_HIP(hipGetLastError());
    MHS_HIP(hipMemcpyAsync(out_host, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

}  // extern "C"
