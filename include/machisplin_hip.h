/*
 * machisplin_hip.h -- C ABI of the MI355X (gfx950) backend for MACHISPLIN's
 * data-parallel hot path: thin-plate-spline fit + grid evaluation, per-cell
 * evaluation of the six ensemble predictors, and the tile/mosaic/feather
 * bookkeeping that shards the grid.
 *
 * The reference (jasonleebrown/machisplin) has NO FFI: NAMESPACE:1-21 carries no
 * useDynLib and there is no src/.  Its operator boundary is R's S3 dispatch into
 * CRAN packages.  Every entry point below names the reference call site it
 * replaces (V73 = R/ensemble.machine.learning.thin.plate.splines.V73.R); the
 * .Call() shim that binds them is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every function returns an int
 *     status (MHS_OK == 0) and mhs_last_error() gives a thread-local message.
 *   - the caller owns every input/output buffer; the library owns the opaque
 *     handles (mhs_tps, mhs_model) and frees them in the *_free calls.
 *   - "host" entry points take host pointers (what R's REAL() hands over) and
 *     block until the result is in the output buffer.  "_dev" entry points take
 *     DEVICE pointers plus a hipStream_t (as void*; NULL = HIP's default stream,
 *     which is also torch's default) and only enqueue work on that stream; they
 *     are what a device-resident pipeline (and bench.py) uses.  Host entry points
 *     run on a private non-blocking stream of the library.
 *   - missing values: any IEEE NaN is NA (R's NA_real_ is a NaN payload) and NaN
 *     is written for NA results.
 *   - rasters are row-major from the NORTH-WEST cell (terra cell order,
 *     V73:128-133): cell (row, col) has centre
 *         x = xmin + (col + 0.5) * xres ,  y = ymax - (row + 0.5) * yres .
 *   - matrices handed over from R (xy) are COLUMN-major, as R stores them.
 */
#ifndef MACHISPLIN_HIP_H
#define MACHISPLIN_HIP_H

#include <stdint.h>

#if defined(__GNUC__)
#define MHS_API __attribute__((visibility("default")))
#else
#define MHS_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MHS_OK = 0,
    MHS_ERR_INVALID = 1,   /* bad argument */
    MHS_ERR_HIP = 2,       /* HIP runtime error */
    MHS_ERR_NODEVICE = 3,  /* no gfx950 device / mhs_init not called */
    MHS_ERR_NUMERIC = 4,   /* degenerate input: collinear stations, not SPD, ... */
    MHS_ERR_ALLOC = 5
};

/* covariate plane element types (terra holds doubles in RAM; the bundled rasters
 * are INT2S on disk, writeRaster default is FLT4S) */
enum { MHS_F64 = 0, MHS_F32 = 1, MHS_I16 = 2 };

/* lambda selection for mhs_tps_fit when lambda is NaN */
enum {
    MHS_GCV_FIELDS = 0,    /* gcv.Krig: 200-point grid + golden section, tol .01*GCVmin */
    MHS_GCV_CONVERGED = 1  /* same bracket, golden section on log(lambda) to 1e-13 */
};

/* geometry of a raster (terra::ext + terra::res + dim) */
typedef struct mhs_grid {
    double xmin, ymax;   /* west and north edge of the extent */
    double xres, yres;   /* cell size, both positive */
    int64_t nrow, ncol;
} mhs_grid;

typedef struct mhs_tps mhs_tps;      /* fitted thin-plate spline (class c("Krig","Tps")) */
typedef struct mhs_model mhs_model;  /* one fitted ensemble member */

/* ---------------------------------------------------------------- runtime -- */
MHS_API const char *mhs_last_error(void);
MHS_API const char *mhs_version(void);
/* select HIP device `device`, create the library stream, upload constant tables.
 * Idempotent for the same device; MHS_ERR_NODEVICE if there is no GPU. */
MHS_API int mhs_init(int device);
MHS_API int mhs_shutdown(void);
MHS_API int mhs_device_count(int *count);
MHS_API int mhs_sync(void *stream);
/* HIP-event timing on the stream work is launched on (bench.py's roofline leg):
 * t0 = mhs_timer_start(stream) ... launches ... mhs_timer_stop(stream,&ms). */
MHS_API int mhs_timer_start(void *stream);
MHS_API int mhs_timer_stop(void *stream, double *elapsed_ms);

/* ---------------------------------------------------------------- TPS fit --
 * replaces fields::Tps(x, Y)   V73:722 (per tile), V73:751 (single tile).
 * xy: N x 2 column-major (LONG column then LAT column), y: N residuals.
 * lambda: smoothing parameter on fields' scale; NaN => choose by GCV (gcv_mode).
 * Replicated locations are collapsed to weighted means as Krig does.
 * Gram assembly, null-space projection, tridiagonalisation and the Cholesky
 * solve run on the GPU; the O(n^2) tridiagonal eigenvalue sweep and the scalar
 * GCV search run on the host.                                                   */
MHS_API int mhs_tps_fit(const double *xy, const double *y, int64_t N, double lambda,
                int gcv_mode, mhs_tps **out);
/* Host-only helper of the fit (no GPU needed; exported so the host logic is testable on
 * a CPU box): given the tridiagonal form T = P'(Q2'KQ2)P (diag[m], offdiag[m-1]) and
 * g = P'Q2'y, evaluate fields' GCV criterion / choose lambda (NaN => search, gcv_mode)
 * and return q = (T + lambda I)^-1 g.  n_unique = m + 3; n_obs >= n_unique (replicates). */
MHS_API int mhs_host_gcv_tridiag(const double *diag, const double *offdiag, const double *g,
                                 int64_t m, int64_t n_unique, int64_t n_obs, double pure_ss,
                                 double lambda, int gcv_mode, double *lambda_out,
                                 double *gcv_out, double *eff_df_out, double *q_out);
/* build a spline object from coefficients captured elsewhere (e.g. from a real
 * fields::Tps object: $c, $d, $knots (scaled), $transform$x.center/$x.scale)   */
MHS_API int mhs_tps_from_coef(const double *knots_uv /* n x 2 column-major, scaled */,
                      const double *c, const double *d3, int64_t n, double lambda,
                      const double *center2, const double *scale2, mhs_tps **out);
MHS_API int mhs_tps_size(const mhs_tps *t, int64_t *n);
/* any output pointer may be NULL.  c[n], d3[3], knots_uv[n*2 column-major] */
MHS_API int mhs_tps_get(const mhs_tps *t, double *c, double *d3, double *knots_uv, double *lambda,
                double *center2, double *scale2, double *eff_df, double *gcv);
MHS_API int mhs_tps_free(mhs_tps *t);

/* --------------------------------------------------------------- TPS eval --
 * replaces terra::interpolate(terra::rast(rb), mod.tps.elev)  V73:726, V73:753
 * (predict.Krig on every cell centre of a geometry-only raster -- no NA mask).
 * Window [r0,r1) x [c0,c1) of grid g; out is (r1-r0) x ld row-major, ld >= c1-c0. */
MHS_API int mhs_tps_predict_grid(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1,
                         int64_t c0, int64_t c1, double *out_host);
MHS_API int mhs_tps_predict_grid_dev(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1,
                             int64_t c0, int64_t c1, double *out_dev, int64_t ld, void *stream);
/* predict(tps, xy): arbitrary points, xy n x 2 column-major (Step-5 station check) */
MHS_API int mhs_tps_predict_points(const mhs_tps *t, const double *xy, int64_t n, double *out_host);

#ifdef __cplusplus
}
#endif
#endif /* MACHISPLIN_HIP_H */
