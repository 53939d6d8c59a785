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
 * Idempotent for the same device; MHS_ERR_NODEVICE if there is no GPU.  Several devices from one
 * process: mhs_init_devices (section "several devices"); mhs_init(d) is mhs_init_devices(1, &d)
 * and leaves an earlier mhs_init_devices whose slot 0 sits on d untouched. */
MHS_API int mhs_init(int device);
MHS_API int mhs_shutdown(void);
MHS_API int mhs_device_count(int *count);
MHS_API int mhs_sync(void *stream);
/* Scheduling knob for a fit that runs beside an ensemble evaluation (the reference computes Step 2's rasters and
 * Step 3's fields::Tps fit one after the other, V73:468-620 then 722-753; here the fit only needs the members'
 * predictions at the stations, so it can run while the grid is still being evaluated -- but grid-filling kernels leave
 * the fit's chain of small dependent kernels no workgroup slots).  While n_cus > 0 (a multiple of 8: n_cus / 8
 * compute units of every XCD):
 *   - ONE long member of every mhs_ensemble_predict(_dev) / mhs_members_predict_dev call on a window of >= 2^22 cells
 *     (randomForest if present, else ksvm, else gbm) is launched on a stream whose CU mask leaves those compute units
 *     out, fenced by events so that it keeps its place in the caller's stream order (results are unchanged);
 *   - mhs_tps_fit's GCV route runs on streams confined to exactly those compute units.
 * CU-masked streams are BLOCKING streams (hipExtStreamCreateWithCUMask takes no flags): they synchronise with the NULL
 * stream, so the ensemble call must be given a non-NULL, non-blocking stream or the fit waits for it after all.
 * 0 switches it off (the default).  *previous may be NULL. */
MHS_API int mhs_fit_reserve_cus(int n_cus, int *previous);
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
/* `count` independent fits in one call -- what the reference's tiled Step 3 is (V73:690-738: one fields::Tps per tile on
 * the 130-250 stations of its fit box), and what a caller with several response layers has.  Fit k reads N[k] rows:
 * xy[k] is N[k] x 2 column-major, y[k] N[k] residuals.  Every fit with 8..256 distinct locations is done by ONE kernel
 * launch, one workgroup per fit (Gram matrix, null-space projection, tridiagonalisation in registers, the GCV search of
 * mhs_tps_fit with its independent evaluations side by side, solve, back-transform: csrc/tps_batch.hip); larger ones
 * take mhs_tps_fit's route one after the other.  out[k] = NULL and status[k] != MHS_OK for a fit that failed
 * (collinear stations, ...); the call itself fails only on bad arguments or a HIP error.  status may be NULL. */
MHS_API int mhs_tps_fit_many(const double *const *xy, const double *const *y, const int64_t *N, int64_t count,
                             double lambda, int gcv_mode, mhs_tps **out, int *status);
/* Several response layers on one station table (the layer loop of machisplin.mltps, V73:176-957: BIO 1..12 on the same
 * stations; V73:154 keeps the same rows for every layer): B = Q2'KQ2 and its reduction depend on the coordinates only,
 * yet every fields::Tps call reduces it again.  Between mhs_tps_reduction_cache(1) and mhs_tps_reduction_cache(0),
 * GCV fits on the small route (<= 259 stations after replicate collapse: the reference-tiled mode's tiles) keep the
 * reduction of each station set (reflectors, tridiagonal, projected rows) and later fits of the SAME coordinates and
 * replicate weights send only their right-hand side through it -- bit for bit the result of a full fit.  (0) frees
 * everything that was kept.  Scope it to one multi-layer call: nothing is reused across calls of the host's loop.    */
MHS_API int mhs_tps_reduction_cache(int enable);
/* Host-only helper of the fit (no GPU needed; exported so the host logic is testable on
 * a CPU box): given the tridiagonal form T = P'(Q2'KQ2)P (diag[m], offdiag[m-1]) and
 * g = P'Q2'y, evaluate fields' GCV criterion / choose lambda (NaN => search, gcv_mode)
 * and return q = (T + lambda I)^-1 g.  n_unique = m + 3; n_obs >= n_unique (replicates). */
MHS_API int mhs_host_gcv_tridiag(const double *diag, const double *offdiag, const double *g,
                                 int64_t m, int64_t n_unique, int64_t n_obs, double pure_ss,
                                 double lambda, int gcv_mode, double *lambda_out,
                                 double *gcv_out, double *eff_df_out, double *q_out);
/* The same on the symmetric BANDED form the GPU fit reduces to (bandwidth bw, lower band
 * column-major: ab[d + (bw+1) j] = M[j+d][j]); bw = 1 is the tridiagonal case. */
MHS_API int mhs_host_gcv_band(const double *ab, int bw, const double *g, int64_t m, int64_t n_unique,
                              int64_t n_obs, double pure_ss, double lambda, int gcv_mode,
                              double *lambda_out, double *gcv_out, double *eff_df_out, double *q_out);
/* Test hooks of the round-4 GCV route (tps_band32.hip; the route itself runs inside mhs_tps_fit, fields::Tps at V73:722,
 * V73:751).  mhs_band32_reduce: the 32-column-panel band reduction of a symmetric matrix B (m x m, column-major, m >= 66)
 * with a right-hand side: ab[j * 33 + d] = Bb[j + d][j] (B = Q Bb Q'), gq = Q'g, and Qr = Q r for a probe vector r;
 * *breakdown = 1 when a panel's Cholesky-QR met a non-positive pivot (the fit then falls back to the 8-column route).
 * mhs_band32_gcv_terms: for each lambda[k] the number of negative pivots of Bb + lambda I (= eigenvalues of Bb below
 * -lambda), tr (Bb + lambda I)^-1 and g'(Bb + lambda I)^-2 g from the twisted, self-differentiating LDL' sweep (deriv = 0:
 * the counts only).  mhs_band32_solve: q = (Bb + lambda I)^-1 g by the same sweep with the factor stored.            */
MHS_API int mhs_band32_reduce(const double *B, const double *g, int64_t m, double *ab, double *gq,
                              const double *r, double *Qr, int *breakdown);
MHS_API int mhs_band32_gcv_terms(const double *ab, const double *g, int64_t m, const double *lambda, int n_lambda,
                                 int deriv, double *neg, double *tr_inv, double *gM2g);
MHS_API int mhs_band32_solve(const double *ab, const double *g, int64_t m, double lambda, double *q);
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
/* How the grid evaluation sums the knots (process-wide): 0 = choose by cost (default), 1 = direct sum
 * over all knots for every cell, 2 = far-field-interpolated (direct sum over the knots near a tile,
 * 16 x 16 Chebyshev interpolation of the analytic sum over the rest; equal to the direct sum to FP64
 * rounding, see csrc/tps_eval.hip).  predict.Krig itself is the direct sum. */
MHS_API int mhs_tps_eval_mode(int mode);
/* What the last grid evaluation of this handle did: tile size in cells (0 x 0 = direct sum) and the number of
 * kernel evaluations phi(r) it performed at tile nodes (far knots) and at cells (near knots).  For profiling. */
MHS_API int mhs_tps_eval_plan(const mhs_tps *t, int *tile_cols, int *tile_rows, int64_t *node_pairs,
                      int64_t *cell_pairs);
MHS_API int mhs_tps_predict_grid_dev(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1,
                             int64_t c0, int64_t c1, double *out_dev, int64_t ld, void *stream);
/* Rows [b0, b1) of the window [r0, r1) x [c0, c1), evaluated with the WINDOW's own plan (far-field tile size and origin, path
 * decision): out_dev holds row b0 first.  What a device that owns a row band of the grid calls: every cell gets exactly the
 * arithmetic of the one-piece evaluation, so the bands of N devices stitch to the one-device plane bit for bit. */
MHS_API int mhs_tps_predict_rows_dev(const mhs_tps *t, const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                                     int64_t b0, int64_t b1, double *out_dev, int64_t ld, void *stream);
/* predict(tps, xy): arbitrary points, xy n x 2 column-major (Step-5 station check) */
MHS_API int mhs_tps_predict_points(const mhs_tps *t, const double *xy, int64_t n, double *out_host);

/* ------------------------------------------------------- ensemble members --
 * Loaders take the FLAT parameter arrays of the fitted R objects (the shim extracts
 * them with REAL()/INTEGER()); the library copies them to the device.  p = number of
 * predictors = covariate layers + 2 (rast_stack <- c(covar.ras, LONG, LAT), V73:138).  */

/* mgcv::gam(resp ~ a+b+...) has no s() terms (V73:195,600): a linear model.
 * coef[p+1], intercept first.  replaces terra::predict(rast_stack, gam) V73:604,606 */
MHS_API int mhs_lm_load(const double *coef, int p, mhs_model **out);
/* The fit of that member: least squares by Householder QR of [1 X] on the device, as mgcv::gam / stats::lm solve a
 * purely parametric formula.  X: n x p column-major (rast_stack order), y: n responses, no NA rows (V73:154);
 * coef[p+1], intercept first; MHS_ERR_NUMERIC for a rank-deficient design.
 * replaces mgcv::gam(mod.form, data = train) V73:252 (CV folds) and V73:600 (final fit)      */
MHS_API int mhs_lm_fit(const double *X, const double *y, int64_t n, int p, double *coef);
/* kernlab::ksvm(mod.form, data) for a numeric response -- type eps-svr, kernel rbfdot, scaled = TRUE, C = 1,
 * epsilon = 0.1, tol = 0.001 are kernlab's defaults and the reference passes none (V73:251 in the CV loop, V73:560 for
 * the final model).  X: n x p column-major, y: n responses, no NA rows.  The columns and the response are scaled to
 * zero mean / unit sd (n - 1), the dual is solved by the libsvm SMO kernlab uses (second-order working-set
 * selection, stop when the maximal KKT violation < tol) in one resident kernel over a Gram matrix kept in HBM.
 * sigma = kpar$sigma (kernlab's "automatic" value comes from sigest() on a RANDOM half of the rows: pass it in).
 * Outputs: beta[n] = alpha_i - alpha*_i (0 for rows that are not support vectors; the non-zero ones with their scaled
 * rows are mhs_svr_load's alpha / sv), b, scaling.  max_iter <= 0: libsvm's max(10^7, 100 n).  MHS_ERR_NUMERIC if the
 * iteration limit is hit.  replaces kernlab::ksvm V73:251, V73:560                                              */
MHS_API int mhs_svr_fit(const double *X, const double *y, int64_t n, int p, double sigma, double C, double epsilon,
                        double tol, int64_t max_iter, double *beta, double *b, double *x_center, double *x_scale,
                        double *y_center, double *y_scale, int64_t *n_iter);
/* nnet::nnet(mod.form, data = trainNN, size = 10, linout = TRUE, maxit = 10000) (V73:249, V73:463): least squares
 * (sum over rows of (y - yhat)^2, decay 0) by R's optim "BFGS" (vmmin; abstol = 1e-4, reltol = 1e-8 are nnet's
 * defaults) in one resident kernel.  wts: IN the initial weights (nnet draws runif(-0.7, 0.7): RNG-dependent, so
 * they are the caller's), OUT the fitted ones, nnet order (mhs_nnet_load).  X: n x p column-major, y as the caller
 * scaled it (V73:455-459).  value: final objective; counts[2]: function / gradient evaluations; fail: 1 = maxit
 * reached.  size must be 10.  replaces nnet::nnet V73:249, V73:463                                              */
MHS_API int mhs_nnet_fit(const double *X, const double *y, int64_t n, int p, int size, double *wts, int maxit,
                         double abstol, double reltol, double *value, int *counts, int *fail);
/* nnet::nnet(size, linout=TRUE) (V73:463): wts in nnet order -- per hidden unit its bias
 * then p input weights, then output bias and `size` hidden->output weights.  The
 * response un-scaling pred*max2.resp.f + min.resp.f (V73:469-470) is y_scale/y_shift.
 * replaces terra::predict(rast_stack, nnet) V73:468,472                              */
MHS_API int mhs_nnet_load(const double *wts, int p, int size, double y_scale, double y_shift,
                          mhs_model **out);
/* earth::earth (V73:539): selected terms only; dirs/cuts are nterms x p ROW-major
 * (dirs: 0 unused, 1 max(0,x-cut), -1 max(0,cut-x), 2 linear).
 * replaces terra::predict(rast_stack, earth) V73:543,545                              */
MHS_API int mhs_earth_load(const double *coef, const int32_t *dirs, const double *cuts, int nterms,
                           int p, mhs_model **out);
/* kernlab::ksvm eps-svr, rbfdot, scaled=TRUE (V73:560): alpha[nsv] signed coefficients,
 * sv nsv x p ROW-major support vectors in scaled coordinates, b, kpar$sigma,
 * scaling$x.scale (center/scale, p each) and scaling$y.scale.
 * replaces terra::predict(rast_stack, ksvm, na.rm=TRUE) V73:582,584                   */
MHS_API int mhs_svr_load(const double *alpha, const double *sv, int64_t nsv, int p, double b,
                         double sigma, const double *x_center, const double *x_scale,
                         double y_center, double y_scale, mhs_model **out);
/* gbm object cut at n.trees = best.trees (V73:497): concatenated per-tree node arrays
 * (SplitVar 0-based / -1 terminal, SplitCodePred, LeftNode, RightNode, MissingNode; child
 * indices tree-local), tree t owns nodes tree_offsets[t] .. tree_offsets[t+1]-1.
 * replaces terra::predict(rast_stack, gbm, n.trees=, type="response") V73:497,499     */
MHS_API int mhs_gbm_load(double init_f, int64_t n_trees, const int64_t *tree_offsets,
                         const int32_t *split_var, const double *split_val, const int32_t *left,
                         const int32_t *right, const int32_t *missing, int p, mhs_model **out);
/* randomForest regression $forest (V73:517): per-tree columns concatenated
 * (leftDaughter/rightDaughter 1-based tree-local, nodestatus -1 terminal, bestvar
 * 1-based, xbestsplit, nodepred).
 * replaces terra::predict(rast_stack, randomForest, type="response", ...) V73:521,523 */
MHS_API int mhs_rf_load(int64_t n_trees, const int64_t *tree_offsets, const int32_t *left,
                        const int32_t *right, const int32_t *status, const int32_t *best_var,
                        const double *split, const double *node_pred, int p, mhs_model **out);
MHS_API int mhs_model_free(mhs_model *m);

/* the covariate layers of rast_stack, planar; LONG and LAT are generated from the grid */
typedef struct mhs_stack {
    const void *data;     /* layer k at data + k*plane_stride elements, row-major            */
    int32_t n_layers;     /* C = p - 2                                                       */
    int32_t dtype;        /* MHS_F64 / MHS_F32 / MHS_I16                                     */
    int64_t plane_stride; /* elements between layers                                         */
    int64_t ld;           /* elements between rows (>= ncol of the grid)                     */
    double nodata;        /* value that means NA (e.g. -32768 for INT2S); NaN = none.  NaN   */
                          /* cells are always NA                                             */
} mhs_stack;

/* terra::predict(rast_stack, model) over the window [r0,r1) x [c0,c1) of grid g:
 *   accumulate == 0:  out  = pred * weight       (V73:475,499,523,545,584,606)
 *   accumulate != 0:  out += pred * weight       (V73:471,498,522,544,583,605)
 * `covars` describes DEVICE planes covering the whole grid g; out is (r1-r0) x ld.      */
MHS_API int mhs_predict_dev(const mhs_model *m, const mhs_grid *g, const mhs_stack *covars,
                            int64_t r0, int64_t r1, int64_t c0, int64_t c1, double weight,
                            int accumulate, double *out_dev, int64_t ld, void *stream);
/* out (+)= sum_k weights[k] * pred_k over the window, members in order -- the accumulation lines of the Step-2 loop
 * (pred.elev <- pred.elev + pred * wt, V73:471,498,522,544,583,605) without the final division; `accumulate` = 0 starts
 * from the first member's plane, 1 adds to what `out_dev` holds.  A run of the consecutive members gam, nnet, earth
 * (V73:340-362 order) is evaluated in one pass over the planes; results are bit-identical to mhs_predict_dev calls. */
MHS_API int mhs_members_predict_dev(const mhs_model *const *models, const double *weights, int n_models,
                            const mhs_grid *g, const mhs_stack *covars, int64_t r0, int64_t r1, int64_t c0,
                            int64_t c1, int accumulate, double *out_dev, int64_t ld, void *stream);
/* the whole Step-2 raster loop V73:447-619: out = (((p1 w1) + p2 w2) + ...) / wt_total,
 * models in mods.run order, weights = the rounded kept weights, wt_total = the
 * unrounded OptX.mfit.wt.tot (V73:337).                                                  */
MHS_API int mhs_ensemble_predict_dev(const mhs_model *const *models, const double *weights,
                                     int n_models, double wt_total, const mhs_grid *g,
                                     const mhs_stack *covars, int64_t r0, int64_t r1, int64_t c0,
                                     int64_t c1, double *out_dev, int64_t ld, void *stream);
/* same, host pointers (covars->data and out on the host) */
MHS_API int mhs_ensemble_predict(const mhs_model *const *models, const double *weights,
                                 int n_models, double wt_total, const mhs_grid *g,
                                 const mhs_stack *covars, int64_t r0, int64_t r1, int64_t c0,
                                 int64_t c1, double *out_host);
/* predict(model, data.frame): X is n x p COLUMN-major (all p predictors given, LONG and
 * LAT included), out[n].  Station residuals V73:477-482,501-505,...                     */
MHS_API int mhs_predict_points(const mhs_model *m, const double *X, int64_t n, double *out_host);
/* gbm::predict.gbm(model, newdata, n.trees = step, 2 step, ...) for a table of points in ONE walk over the trees:
 * out_host[(k - 1) * n + i] = prediction of row i with the first k * step trees, k = 1 .. n_trees / step.  This is
 * what machisplin.gbm.step evaluates on every fold's hold-out rows after each gbm.more (V73:1843, 1919) to build
 * the hold-out deviance curve its tree-count search runs on (V73:1884-1981).  X: n x p column-major.             */
MHS_API int mhs_gbm_staged_points(const mhs_model *m, const double *X, int64_t n, int step, double *out_host);
/* What a loaded model is: kind (0 lm, 1 nnet, 2 earth, 3 ksvm, 4 gbm, 5 randomForest), its number of predictors p and,
 * for the tree ensembles, its tree count (0 otherwise).  Callers size their outputs from it -- mhs_gbm_staged_points
 * writes n * (n_trees / step) values whatever n.trees the R side believes the model has.  Any output pointer may be NULL. */
MHS_API int mhs_model_info(const mhs_model *m, int *kind, int *p, int64_t *n_trees);

/* Diagnostic: what the most recent gbm evaluation of a window of >= 2^20 cells measured before choosing its kernel (the
 * probe of gbm_coherent_kernel: 64 tiles x 256 trees classified).  cost / count = the coherent kernel's estimated time per
 * tree and wave in hundredths of the tree-order kernel's, without its fixed 12; below 83 the coherent kernel ran.  count = 0:
 * no probe has run (small windows, non-grid calls, or a model that is not gbm).  Synchronises the device. */
MHS_API int mhs_gbm_probe_last(const mhs_model *m, int64_t *cost, int64_t *count);
/* res.FINAL in one call (V73:477-482, 501-505, 525-528, 547-549, 586-589, 608-611, 620): the kept members at the
 * n station rows X (as above), out[i] = ((resp_i - pred_1) w_1 + (resp_i - pred_2) w_2 + ...) / wt_total, accumulated
 * member after member; weights = the rounded kept weights, wt_total the unrounded total (at most 8 members). */
MHS_API int mhs_residual_points(const mhs_model *const *models, const double *weights, int n_models, double wt_total,
                                const double *X, const double *resp, int64_t n, double *out_host);

/* out = a / divisor (b == NULL) or a / divisor + b: pred.elev / wt.tot (V73:619) and the
 * Step-5 sum pred + TPS (V73:906-907); NaN if either is NaN.  n elements, device. */
MHS_API int mhs_scale_add_dev(const double *a, double divisor, const double *b, double *out,
                              int64_t n, void *stream);

/* ------------------------------------------------------- tile bookkeeping --
 * Integer windows are half-open [r0,r1) x [c0,c1) in the full grid, rows from the north;
 * a window array holds 4 int64 per tile: r0, r1, c0, c1.  Tiles are numbered row-major
 * from the SOUTH-WEST as the reference does (V73:670-681, 1192-1197).  The host functions
 * need no GPU.                                                                           */

/* terra::crop(x, e): SpatRaster::align(e, "near") intersected with x's extent, window from
 * colFromX/rowFromY half a cell inside.  ext4 = xmin, xmax, ymin, ymax (terra::ext order).
 * replaces the geometry of terra::crop at V73:699,728,779-780,835-836,1207,1415-1416     */
MHS_API int mhs_crop_window(const mhs_grid *g, const double *ext4, int64_t *win4);
/* Step-3 TPS tile grid (V73:656-681): nRx = ceil(nrow/tile_edge), nCx likewise; fit box =
 * tile +- fit_overlap (0.2), keep box = tile +- keep_overlap (0.025); fit_win = crop(rast_stack,
 * b) (V73:699), keep_win = crop(pred, d) (V73:728) in full-grid indices.  Pass NULL arrays to
 * query nRx/nCx.  The reference hard-codes tile_edge = 1500.                               */
MHS_API int mhs_step3_tile_windows(const mhs_grid *g, int64_t tile_edge, double fit_overlap,
                                   double keep_overlap, int64_t *nRx, int64_t *nCx,
                                   int64_t *fit_win, int64_t *keep_win, int64_t capacity);
/* machisplin.tiles.create (V73:1165-1208): tile box = grid cell +- feather_d/2 pixels;
 * boxes[4*n] (xmin,xmax,ymin,ymax) and crop windows win[4*n].                              */
MHS_API int mhs_tiles_create_windows(const mhs_grid *g, int64_t out_ncol, int64_t out_nrow,
                                     double feather_d, double *boxes, int64_t *win);
/* number of seams of an nRx x nCx layout: (nCx-1) nRx vertical + nCx (nRx-1) horizontal     */
MHS_API int mhs_seam_count(int64_t nRx, int64_t nCx, int64_t *n_seams);
/* terra::cellFromXY / extract bookkeeping: rows/cols of the cells holding the points (xy is
 * n x 2 column-major); -1 outside the grid.  V73:145,701,910                               */
MHS_API int mhs_cells_from_xy(const mhs_grid *g, const double *xy, int64_t n, int64_t *rows,
                              int64_t *cols);
/* mean mosaic + seam feathering + first-non-NA overlay: Step 3 mosaic and Step 4 of
 * machisplin.mltps (V73:739-747, 760-895; merge_mode = 0) or machisplin.tiles.merge
 * (V73:1392-1546; merge_mode = 1).  tile_dev[h] is a DEVICE buffer holding tile h on its
 * window tile_win[4h..] (row-major, ld = window width); out_dev is the full grid.
 * seam_win_out (may be NULL, 4 int64 per seam in creation order, -1 = no strip) returns the
 * strip windows actually used.  Blocks until done.                                        */
MHS_API int mhs_mosaic_feather_dev(const mhs_grid *g, int64_t nRx, int64_t nCx,
                                   const int64_t *tile_win, const double *const *tile_dev,
                                   int merge_mode, double *out_dev, int64_t ld,
                                   int64_t *seam_win_out, void *stream);
/* same with HOST tiles and a host output grid (what a .Call() shim hands over for
 * machisplin.tiles.merge: each rast.in[[h]] as terra::values in cell order)                 */
MHS_API int mhs_mosaic_feather(const mhs_grid *g, int64_t nRx, int64_t nCx, const int64_t *tile_win,
                               const double *const *tile_host, int merge_mode, double *out_host);
/* terra::extract(r, xy) after mhs_cells_from_xy: gather n cells of a device plane to the
 * host (NaN for row/col -1).  Step-5 station check V73:910                                */
MHS_API int mhs_gather_cells_dev(const double *plane_dev, int64_t ld, const int64_t *rows,
                                 const int64_t *cols, int64_t n, double *out_host, void *stream);

/* Step 3 + Step 4 of machisplin.mltps in one call (V73:636-897): the thin-plate spline of the
 * station residuals over the whole grid.  tile_edge > 0: ceil(nrow/tile_edge) x ceil(ncol/
 * tile_edge) tiles (the reference hard-codes 1500), each fitted on the stations of its +-20 %
 * box (fewer than 10 => zero tile), evaluated on its +-2.5 % box, mean-mosaicked and seam-
 * feathered; tile_edge <= 0 or a single tile: one global fit (V73:748-753).  xy n x 2
 * column-major = the LONG/LAT columns of dat_tps (cell-centre coordinates); cov1_at_stations
 * (may be NULL) = first covariate at each station, NaN rows are dropped as complete.cases does
 * (V73:701-706).  tiles_out (may be NULL) receives nRx, nCx.                               */
MHS_API int mhs_tps_surface(const mhs_grid *g, const double *xy, const double *resid, int64_t n,
                            const double *cov1_at_stations, int64_t tile_edge, double lambda,
                            int gcv_mode, double *out_host, int64_t *tiles_out);
/* A SUBSET of the Step-3 tiles (V73:690-731), for a driver that deals the tiles over several GPUs (SURVEY.md section 8e,
 * reference-tiled mode): tile tile_ids[k] (numbering of mhs_step3_tile_windows, row-major from the south-west) is fitted
 * on the stations of its fit box and evaluated on its keep window into out_dev_ptrs[k] (device, rows x cols of the keep
 * window, contiguous; all zeros below 10 stations, V73:710-721).  The tiles are fitted side by side on the library's
 * lanes; blocks until they are done.  Same arithmetic as mhs_tps_surface (bit-identical planes). */
MHS_API int mhs_tps_tiles_dev(const mhs_grid *g, const double *xy, const double *resid, int64_t n,
                      const double *cov1_at_stations, int64_t tile_edge, double lambda, int gcv_mode,
                      const int64_t *tile_ids, int64_t n_ids, double *const *out_dev_ptrs);
MHS_API int mhs_tps_surface_dev(const mhs_grid *g, const double *xy, const double *resid, int64_t n,
                                const double *cov1_at_stations, int64_t tile_edge, double lambda,
                                int gcv_mode, double *out_dev, int64_t ld, int64_t *tiles_out,
                                void *stream);

/* ------------------------------------------------------------ several devices --
 * ONE host process drives 1..16 devices ("mhs_init(int n_devices) ... library may use internal host
 * threads + HIP streams (one per GPU)"; the reference's host is a single-threaded R session, V73:117).  The reference
 * side: machisplin.tiles.create -> machisplin.mltps per tile -> machisplin.tiles.merge (README.md:157-215,
 * V73:1165-1256, 1392-1548) and the cell-wise independence of machisplin.mltps Steps 2-5 (V73:442-930).
 *
 * mhs_init_devices(n, ids) brings up device SLOTS 0..n-1 on the physical devices ids[k] (NULL = 0..n-1).  Ids may
 * repeat: several slots on one GPU run every code path below on a one-GPU box (the collective then degrades to device
 * copies, because RCCL refuses two ranks on one device).  Every entry point declared ABOVE this section keeps working on
 * slot 0 (mhs_init(d) == mhs_init_devices(1, &d)); model and spline handles made there are replicated onto the other
 * slots by the calls below, on first use, and the replicas are freed with the handle.                               */
MHS_API int mhs_init_devices(int n_devices, const int *device_ids);
MHS_API int mhs_device_slots(int *n_slots, int *device_ids /* may be NULL; room for 16 */);

/* covariate planes cut into row bands, band k resident on slot k (cuts at multiples of 16 rows) */
typedef struct mhs_multi_stack mhs_multi_stack;
/* slot0_share: the share of the rows slot 0 takes -- it also carries the spline fit; NaN = equal bands.
 * THREADING: the calls on a resident stack (mhs_multi_stack_create / _free, mhs_mltps_grid_multi_dev,
 * mhs_multi_final_download) share the slots' streams, events and host-thread team: issue them from ONE host thread at a
 * time (R's single main thread does; the two host-plane calls mhs_mltps_grid_multi / mhs_tiles_units_multi serialise
 * themselves).  A stack belongs to the devices its slots were bound to when it was created: after another
 * mhs_init_devices it is refused (MHS_ERR_INVALID) and mhs_multi_stack_free only drops the handle. */
MHS_API int mhs_multi_stack_create(const mhs_grid *g, const mhs_stack *covars_host, double slot0_share,
                                   mhs_multi_stack **out);
MHS_API int mhs_multi_stack_free(mhs_multi_stack *ms);
/* host only: the bands mhs_multi_stack_create cuts (r0 / r1: n_slots entries).  Every cut is a multiple of 16 rows; chunks of
 * `band` rows, slot 0's rows parked `lead` rows into its chunk, tile the grid (the in-place all-gather's layout).          */
MHS_API int mhs_plan_row_bands(int64_t nrow, int n_slots, double slot0_share, int64_t *r0, int64_t *r1, int64_t *band,
                               int64_t *lead);
MHS_API int mhs_multi_stack_bands(const mhs_multi_stack *ms, int *n_slots, int64_t *r0 /* 16 */, int64_t *r1 /* 16 */);

typedef struct mhs_mltps_info {
    double rsq_model, rsq_final;       /* rsq.model / rsq.final (V73:917-925) */
    double lambda;                     /* the global fit's lambda (NaN with the reference-tiled Step 3) */
    int64_t n_knots;                   /* ... and its unique knots */
    int64_t tiles_rows, tiles_cols;    /* Step-3 tile layout (1 x 1 = global fit) */
    int32_t used_tps;                  /* 1: final = pred.elev + final.TPS; 0: pred.elev alone (V73:925-930) */
    int32_t n_slots;
    int32_t collective;                /* 0 none (host output), 1 RCCL all-gather, 2 peer copies (aliased slots / no librccl) */
    int32_t reserved_;
    int64_t band_r0[16], band_r1[16];  /* rows of every slot */
    double band_ms[16];                /* device time of every slot's ensemble band (HIP events) */
    double tiles_ms[16];               /* reference-tiled Step 3: wall time of every slot's tile fits + evaluations */
    double fit_ms;                     /* wall time of the global spline fit on slot 0 */
    double step_ms;                    /* wall time of the whole call */
    double upload_ms, download_ms;     /* mhs_mltps_grid_multi only: host <-> device */
    double suggested_slot0_share;      /* the slot0_share that would have balanced THIS step (NaN if undetermined) */
    int64_t tiles_pulled_bytes[16];    /* reference-tiled Step 3: bytes of tile planes every slot pulled from its peers (the tiles that
                                        * reach its rows and are owned elsewhere) */
    int32_t tiles_owned[16];           /* ... and the tiles it fitted and evaluated itself (a tile goes to the slot whose band holds
                                        * most of its keep window) */
} mhs_mltps_info;

/* machisplin.mltps Steps 2-5 for one response layer (V73:442-930) over the device slots: ensemble on every band;
 * res.FINAL at the stations and the fields::Tps fit on slot 0 (tile_edge <= 0 or one tile: the global fit of V73:748-753;
 * otherwise the reference's tiles, V73:636-747, each fitted and evaluated by the slot whose band holds most of it; a slot pulls
 * the tiles that reach its rows from their owners and mosaics + feathers its rows only); global fit: every slot evaluates the
 * spline on its rows with the WHOLE grid's evaluation plan; then sums, reads its stations' cells; rsq.final > rsq.model selects the sum (V73:925).
 * X is the n x p station table dat_tps (column-major: covariates, LONG, LAT -- cell-centre coordinates), complete cases
 * only (V73:154).  gather != 0: ONE all-gather (RCCL over xGMI) stitches the final plane on every device
 * (mhs_multi_final_dev).  The N-slot planes equal the one-slot planes bit for bit.                                   */
MHS_API int mhs_mltps_grid_multi_dev(const mhs_model *const *models, const double *weights, int n_models,
                                     double wt_total, mhs_multi_stack *ms, const double *X, const double *resp,
                                     int64_t n, int64_t tile_edge, double lambda, int gcv_mode, int gather,
                                     mhs_mltps_info *info /* may be NULL */);
/* the last step's final plane: every slot's rows down its own PCIe link into final_host (nrow x ncol, row-major) */
MHS_API int mhs_multi_final_download(const mhs_multi_stack *ms, double *final_host);
/* ... or its device pointers on one slot: the slot's rows (band_dev, ld = ncol) and, after gather, the whole grid */
MHS_API int mhs_multi_final_dev(const mhs_multi_stack *ms, int slot, double **band_dev, int64_t *r0, int64_t *r1,
                                double **full_dev);
/* Releases what the two host-plane calls (mhs_mltps_grid_multi, mhs_tiles_units_multi) keep between calls -- device band
 * buffers, unit arenas, pinned rings -- without touching the slots; the next call builds them again.  (mhs_shutdown and
 * mhs_init_devices with other devices do the same.)                                                                      */
MHS_API int mhs_multi_trim(void);
/* Host planes in, host plane out, ONE call: what the R shim binds in place of V73:447-930 for a layer.  The copies are
 * inside the call and pipelined with it: every slot's band comes up in sub-bands under its first members, finished
 * sub-bands of the sum go down under its last one (the plane equals the resident call's bit for bit).  The device buffers
 * are kept for the next call of the same shape (released by mhs_shutdown / a call of another shape); calls from several
 * host threads are served one after the other.  covars_host and final_host may be pageable memory and must stay valid
 * until the call returns (also when it fails: no copy is left in flight).  slot0_share NaN = automatic (equal bands,
 * then what the last call of the same shape measured).  info->upload_ms: time the slowest slot's copy thread spent in
 * its copies up (most of it under kernels); info->download_ms: what was left of the copies down after the last kernel. */
MHS_API int mhs_mltps_grid_multi(const mhs_model *const *models, const double *weights, int n_models, double wt_total,
                                 const mhs_grid *g, const mhs_stack *covars_host, const double *X, const double *resp,
                                 int64_t n, int64_t tile_edge, double lambda, int gcv_mode, double slot0_share,
                                 double *final_host, mhs_mltps_info *info /* may be NULL */);

/* One (tile, layer) run of machisplin.tiles.* (README.md:157-215): the fitted members of that tile and layer and the
 * tile's station table (complete cases; X n x p column-major with cell-centre LONG / LAT of the TILE's raster).        */
typedef struct mhs_unit {
    const mhs_model *const *models;
    const double *weights;
    int32_t n_models, reserved_;
    double wt_total;
    const double *X, *resp;
    int64_t n;
} mhs_unit;
typedef struct mhs_units_info {
    int32_t n_slots, reserved_;
    int64_t n_units;
    double step_ms, unit_ms_sum, unit_ms_max;
    double slot_ms[16];                /* sum of the unit times per slot */
} mhs_units_info;
/* machisplin.tiles.create (out_ncol x out_nrow tiles, feather_d pixels of overlap, V73:1165-1256) -> machisplin.mltps
 * Steps 2-5 per tile and response layer -> machisplin.tiles.merge per layer (V73:1392-1548), over the device slots:
 * unit u = layer * n_tiles + tile (tiles row-major from the south-west) runs on slot u mod N with no exchange; layer l is
 * merged on slot l mod N (its tiles arrive over xGMI) and lands in merged_host[l] (nrow x ncol; a NULL entry skips the
 * layer's merge).  units[u] as above; tps = 0 returns pred.elev alone (V73:934-953); rsq (may be NULL) receives
 * rsq.model, rsq.final per unit.  A tile's crop of the covariate planes is uploaded once per slot that works on it and
 * stays resident for the call; a layer is merged and written to merged_host[l] as soon as its last tile is final, by a
 * helper thread of its slot, while the following layers' units run.  Every slot keeps ONE device arena (crops, scratch, unit
 * planes, merge buffers: 14 GB for 4 x 12 units on a 10 000 x 10 000 grid) and a 64 MB pinned ring between calls; both are
 * released by mhs_shutdown.  Calls from several host threads are served one after the other.                             */
MHS_API int mhs_tiles_units_multi(const mhs_grid *g, const mhs_stack *covars_host, int64_t out_ncol, int64_t out_nrow,
                                  double feather_d, int n_layers, const mhs_unit *units, int tps, int64_t tile_edge,
                                  double lambda, int gcv_mode, double *const *merged_host, double *rsq,
                                  mhs_units_info *info /* may be NULL */);

/* ------------------------------------------------------------- raster wire formats --
 * the formats on either side of the path.  Covariates arrive
 * as GeoTIFF (terra::rast(); the bundled rasters are INT2S, LZW/deflate, NoData -32768,
 * georeferenced by tags or .tfw sidecars: the .tif.ovr and .tfw files of inst/extdata) and results leave
 * as FLT4S GeoTIFF, terra::writeRaster's default (machisplin.write.geotiff, V73:1011,1020).
 * Reader: classic TIFF / BigTIFF, either byte order, strips or tiles, none / LZW / deflate,
 * predictor 1-3, one band.  Host functions need no GPU.                                   */
typedef struct mhs_tiff_info {
    int64_t width, height;
    int32_t bits, sample_format;   /* TIFF SampleFormat: 1 unsigned, 2 signed, 3 IEEE float       */
    int32_t compression, n_ifd;    /* TIFF Compression tag; image directories (overview levels)    */
    int32_t dtype;                 /* MHS_I16 / MHS_F32 / MHS_F64, or -1 if no device plane type   */
    int32_t has_geo;               /* ModelPixelScale + ModelTiepoint present                      */
    double nodata;                 /* GDAL_NODATA, NaN if absent                                   */
    double xmin, ymax, xres, yres; /* north-west corner and cell size when has_geo                 */
} mhs_tiff_info;
MHS_API int mhs_tiff_info_read(const char *path, int ifd, mhs_tiff_info *out);
/* decode image directory `ifd` into a host buffer of the file's native sample type, row-major */
MHS_API int mhs_tiff_read_host(const char *path, int ifd, void *out, int64_t out_bytes);
/* decode on host threads and stream into a DEVICE plane (element type = info.dtype, ld in
 * elements): bands of ~32 MB, band k+1 is decoded while band k is in flight over PCIe         */
MHS_API int mhs_tiff_read_dev(const char *path, int ifd, void *out_dev, int64_t ld_elems, void *stream);
/* terra::writeRaster(x, "<layer>.tif") for a double raster: float32 strips, compression 1 (none)
 * or 8 (deflate); NaN cells are written as `nodata` when it is not NaN (and the GDAL_NODATA tag
 * is set), else as NaN.  Geo tags: pixel scale, tiepoint, EPSG:4326 (V73:164,775).              */
MHS_API int mhs_tiff_write_f32_host(const char *path, const mhs_grid *g, const float *data, double nodata,
                                    int compression);
MHS_API int mhs_tiff_write_f32_dev(const char *path, const mhs_grid *g, const double *plane_dev, int64_t ld,
                                   double nodata, int compression, void *stream);
/* ESRI world file (.tfw): six numbers -- xres, rot, rot, -yres, x centre and y centre of the NW cell */
MHS_API int mhs_tfw_read(const char *path, double *six);

#ifdef __cplusplus
}
#endif
#endif /* MACHISPLIN_HIP_H */
