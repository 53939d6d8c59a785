/* .Call() shim: binds R to the C ABI of include/machisplin_hip.h.  UNTESTED HERE (no R in the
 * build image); it is mechanically derived from the header: every SEXP is unpacked with
 * REAL()/INTEGER(), the library is called, a non-zero status becomes Rf_error() AFTER all C
 * resources are released, handles are external pointers with finalizers.
 *
 * Build inside the R package:  PKG_CPPFLAGS=-I<repo>/include  PKG_LIBS=-L<repo>/machisplin_amd -lmachisplin_hip
 */
#include <R.h>
#include <Rinternals.h>
#include <stdint.h>
#include "machisplin_hip.h"

static void chk(int rc) { if (rc != MHS_OK) Rf_error("machisplin_hip: %s", mhs_last_error()); }

static mhs_grid grid_from(SEXP geom) { /* c(xmin, ymax, xres, yres, nrow, ncol) */
    double *g = REAL(geom);
    mhs_grid out = { g[0], g[1], g[2], g[3], (int64_t)g[4], (int64_t)g[5] };
    return out;
}

static void tps_finalizer(SEXP p) { mhs_tps_free((mhs_tps *)R_ExternalPtrAddr(p)); R_ClearExternalPtr(p); }
static void model_finalizer(SEXP p) { mhs_model_free((mhs_model *)R_ExternalPtrAddr(p)); R_ClearExternalPtr(p); }
static SEXP wrap(void *h, R_CFinalizer_t fin) {
    SEXP p = PROTECT(R_MakeExternalPtr(h, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(p, fin, TRUE);
    UNPROTECT(1);
    return p;
}

SEXP mhsr_init(SEXP device) { chk(mhs_init(Rf_asInteger(device))); return R_NilValue; }
/* 0 = by cost (default), 1 = direct sum (predict.Krig's own loop; bit-identical across windows), 2 = far-field-
   interpolated sum; see mhs_tps_eval_mode in machisplin_hip.h */
SEXP mhsr_tps_eval_mode(SEXP mode) { chk(mhs_tps_eval_mode(Rf_asInteger(mode))); return R_NilValue; }

/* fields::Tps(x, Y)  (V73:722, V73:751).  xy: n x 2 numeric matrix, lambda NA => GCV */
SEXP mhsr_tps_fit(SEXP xy, SEXP y, SEXP lambda, SEXP mode) {
    mhs_tps *t = NULL;
    double lam = Rf_asReal(lambda);
    chk(mhs_tps_fit(REAL(xy), REAL(y), (int64_t)Rf_length(y), ISNA(lam) ? R_NaN : lam, Rf_asInteger(mode), &t));
    return wrap(t, tps_finalizer);
}

/* lapply(seq_along(xs), function(k) fields::Tps(xs[[k]], ys[[k]])) in ONE library call (the tiles of V73:690-738, or the
 * response layers of one station table): xs a list of n_k x 2 numeric matrices, ys a list of numeric vectors.  Every fit with
 * 8..256 distinct locations is done by one kernel launch, a workgroup per fit (mhs_tps_fit_many).  Returns a list of external
 * pointers, NULL where a fit failed (collinear stations, ...). */
SEXP mhsr_tps_fit_many(SEXP xs, SEXP ys, SEXP lambda, SEXP mode) {
    const R_xlen_t k = Rf_xlength(xs);
    double lam = Rf_asReal(lambda);
    const double **px = (const double **)R_alloc((size_t)(k ? k : 1), sizeof(double *));
    const double **py = (const double **)R_alloc((size_t)(k ? k : 1), sizeof(double *));
    int64_t *n = (int64_t *)R_alloc((size_t)(k ? k : 1), sizeof(int64_t));
    mhs_tps **h = (mhs_tps **)R_alloc((size_t)(k ? k : 1), sizeof(mhs_tps *));
    int *st = (int *)R_alloc((size_t)(k ? k : 1), sizeof(int));
    if (Rf_xlength(ys) != k) Rf_error("mhsr_tps_fit_many: xs and ys have different lengths");
    for (R_xlen_t i = 0; i < k; ++i) {
        SEXP x = VECTOR_ELT(xs, i), y = VECTOR_ELT(ys, i);
        if (Rf_xlength(x) != 2 * Rf_xlength(y)) Rf_error("mhsr_tps_fit_many: element %d: x must be an n x 2 matrix beside n values", (int)i + 1);
        px[i] = REAL(x); py[i] = REAL(y); n[i] = (int64_t)Rf_xlength(y); h[i] = NULL;
    }
    SEXP out = PROTECT(Rf_allocVector(VECSXP, k));
    int rc = mhs_tps_fit_many(px, py, n, (int64_t)k, ISNA(lam) ? R_NaN : lam, Rf_asInteger(mode), h, st);
    /* every handle gets its owner BEFORE an error can unwind: nothing leaks */
    for (R_xlen_t i = 0; i < k; ++i)
        if (h[i]) SET_VECTOR_ELT(out, i, wrap(h[i], tps_finalizer));
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* predict(tps, xy): the spline at arbitrary points -- what terra::interpolate feeds an S3 predict method
 * block by block; xy is an n x 2 numeric matrix */
SEXP mhsr_tps_predict_points(SEXP tps, SEXP xy) {
    R_xlen_t n = Rf_xlength(xy) / 2;
    SEXP out = PROTECT(Rf_allocVector(REALSXP, n));
    int rc = mhs_tps_predict_points((mhs_tps *)R_ExternalPtrAddr(tps), REAL(xy), (int64_t)n, REAL(out));
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* terra::interpolate(terra::rast(rb), tps)  (V73:726, V73:753): values in terra cell order */
SEXP mhsr_tps_predict_grid(SEXP tps, SEXP geom, SEXP win) {
    mhs_grid g = grid_from(geom);
    int *w = INTEGER(win); /* r0, r1, c0, c1 (0-based, half-open) */
    SEXP out = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)(w[1] - w[0]) * (w[3] - w[2])));
    int rc = mhs_tps_predict_grid((mhs_tps *)R_ExternalPtrAddr(tps), &g, w[0], w[1], w[2], w[3], REAL(out));
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* Step 3 + Step 4 in one call (V73:636-897) */
SEXP mhsr_tps_surface(SEXP geom, SEXP xy, SEXP resid, SEXP cov1, SEXP tile_edge, SEXP lambda, SEXP mode) {
    mhs_grid g = grid_from(geom);
    double lam = Rf_asReal(lambda);
    SEXP out = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)g.nrow * g.ncol));
    int rc = mhs_tps_surface(&g, REAL(xy), REAL(resid), (int64_t)Rf_length(resid),
                             Rf_isNull(cov1) ? NULL : REAL(cov1), (int64_t)Rf_asInteger(tile_edge),
                             ISNA(lam) ? R_NaN : lam, Rf_asInteger(mode), REAL(out), NULL);
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* model loaders: flat arrays pulled out of the fitted objects by R/backend_hip.R */
SEXP mhsr_lm_load(SEXP coef) {
    mhs_model *m = NULL;
    chk(mhs_lm_load(REAL(coef), Rf_length(coef) - 1, &m));
    return wrap(m, model_finalizer);
}
SEXP mhsr_nnet_load(SEXP wts, SEXP p, SEXP size, SEXP scale, SEXP shift) {
    mhs_model *m = NULL;
    chk(mhs_nnet_load(REAL(wts), Rf_asInteger(p), Rf_asInteger(size), Rf_asReal(scale), Rf_asReal(shift), &m));
    return wrap(m, model_finalizer);
}
SEXP mhsr_earth_load(SEXP coef, SEXP dirs_rowmajor, SEXP cuts_rowmajor, SEXP p) {
    mhs_model *m = NULL;
    chk(mhs_earth_load(REAL(coef), INTEGER(dirs_rowmajor), REAL(cuts_rowmajor), Rf_length(coef), Rf_asInteger(p), &m));
    return wrap(m, model_finalizer);
}
SEXP mhsr_svr_load(SEXP alpha, SEXP sv_rowmajor, SEXP p, SEXP b, SEXP sigma, SEXP xc, SEXP xs, SEXP yc, SEXP ys) {
    mhs_model *m = NULL;
    chk(mhs_svr_load(REAL(alpha), REAL(sv_rowmajor), (int64_t)Rf_length(alpha), Rf_asInteger(p), Rf_asReal(b),
                     Rf_asReal(sigma), REAL(xc), REAL(xs), Rf_asReal(yc), Rf_asReal(ys), &m));
    return wrap(m, model_finalizer);
}
SEXP mhsr_gbm_load(SEXP initF, SEXP offsets /* numeric, n.trees+1 */, SEXP var, SEXP val, SEXP left, SEXP right,
                   SEXP missing, SEXP p) {
    R_xlen_t nt = Rf_xlength(offsets) - 1;
    int64_t *off = (int64_t *)R_alloc((size_t)nt + 1, sizeof(int64_t));
    for (R_xlen_t i = 0; i <= nt; ++i) off[i] = (int64_t)REAL(offsets)[i];
    mhs_model *m = NULL;
    chk(mhs_gbm_load(Rf_asReal(initF), (int64_t)nt, off, INTEGER(var), REAL(val), INTEGER(left), INTEGER(right),
                     INTEGER(missing), Rf_asInteger(p), &m));
    return wrap(m, model_finalizer);
}
SEXP mhsr_rf_load(SEXP offsets, SEXP left, SEXP right, SEXP status, SEXP bestvar, SEXP split, SEXP nodepred, SEXP p) {
    R_xlen_t nt = Rf_xlength(offsets) - 1;
    int64_t *off = (int64_t *)R_alloc((size_t)nt + 1, sizeof(int64_t));
    for (R_xlen_t i = 0; i <= nt; ++i) off[i] = (int64_t)REAL(offsets)[i];
    mhs_model *m = NULL;
    chk(mhs_rf_load((int64_t)nt, off, INTEGER(left), INTEGER(right), INTEGER(status), INTEGER(bestvar), REAL(split),
                    REAL(nodepred), Rf_asInteger(p), &m));
    return wrap(m, model_finalizer);
}

/* the Step-2 raster loop (V73:447-619).  covars: terra::values(covar.ras), an ncell x C numeric
 * matrix -- column-major, i.e. already planar; NA_real_ is a NaN and is treated as NA */
SEXP mhsr_ensemble_predict(SEXP models, SEXP weights, SEXP wt_total, SEXP geom, SEXP covars) {
    mhs_grid g = grid_from(geom);
    int n = Rf_length(models);
    const mhs_model **h = (const mhs_model **)R_alloc((size_t)n, sizeof(*h));
    for (int i = 0; i < n; ++i) h[i] = (const mhs_model *)R_ExternalPtrAddr(VECTOR_ELT(models, i));
    mhs_stack st = { REAL(covars), Rf_ncols(covars), MHS_F64, (int64_t)g.nrow * g.ncol, g.ncol, R_NaN };
    SEXP out = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)g.nrow * g.ncol));
    int rc = mhs_ensemble_predict(h, REAL(weights), n, Rf_asReal(wt_total), &g, &st, 0, g.nrow, 0, g.ncol, REAL(out));
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* predict(model, data.frame) at the stations (V73:477, 501, 525, 586, 608) */
SEXP mhsr_predict_points(SEXP model, SEXP X) {
    SEXP out = PROTECT(Rf_allocVector(REALSXP, Rf_nrows(X)));
    int rc = mhs_predict_points((const mhs_model *)R_ExternalPtrAddr(model), REAL(X), (int64_t)Rf_nrows(X), REAL(out));
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* mgcv::gam(mod.form, data) with the parametric formula (V73:252, V73:600): X = as.matrix(data[, predictors]) */
SEXP mhsr_lm_fit(SEXP X, SEXP y) {
    SEXP out = PROTECT(Rf_allocVector(REALSXP, Rf_ncols(X) + 1));
    int rc = mhs_lm_fit(REAL(X), REAL(y), (int64_t)Rf_nrows(X), Rf_ncols(X), REAL(out));
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* kernlab::ksvm(mod.form, data) (V73:251, V73:560) with kpar = list(sigma = sigma): returns list(beta[n], b,
 * x.center[p], x.scale[p], y.center, y.scale, iterations); the support vectors are the rows with beta != 0 */
SEXP mhsr_svr_fit(SEXP X, SEXP y, SEXP sigma, SEXP C, SEXP epsilon, SEXP tol) {
    if (!Rf_isReal(X) || !Rf_isMatrix(X) || !Rf_isReal(y)) Rf_error("mhsr_svr_fit: X must be a double matrix and y a double vector");
    int n = Rf_nrows(X), p = Rf_ncols(X);
    if (Rf_length(y) != n) Rf_error("mhsr_svr_fit: length(y) = %d but X has %d rows", Rf_length(y), n);
    if (p < 1 || p > 12) Rf_error("mhsr_svr_fit: 1..12 predictors supported, got %d", p);
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 7));
    SEXP beta = PROTECT(Rf_allocVector(REALSXP, n)), xc = PROTECT(Rf_allocVector(REALSXP, p)), xs = PROTECT(Rf_allocVector(REALSXP, p));
    double b = 0, yc = 0, ys = 0;
    int64_t it = 0;
    int rc = mhs_svr_fit(REAL(X), REAL(y), (int64_t)n, p, Rf_asReal(sigma), Rf_asReal(C), Rf_asReal(epsilon), Rf_asReal(tol), 0,
                         REAL(beta), &b, REAL(xc), REAL(xs), &yc, &ys, &it);
    if (rc == 0) {
        SET_VECTOR_ELT(out, 0, beta); SET_VECTOR_ELT(out, 1, Rf_ScalarReal(b)); SET_VECTOR_ELT(out, 2, xc); SET_VECTOR_ELT(out, 3, xs);
        SET_VECTOR_ELT(out, 4, Rf_ScalarReal(yc)); SET_VECTOR_ELT(out, 5, Rf_ScalarReal(ys)); SET_VECTOR_ELT(out, 6, Rf_ScalarReal((double)it));
    }
    UNPROTECT(4);
    chk(rc);
    return out;
}

/* nnet::nnet(mod.form, data = trainNN, size = 10, linout = TRUE, maxit = maxit) (V73:249, V73:463) from the initial
 * weights wts0 (nnet's own: runif(length, -0.7, 0.7)); y already scaled as V73:455-459.
 * returns list(wts, value, counts[2], fail) */
SEXP mhsr_nnet_fit(SEXP X, SEXP y, SEXP wts0, SEXP maxit) {
    if (!Rf_isReal(X) || !Rf_isMatrix(X) || !Rf_isReal(y) || !Rf_isReal(wts0)) Rf_error("mhsr_nnet_fit: X must be a double matrix, y and wts0 double vectors");
    int nw = Rf_length(wts0);
    if (Rf_ncols(X) < 1 || Rf_ncols(X) > 12) Rf_error("mhsr_nnet_fit: 1..12 predictors supported, got %d", Rf_ncols(X));
    if (Rf_length(y) != Rf_nrows(X)) Rf_error("mhsr_nnet_fit: length(y) = %d but X has %d rows", Rf_length(y), Rf_nrows(X));
    if (nw != (Rf_ncols(X) + 1) * 10 + 11) Rf_error("mhsr_nnet_fit: wts0 must hold (ncol(X) + 1) * 10 + 11 = %d weights, got %d", (Rf_ncols(X) + 1) * 10 + 11, nw);
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 4));
    SEXP w = PROTECT(Rf_allocVector(REALSXP, nw)), counts = PROTECT(Rf_allocVector(INTSXP, 2));
    for (int i = 0; i < nw; ++i) REAL(w)[i] = REAL(wts0)[i];
    double value = 0;
    int fail = 0;
    int rc = mhs_nnet_fit(REAL(X), REAL(y), (int64_t)Rf_nrows(X), Rf_ncols(X), 10, REAL(w), Rf_asInteger(maxit), 1e-4, 1e-8,
                          &value, INTEGER(counts), &fail);
    if (rc == 0) {
        SET_VECTOR_ELT(out, 0, w); SET_VECTOR_ELT(out, 1, Rf_ScalarReal(value)); SET_VECTOR_ELT(out, 2, counts);
        SET_VECTOR_ELT(out, 3, Rf_ScalarInteger(fail));
    }
    UNPROTECT(3);
    chk(rc);
    return out;
}

/* gbm::predict.gbm(model, x.data[pred.mask, ], n.trees = k * step) for k = 1 .. in one pass (V73:1843, 1919): returns the
 * n x stages matrix machisplin.gbm.step's hold-out deviance curve is computed from */
SEXP mhsr_gbm_staged_points(SEXP model, SEXP X, SEXP step, SEXP n_trees) {
    const mhs_model *m = (const mhs_model *)R_ExternalPtrAddr(model);
    int kind = -1, p = 0, st = Rf_asInteger(step);
    int64_t trees = 0;
    chk(mhs_model_info(m, &kind, &p, &trees));
    if (kind != 4) Rf_error("mhsr_gbm_staged_points: the handle is not a gbm model");
    if (st == NA_INTEGER || st < 1) Rf_error("mhsr_gbm_staged_points: step must be >= 1");
    if (!Rf_isReal(X) || !Rf_isMatrix(X) || Rf_ncols(X) != p) Rf_error("mhsr_gbm_staged_points: X must be a double matrix with %d columns", p);
    /* the library walks the trees the HANDLE holds and writes n x (trees / step) values: the matrix is sized from the
     * handle, and a caller whose n.trees disagrees with it is told so instead of being overrun */
    if (Rf_asInteger(n_trees) != NA_INTEGER && (int64_t)Rf_asInteger(n_trees) != trees)
        Rf_error("mhsr_gbm_staged_points: n.trees = %d but the loaded model has %lld trees", Rf_asInteger(n_trees), (long long)trees);
    int n = Rf_nrows(X), stages = (int)(trees / st);
    SEXP out = PROTECT(Rf_allocMatrix(REALSXP, n, stages));
    int rc = stages > 0 && n > 0 ? mhs_gbm_staged_points(m, REAL(X), (int64_t)n, st, REAL(out)) : 0;
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* machisplin.tiles.merge (V73:1392-1546): tiles = list of numeric vectors (terra::values of each
 * rast.in[[h]] in cell order), win = integer matrix 4 x n (r0, r1, c0, c1 per tile, 0-based) */
SEXP mhsr_tiles_merge(SEXP geom, SEXP tiles, SEXP win, SEXP in_ncol, SEXP in_nrow) {
    mhs_grid g = grid_from(geom);
    int n = Rf_length(tiles);
    const double **h = (const double **)R_alloc((size_t)n, sizeof(*h));
    int64_t *w = (int64_t *)R_alloc((size_t)4 * n, sizeof(int64_t));
    for (int i = 0; i < n; ++i) h[i] = REAL(VECTOR_ELT(tiles, i));
    for (int i = 0; i < 4 * n; ++i) w[i] = (int64_t)INTEGER(win)[i];
    SEXP out = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)g.nrow * g.ncol));
    int rc = mhs_mosaic_feather(&g, Rf_asInteger(in_nrow), Rf_asInteger(in_ncol), w, h, 1, REAL(out));
    UNPROTECT(1);
    chk(rc);
    return out;
}

/* bracket of the layer loop of machisplin.mltps (V73:176-957): fits of later layers reuse the tiles' reductions */
SEXP mhsr_tps_reduction_cache(SEXP enable) {
    chk(mhs_tps_reduction_cache(Rf_asInteger(enable)));
    return enable;
}

/* ---- several devices from the one R process (include/machisplin_hip.h, "several devices") ---- */

/* mhs_init_devices: n device slots on the devices `ids` (NULL = 0 .. n - 1) */
SEXP mhsr_init_devices(SEXP n, SEXP ids) {
    int nn = Rf_asInteger(n);
    if (nn == NA_INTEGER || nn < 1 || nn > 16) Rf_error("mhsr_init_devices: n must be 1..16");
    if (!Rf_isNull(ids) && Rf_length(ids) != nn) Rf_error("mhsr_init_devices: length(ids) must equal n");
    chk(mhs_init_devices(nn, Rf_isNull(ids) ? NULL : INTEGER(ids)));
    int slots = 0;
    chk(mhs_device_slots(&slots, NULL));
    return Rf_ScalarInteger(slots);
}

/* mhs_multi_trim: give back the device buffers, arenas and pinned rings the two host-plane calls keep between calls */
SEXP mhsr_multi_trim(void) {
    chk(mhs_multi_trim());
    return R_NilValue;
}

static const mhs_model **model_handles(SEXP models, int *n) {
    *n = Rf_length(models);
    const mhs_model **h = (const mhs_model **)R_alloc((size_t)(*n > 0 ? *n : 1), sizeof(*h));
    for (int i = 0; i < *n; ++i) {
        h[i] = (const mhs_model *)R_ExternalPtrAddr(VECTOR_ELT(models, i));
        if (!h[i]) Rf_error("machisplin_hip: a model handle has been freed");
    }
    return h;
}

/* machisplin.mltps Steps 2-5 for one response layer (V73:447-930) over every device slot in ONE call: covars =
 * terra::values(covar.ras) (ncell x C, planar as it stands), X = as.matrix(dat_tps[, predictors]) (n x p: covariates, LONG,
 * LAT), resp the response column.  Returns list(final (terra cell order), rsq.model, rsq.final, lambda, used.tps,
 * n.slots, slot0.share for the next call). */
SEXP mhsr_mltps_grid_multi(SEXP models, SEXP weights, SEXP wt_total, SEXP geom, SEXP covars, SEXP X, SEXP resp, SEXP tile_edge,
                           SEXP lambda, SEXP mode, SEXP slot0_share) {
    mhs_grid g = grid_from(geom);
    int n = 0;
    const mhs_model **h = model_handles(models, &n);
    if (!Rf_isReal(covars) || !Rf_isMatrix(covars) || (int64_t)Rf_nrows(covars) != g.nrow * g.ncol)
        Rf_error("mhsr_mltps_grid_multi: covars must be terra::values(covar.ras), an ncell x layers double matrix");
    if (!Rf_isReal(X) || !Rf_isMatrix(X) || Rf_ncols(X) != Rf_ncols(covars) + 2 || Rf_length(resp) != Rf_nrows(X))
        Rf_error("mhsr_mltps_grid_multi: X must be n x (layers + 2) and resp of length n");
    if (Rf_length(weights) != n) Rf_error("mhsr_mltps_grid_multi: one weight per model");
    mhs_stack st = { REAL(covars), Rf_ncols(covars), MHS_F64, (int64_t)g.nrow * g.ncol, g.ncol, R_NaN };
    double lam = Rf_asReal(lambda), share = Rf_asReal(slot0_share);
    mhs_mltps_info info;
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 7));
    SEXP fin = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)g.nrow * g.ncol));
    int rc = mhs_mltps_grid_multi(h, REAL(weights), n, Rf_asReal(wt_total), &g, &st, REAL(X), REAL(resp), (int64_t)Rf_nrows(X),
                                  (int64_t)Rf_asInteger(tile_edge), ISNA(lam) ? R_NaN : lam, Rf_asInteger(mode),
                                  ISNA(share) ? R_NaN : share, REAL(fin), &info);
    if (rc == 0) {
        SET_VECTOR_ELT(out, 0, fin); SET_VECTOR_ELT(out, 1, Rf_ScalarReal(info.rsq_model)); SET_VECTOR_ELT(out, 2, Rf_ScalarReal(info.rsq_final));
        SET_VECTOR_ELT(out, 3, Rf_ScalarReal(info.lambda)); SET_VECTOR_ELT(out, 4, Rf_ScalarInteger(info.used_tps));
        SET_VECTOR_ELT(out, 5, Rf_ScalarInteger(info.n_slots)); SET_VECTOR_ELT(out, 6, Rf_ScalarReal(info.suggested_slot0_share));
    }
    UNPROTECT(2);
    chk(rc);
    return out;
}

/* machisplin.tiles.create -> machisplin.mltps Steps 2-5 per (tile, layer) -> machisplin.tiles.merge (README.md:157-215,
 * V73:1165-1256, 1392-1548) over every device slot in ONE call.  units: list of length n.layers * n.tiles, layer-major
 * (unit (l - 1) * n.tiles + t), each list(models, weights, wt.total, X, resp) of that tile's fits and station table.
 * Returns list(planes = list over layers of numeric vectors in terra cell order, rsq = 2 x n.units matrix). */
SEXP mhsr_tiles_units_multi(SEXP geom, SEXP covars, SEXP out_ncol, SEXP out_nrow, SEXP feather_d, SEXP units, SEXP n_layers,
                            SEXP tps, SEXP tile_edge, SEXP lambda, SEXP mode) {
    mhs_grid g = grid_from(geom);
    int L = Rf_asInteger(n_layers), nc = Rf_asInteger(out_ncol), nr = Rf_asInteger(out_nrow);
    if (L == NA_INTEGER || L < 1 || nc < 1 || nr < 1) Rf_error("mhsr_tiles_units_multi: bad layout");
    int nu = Rf_length(units);
    if (nu != L * nc * nr) Rf_error("mhsr_tiles_units_multi: need n.layers * out.ncol * out.nrow = %d units, got %d", L * nc * nr, nu);
    if (!Rf_isReal(covars) || !Rf_isMatrix(covars) || (int64_t)Rf_nrows(covars) != g.nrow * g.ncol)
        Rf_error("mhsr_tiles_units_multi: covars must be terra::values(covar.ras), an ncell x layers double matrix");
    mhs_unit *u = (mhs_unit *)R_alloc((size_t)nu, sizeof(*u));
    for (int i = 0; i < nu; ++i) {
        SEXP e = VECTOR_ELT(units, i);
        if (TYPEOF(e) != VECSXP || Rf_length(e) != 5) Rf_error("mhsr_tiles_units_multi: unit %d must be list(models, weights, wt.total, X, resp)", i + 1);
        int nm = 0;
        u[i].models = model_handles(VECTOR_ELT(e, 0), &nm);
        u[i].n_models = nm; u[i].reserved_ = 0;
        if (Rf_length(VECTOR_ELT(e, 1)) != nm) Rf_error("mhsr_tiles_units_multi: unit %d: one weight per model", i + 1);
        u[i].weights = REAL(VECTOR_ELT(e, 1));
        u[i].wt_total = Rf_asReal(VECTOR_ELT(e, 2));
        SEXP X = VECTOR_ELT(e, 3), y = VECTOR_ELT(e, 4);
        if (!Rf_isReal(X) || !Rf_isMatrix(X) || Rf_ncols(X) != Rf_ncols(covars) + 2 || Rf_length(y) != Rf_nrows(X))
            Rf_error("mhsr_tiles_units_multi: unit %d: X must be n x (layers + 2) and resp of length n", i + 1);
        u[i].X = REAL(X); u[i].resp = REAL(y); u[i].n = (int64_t)Rf_nrows(X);
    }
    mhs_stack st = { REAL(covars), Rf_ncols(covars), MHS_F64, (int64_t)g.nrow * g.ncol, g.ncol, R_NaN };
    double lam = Rf_asReal(lambda);
    SEXP out = PROTECT(Rf_allocVector(VECSXP, 2));
    SEXP planes = PROTECT(Rf_allocVector(VECSXP, L));
    SEXP rsq = PROTECT(Rf_allocMatrix(REALSXP, 2, nu));
    double **ptr = (double **)R_alloc((size_t)L, sizeof(*ptr));
    for (int l = 0; l < L; ++l) {
        SEXP v = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)g.nrow * g.ncol));
        SET_VECTOR_ELT(planes, l, v);
        UNPROTECT(1);
        ptr[l] = REAL(v);
    }
    int rc = mhs_tiles_units_multi(&g, &st, nc, nr, Rf_asReal(feather_d), L, u, Rf_asInteger(tps), (int64_t)Rf_asInteger(tile_edge),
                                   ISNA(lam) ? R_NaN : lam, Rf_asInteger(mode), ptr, REAL(rsq), NULL);
    if (rc == 0) { SET_VECTOR_ELT(out, 0, planes); SET_VECTOR_ELT(out, 1, rsq); }
    UNPROTECT(3);
    chk(rc);
    return out;
}

/* library.dynam.unload / R exit: the library's streams go while the HIP runtime is still whole (mhs_init also registers
 * an atexit handler, so this is belt and braces) */
void R_unload_machisplin_hip(DllInfo *info) {
    (void)info;
    (void)mhs_shutdown();
}
