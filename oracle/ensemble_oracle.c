/* Oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py): plain-C restatement of the
 * per-cell arithmetic behind terra::predict(rast_stack, model) for the six ensemble members
 * (V73:468-606) and of the weighted accumulation (V73:471..605, 619).  PARITY UNPINNED vs R:
 * gbm / randomForest / nnet / earth / kernlab / mgcv are un-vendored CRAN packages; the loops
 * below follow their published predict algorithms (gbm_pred, regForest/predictRegTree,
 * VR_nntest with nnet.c's clamped sigmoid, earth's bx %*% beta, kernlab's rbf kernelMult,
 * a linear model) exactly as oracle/ensemble.py does, one cell at a time as the reference's
 * CPU path does.  Used as the fast checker and by bench.py's cpu_baseline leg.
 *
 * X: cells x p row-major, predictors in rast_stack order (covariates, LONG, LAT); NaN = NA.
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int any_nan(const double *x, int p) {
    for (int j = 0; j < p; ++j) if (isnan(x[j])) return 1;
    return 0;
}

void oracle_predict_lm(const double *coef, int p, const double *X, int64_t n, int threads, double *out) {
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < n; ++i) {
        double acc = coef[0];
        for (int j = 0; j < p; ++j) acc = acc + coef[j + 1] * X[i * p + j];
        out[i] = acc;
    }
}

static double nnet_sigmoid(double z) {
    if (z < -15.0) return 0.0;
    if (z > 15.0) return 1.0;
    return 1.0 / (1.0 + exp(-z));
}

void oracle_predict_nnet(const double *w, int p, int H, double y_scale, double y_shift, const double *X,
                         int64_t n, int threads, double *out) {
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < n; ++i) {
        const double *x = X + i * p;
        if (any_nan(x, p)) { out[i] = NAN; continue; }
        double acc = w[(p + 1) * H];
        for (int h = 0; h < H; ++h) {
            const double *wh = w + h * (p + 1);
            double z = wh[0];
            for (int j = 0; j < p; ++j) z = z + wh[1 + j] * x[j];
            acc = acc + w[(p + 1) * H + 1 + h] * nnet_sigmoid(z);
        }
        out[i] = acc * y_scale + y_shift;
    }
}

void oracle_predict_earth(const double *coef, const int32_t *dirs, const double *cuts, int nterms, int p,
                          const double *X, int64_t n, int threads, double *out) {
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < n; ++i) {
        const double *x = X + i * p;
        if (any_nan(x, p)) { out[i] = NAN; continue; }
        double acc = 0.0;
        for (int k = 0; k < nterms; ++k) {
            double term = 1.0;
            for (int v = 0; v < p; ++v) {
                const int d = dirs[k * p + v];
                if (d == 0) continue;
                const double c = cuts[k * p + v];
                term = term * (d == 2 ? x[v] : (d == 1 ? fmax(0.0, x[v] - c) : fmax(0.0, c - x[v])));
            }
            acc = acc + coef[k] * term;
        }
        out[i] = acc;
    }
}

void oracle_predict_svr(const double *alpha, const double *sv, int64_t nsv, int p, double b, double sigma,
                        const double *xc, const double *xs, double yc, double ys, const double *X, int64_t n,
                        int threads, double *out) {
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t i = 0; i < n; ++i) {
        const double *x = X + i * p;
        if (any_nan(x, p)) { out[i] = NAN; continue; }
        double xt[64];
        for (int j = 0; j < p; ++j) xt[j] = (x[j] - xc[j]) / xs[j];
        double acc = 0.0;
        for (int64_t v = 0; v < nsv; ++v) {
            double d2 = 0.0;
            for (int j = 0; j < p; ++j) { const double d = xt[j] - sv[v * p + j]; d2 += d * d; }
            acc += alpha[v] * exp(-sigma * d2);
        }
        out[i] = (acc - b) * ys + yc;
    }
}

void oracle_predict_gbm(double init_f, int64_t n_trees, const int64_t *off, const int32_t *var, const double *val,
                        const int32_t *left, const int32_t *right, const int32_t *missing, int p, const double *X,
                        int64_t n, int threads, double *out, int64_t *visits) {
    int64_t nv = 0;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) reduction(+ : nv)
    for (int64_t i = 0; i < n; ++i) {
        const double *x = X + i * p;
        double acc = init_f;
        for (int64_t t = 0; t < n_trees; ++t) {
            const int64_t o = off[t];
            int64_t k = 0;
            while (var[o + k] >= 0) {
                const double xv = x[var[o + k]];
                k = isnan(xv) ? missing[o + k] : (xv < val[o + k] ? left[o + k] : right[o + k]);
                ++nv;
            }
            acc = acc + val[o + k];
        }
        out[i] = acc;
    }
    if (visits) *visits = nv;
}

void oracle_predict_rf(int64_t n_trees, const int64_t *off, const int32_t *left, const int32_t *right,
                       const int32_t *status, const int32_t *best_var, const double *split, const double *node_pred,
                       int p, const double *X, int64_t n, int threads, double *out, int64_t *visits) {
    int64_t nv = 0;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1) reduction(+ : nv)
    for (int64_t i = 0; i < n; ++i) {
        const double *x = X + i * p;
        if (any_nan(x, p)) { out[i] = NAN; continue; }
        double acc = 0.0;
        for (int64_t t = 0; t < n_trees; ++t) {
            const int64_t o = off[t];
            int64_t k = 0;
            while (status[o + k] != -1) {
                k = (x[best_var[o + k] - 1] <= split[o + k] ? left[o + k] : right[o + k]) - 1;
                ++nv;
            }
            acc = acc + node_pred[o + k];
        }
        out[i] = acc / (double)n_trees;
    }
    if (visits) *visits = nv;
}
