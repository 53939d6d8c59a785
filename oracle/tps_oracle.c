/* Oracle (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py): plain-C restatement of the
 * CPU arithmetic the reference executes for the TPS hot loops.  PARITY UNPINNED vs R.
 *
 *  - oracle_tps_eval_grid: predict.Krig as reached from
 *        terra::interpolate(terra::rast(rb), mod.tps.elev)      V73:726, V73:753
 *    i.e. fields.mkpoly(x,2) %*% d + Rad.cov(x, knots, C = c): for every cell centre a
 *    direct loop over the knots (fields' multebC / radfun, never forming the matrix),
 *    radfun(d2) = 0.5*log(d2)*d2 with d2 floored at 1e-20, times radbas.constant = 1/(8 pi).
 *  - oracle_tps_gram: Rad.cov(x, x) as Krig.engine.default assembles it for fields::Tps
 *        fields::Tps(...)                                         V73:722, V73:751
 *
 * Used by tests/ as the checker at sizes numpy is too slow for, and by bench.py's
 * cpu_baseline leg (threads = OpenMP threads requested by the caller; the reference itself
 * runs this loop on ONE core, V73:117).
 */
#include <math.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline double radfun(double d2) {
    if (d2 < 1e-20) d2 = 1e-20;
    return 0.5 * log(d2) * d2;
}

/* knots_uv: n x 2 column-major scaled knots; out: (r1-r0) x (c1-c0) row-major */
void oracle_tps_eval_grid(const double *knots_uv, const double *c, const double *d3, int64_t n,
                          const double *center2, const double *scale2, double xmin, double ymax,
                          double xres, double yres, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                          int threads, double *out) {
    const double k8pi = 1.0 / (8.0 * M_PI);
    const int64_t nc = c1 - c0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t r = r0; r < r1; ++r) {
        const double y = ymax - ((double)r + 0.5) * yres;
        const double v = (y - center2[1]) / scale2[1];
        for (int64_t cc = c0; cc < c1; ++cc) {
            const double x = xmin + ((double)cc + 0.5) * xres;
            const double u = (x - center2[0]) / scale2[0];
            double acc = 0.0;
            for (int64_t j = 0; j < n; ++j) {
                const double dx = u - knots_uv[j], dy = v - knots_uv[n + j];
                acc += c[j] * radfun(dx * dx + dy * dy);
            }
            out[(r - r0) * nc + (cc - c0)] = d3[0] + d3[1] * u + d3[2] * v + k8pi * acc;
        }
    }
}

/* K (n x n, row-major == column-major) = (1/8pi) radfun(|u_i - u_j|^2) */
void oracle_tps_gram(const double *knots_uv, int64_t n, int threads, double *K) {
    const double k8pi = 1.0 / (8.0 * M_PI);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) {
            const double dx = knots_uv[i] - knots_uv[j], dy = knots_uv[n + i] - knots_uv[n + j];
            K[i * n + j] = k8pi * radfun(dx * dx + dy * dy);
        }
}
