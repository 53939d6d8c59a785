// Fit of the ensemble's linear member on the device: mgcv::gam(resp ~ a + b + ...) without smooth terms
// (V73:195 formula; V73:252 the CV fits, V73:600 the final fit) is ordinary least squares, which mgcv and
// stats::lm solve by a Householder QR of the model matrix.  Same here: one block reduces [1 X y] (n x (p+2),
// column-major, a few hundred KB: L2-resident) column by column; the host back-substitutes the (p+1) x (p+1)
// triangle.  Not a hot path (the matrix is tiny): it is here so that Step 1's linear member and its ten hold-out
// refits need no host round trip when the predictors already live on the device side of the shim.
#include <cmath>
#include <vector>
#include "common.h"
#include "devmath.h"

namespace mhs {

constexpr int LM_QMAX = 16;   // columns of [1 X y]: p <= 14

// In-place Householder QR of M (n x q); afterwards the upper triangle holds R (the last column: Q'y) and the
// reflectors lie below the diagonal.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void lm_qr_kernel(double *__restrict__ M, int n, int q, double *__restrict__ Rout) {
    __shared__ double scratch[17];
    __shared__ double dots[LM_QMAX];
    for (int j = 0; j < q && j < n - 1; ++j) {
        double *x = M + (int64_t)j * n;
        double part = 0.0;
        for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) part = fma(x[i], x[i], part);
        const double ss = block_sum(part, scratch);
        const double alpha = x[j];
        __syncthreads();
        double beta = alpha, tau = 0.0, scal = 0.0;
        if (ss != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + ss), alpha);
            tau = (beta - alpha) / beta;
            scal = 1.0 / (alpha - beta);
        }
        for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) x[i] *= scal;    // v (v_j = 1 implied)
        if (threadIdx.x == 0) x[j] = beta;
        __syncthreads();
        for (int k = j + 1; k < q; ++k) {          // apply H_j to the columns to the right
            double *c = M + (int64_t)k * n;
            double d = 0.0;
            for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) d = fma(x[i], c[i], d);
            d = block_sum(d, scratch);
            if (threadIdx.x == 0) dots[k] = d + c[j];
            __syncthreads();
            const double w = tau * dots[k];
            for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) c[i] -= w * x[i];
            if (threadIdx.x == 0) c[j] -= w;
            __syncthreads();
        }
    }
    for (int e = threadIdx.x; e < q * q; e += blockDim.x) {
        const int r = e % q, c = e / q;
        Rout[e] = r <= c && r < n ? M[(int64_t)c * n + r] : 0.0;
    }
}

}  // namespace mhs

using namespace mhs;

extern "C" int mhs_lm_fit(const double *X, const double *y, int64_t n, int p, double *coef) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(X && y && coef, "NULL argument");
    MHS_REQUIRE(p >= 1 && p + 2 <= LM_QMAX, "p out of range");
    MHS_REQUIRE(n > p + 1 && n < (1LL << 30), "need more rows than coefficients");
    const int q = p + 2;
    std::vector<double> M((size_t)n * q);
    for (int64_t i = 0; i < n; ++i) {
        M[(size_t)i] = 1.0;
        if (!std::isfinite(y[i])) { set_error("mhs_lm_fit: non-finite response at row %lld", (long long)i); return MHS_ERR_INVALID; }
        M[(size_t)(q - 1) * n + i] = y[i];
    }
    for (int j = 0; j < p; ++j)
        for (int64_t i = 0; i < n; ++i) {
            const double v = X[(size_t)j * n + i];
            if (!std::isfinite(v)) { set_error("mhs_lm_fit: non-finite predictor %d at row %lld (drop NA rows first, V73:154)", j, (long long)i); return MHS_ERR_INVALID; }
            M[(size_t)(j + 1) * n + i] = v;
        }
    hipStream_t s = ctx().stream;
    DevBuf<double> dM, dR;
    MHS_HIP(dM.alloc(M.size()));
    MHS_HIP(dR.alloc((size_t)q * q));
    MHS_HIP(hipMemcpyAsync(dM.p, M.data(), sizeof(double) * M.size(), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lm_qr_kernel, dim3(1), dim3(1024), 0, s, dM.p, (int)n, q, dR.p);
    MHS_HIP(hipGetLastError());
    std::vector<double> R((size_t)q * q);
    MHS_HIP(hipMemcpyAsync(R.data(), dR.p, sizeof(double) * R.size(), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    // R[r + q c]: the leading (p+1) x (p+1) triangle and, in column q-1, the first p+1 entries of Q'y
    const int k = p + 1;
    // lm's rank test (dqrdc2, tol = 1e-7): a column is dropped when what is left of it after the earlier columns
    // have been projected out, |R_jj|, falls below tol x the column's ORIGINAL norm -- a per-column, unit-free test
    // (elevation in metres beside a narrow-range covariate must not read as rank deficiency).
    for (int j = 0; j < k; ++j) {
        double cn = 0.0;
        for (int64_t i = 0; i < n; ++i) cn += M[(size_t)j * n + i] * M[(size_t)j * n + i];
        cn = sqrt(cn);
        if (!(fabs(R[(size_t)j + (size_t)q * j]) > 1e-7 * cn)) {
            set_error("mhs_lm_fit: rank-deficient design (column %d)", j);
            return MHS_ERR_NUMERIC;
        }
    }
    for (int j = k - 1; j >= 0; --j) {
        double sum = R[(size_t)j + (size_t)q * (q - 1)];
        for (int c = j + 1; c < k; ++c) sum -= R[(size_t)j + (size_t)q * c] * coef[c];
        coef[j] = sum / R[(size_t)j + (size_t)q * j];
    }
    return MHS_OK;
}
