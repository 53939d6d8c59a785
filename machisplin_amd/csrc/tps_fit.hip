// Thin-plate-spline fit on gfx950: fields::Tps(x, Y) (V73:722, V73:751).
//
//   host   collapse replicates, range-scale, Householder QR of T~ = W^1/2 [1 u v]   O(n)
//   GPU    Gram  A = W^1/2 K W^1/2,  K_ij = (1/8pi) 0.5 log(r2) r2                    n^2 logs
//   GPU    A <- Q' A Q  (three two-sided Householder updates); B = A[3:,3:] is SPD
//   fixed lambda:  GPU blocked Cholesky of B + lambda I (FP64 MFMA trailing update)
//                  + triangular solves                                               n^3/3
//   GCV:           GPU Householder tridiagonalisation B = P T P' (symv + rank-2),
//                  g = P' Q2' y rotated along; host picks lambda on T in O(n) per
//                  evaluation (tps_gcv_host.hip); q = (T + lambda I)^-1 g on host;
//                  GPU back-transform c2 = P q                                      4n^3/3
//   host   c = W^1/2 Q [0; c2],  d = R^-1 (Q1'y~ - A[0:3,3:] c2)
//
// A is n x n, full symmetric storage, column-major with leading dimension ld.
#include <cmath>
#include <cstring>
#include <map>
#include <vector>
#include "common.h"
#include "devmath.h"
#include "tps_host.h"

namespace mhs {

// ------------------------------------------------------------------ Gram matrix --
__global__ __launch_bounds__(256) void gram_kernel(const double *__restrict__ u,
                                                   const double *__restrict__ v,
                                                   const double *__restrict__ sw, int n, int64_t ld,
                                                   const double2 *__restrict__ gtab,
                                                   double *__restrict__ A) {
    __shared__ double2 tab[LOG_TAB_N];
    stage_log_table(tab, gtab);
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 16;
    if (i >= n) return;
    const double ui = u[i], vi = v[i], si = sw[i] * (0.5 / (8.0 * M_PI));
    for (int j = j0; j < j0 + 16 && j < n; ++j) {
        const double dx = ui - u[j], dy = vi - v[j];
        const double d2 = fma(dy, dy, dx * dx);
        A[i + (int64_t)j * ld] = si * sw[j] * r2logr2(d2, tab);
    }
}

// --------------------------------------------- two-sided Householder machinery --
// p = A_sub v  (A_sub = A[off:off+t, off:off+t], symmetric full storage): one wave per row
__global__ __launch_bounds__(256) void symv_kernel(const double *__restrict__ A, int64_t ld, int off,
                                                   int t, const double *__restrict__ v,
                                                   double *__restrict__ p) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= t) return;
    const int lane = threadIdx.x & 63;
    const double *col = A + (int64_t)(off + row) * ld + off;  // column `row` == row `row`
    double s = 0.0;
    for (int j = lane; j < t; j += 64) s = fma(col[j], v[j], s);
    s = wave_sum(s);
    if (lane == 0) p[row] = s;
}

// single block: w = tau p - 0.5 tau^2 (p'v) v ; optionally rotate g: g -= tau (v'g) v
__global__ __launch_bounds__(1024) void house_w_kernel(const double *__restrict__ p,
                                                       const double *__restrict__ v,
                                                       const double *__restrict__ tau_ptr, int t,
                                                       double *__restrict__ w, double *g) {
    __shared__ double scratch[17];
    const double tau = *tau_ptr;
    double s = 0.0, sg = 0.0;
    for (int i = threadIdx.x; i < t; i += blockDim.x) {
        s = fma(p[i], v[i], s);
        if (g) sg = fma(g[i], v[i], sg);
    }
    s = block_sum(s, scratch);
    if (g) sg = block_sum(sg, scratch);
    const double alpha = -0.5 * tau * tau * s;
    for (int i = threadIdx.x; i < t; i += blockDim.x) {
        w[i] = fma(alpha, v[i], tau * p[i]);
        if (g) g[i] -= tau * sg * v[i];
    }
}

// A_sub -= v w' + w v'
__global__ __launch_bounds__(256) void syr2_kernel(double *__restrict__ A, int64_t ld, int off, int t,
                                                   const double *__restrict__ v,
                                                   const double *__restrict__ w) {
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 16;
    if (i >= t) return;
    const double vi = v[i], wi = w[i];
    double *a = A + (int64_t)off * ld + off + i;
    for (int j = j0; j < j0 + 16 && j < t; ++j) {
        double x = a[(int64_t)j * ld];
        x -= vi * w[j];
        x -= wi * v[j];
        a[(int64_t)j * ld] = x;
    }
}

// single block: Householder vector of column k of B (B = A[3:,3:]); x = B[k+1:, k], t = len(x).
// Writes v (v[0] = 1) to vbuf and into A below the sub-diagonal, tau[k], offd[k] = beta.
__global__ __launch_bounds__(1024) void house_vec_kernel(double *__restrict__ A, int64_t ld, int col,
                                                         int t, double *__restrict__ vbuf,
                                                         double *__restrict__ tau,
                                                         double *__restrict__ offd, int k) {
    __shared__ double scratch[17];
    double *x = A + (int64_t)col * ld + col + 1;
    double s = 0.0;
    for (int i = 1 + threadIdx.x; i < t; i += blockDim.x) s = fma(x[i], x[i], s);
    s = block_sum(s, scratch);
    const double alpha = x[0];
    __syncthreads();
    if (s == 0.0) {
        if (threadIdx.x == 0) { tau[k] = 0.0; offd[k] = alpha; }
        for (int i = threadIdx.x; i < t; i += blockDim.x) vbuf[i] = (i == 0) ? 1.0 : 0.0;
        return;
    }
    const double beta = -copysign(sqrt(alpha * alpha + s), alpha);
    const double scal = 1.0 / (alpha - beta);
    for (int i = threadIdx.x; i < t; i += blockDim.x) {
        const double vi = (i == 0) ? 1.0 : x[i] * scal;
        vbuf[i] = vi;
        if (i > 0) x[i] = vi;  // keep the reflector for the back-transform
    }
    if (threadIdx.x == 0) { tau[k] = (beta - alpha) / beta; offd[k] = beta; }
}

// single block: r <- H_0 H_1 ... H_{m-3} r, reflectors stored in A below the sub-diagonal
__global__ __launch_bounds__(1024) void backtransform_kernel(const double *__restrict__ A, int64_t ld,
                                                             int off0, int m,
                                                             const double *__restrict__ tau,
                                                             double *__restrict__ r) {
    __shared__ double scratch[17];
    for (int k = m - 3; k >= 0; --k) {
        const int t = m - k - 1;
        const double tk = tau[k];
        if (tk == 0.0) continue;  // uniform
        const double *x = A + (int64_t)(off0 + k) * ld + off0 + k + 1;
        double *rs = r + k + 1;
        double s = 0.0;
        for (int i = threadIdx.x; i < t; i += blockDim.x) s = fma(i == 0 ? 1.0 : x[i], rs[i], s);
        s = block_sum(s, scratch) * tk;
        for (int i = threadIdx.x; i < t; i += blockDim.x) rs[i] -= s * (i == 0 ? 1.0 : x[i]);
        __syncthreads();
    }
}

// ---- fused tridiagonalisation step (two launches per Householder step) ---------------------
// tri_step_kernel (single block): given p = A_trail v_k, finish step k -- w_k = tau p - tau^2/2
// (p'v) v, rotate g -- then bring column k+1 up to date with the rank-2 update, take ITS Householder
// vector v_{k+1} and store the reflector.  tri_fused_kernel (one wave per column, whole chip):
// apply the rank-2 update of step k to the remaining (t-1)^2 block and, in the same pass over the
// matrix, form p_{k+1} = A_new v_{k+1}.  The matrix is read and written once per step (16 B per
// element; HBM/Infinity-Cache bound) instead of read, read, written by separate symv / syr2 passes.
__global__ __launch_bounds__(1024) void tri_step_kernel(double *__restrict__ A, int64_t ld, int col, int t,
                                                        const double *__restrict__ p,
                                                        const double *__restrict__ v,
                                                        double *__restrict__ vnext, double *__restrict__ w,
                                                        double *__restrict__ g, double *__restrict__ tau,
                                                        double *__restrict__ offd, int k) {
    __shared__ double scratch[17];
    const double tk = tau[k];
    double s = 0.0, sg = 0.0;
    for (int i = threadIdx.x; i < t; i += blockDim.x) { s = fma(p[i], v[i], s); sg = fma(g[i], v[i], sg); }
    s = block_sum(s, scratch);
    sg = block_sum(sg, scratch);
    const double alpha_w = -0.5 * tk * tk * s;
    const double v0 = v[0];
    const double w0 = fma(alpha_w, v0, tk * p[0]);
    double *x = A + (int64_t)col * ld + col;  // column k+1 of B from its diagonal down
    double ss = 0.0;
    for (int i = threadIdx.x; i < t; i += blockDim.x) {
        const double wi = fma(alpha_w, v[i], tk * p[i]);
        w[i] = wi;
        g[i] -= tk * sg * v[i];
        if (i == 0) x[0] = x[0] - (v0 * w0 + w0 * v0);
        else {
            const double xi = x[i] - (v[i] * w0 + wi * v0);
            x[i] = xi;
            if (i >= 2) ss = fma(xi, xi, ss);
        }
    }
    if (t < 2) return;
    ss = block_sum(ss, scratch);  // also orders the x[] writes before the reads below
    const double alpha = x[1];
    if (ss == 0.0) {
        for (int i = threadIdx.x; i + 1 < t; i += blockDim.x) vnext[i] = (i == 0) ? 1.0 : 0.0;
        if (threadIdx.x == 0) { tau[k + 1] = 0.0; offd[k + 1] = alpha; }
        return;
    }
    const double beta = -copysign(sqrt(alpha * alpha + ss), alpha);
    const double scal = 1.0 / (alpha - beta);
    __syncthreads();
    for (int i = 1 + threadIdx.x; i < t; i += blockDim.x) {
        const double vi = (i == 1) ? 1.0 : x[i] * scal;
        vnext[i - 1] = vi;
        if (i >= 2) x[i] = vi;  // reflector kept for the back-transform
    }
    if (threadIdx.x == 0) { tau[k + 1] = (beta - alpha) / beta; offd[k + 1] = beta; }
}

// block (col0+1.., col0+1..) of size (t-1): a_ji -= v_j w_i + w_j v_i ; pnext_i = sum_j a_ji vnext_j
__global__ __launch_bounds__(256) void tri_fused_kernel(double *__restrict__ A, int64_t ld, int col0, int t,
                                                        const double *__restrict__ v,
                                                        const double *__restrict__ w,
                                                        const double *__restrict__ vnext,
                                                        double *__restrict__ pnext) {
    const int i = 1 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= t) return;
    const int lane = threadIdx.x & 63;
    double *a = A + (int64_t)(col0 + i) * ld + col0;
    const double vi = v[i], wi = w[i];
    double acc = 0.0;
    for (int j = 1 + lane; j < t; j += 64) {
        const double x = a[j] - (v[j] * wi + w[j] * vi);
        a[j] = x;
        acc = fma(x, vnext[j - 1], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) pnext[i - 1] = acc;
}

__global__ void add_diag_kernel(double *A, int64_t ld, int off, int m, double lam) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) A[(int64_t)(off + i) * ld + off + i] += lam;
}

// ------------------------------------------------------------------- Cholesky --
constexpr int NB = 64;  // panel width

// single block (256 threads): unblocked Cholesky of the nb x nb diagonal block in LDS
__global__ __launch_bounds__(256) void potf2_kernel(double *__restrict__ A, int64_t ld, int off, int nb,
                                                    int *__restrict__ info) {
    __shared__ double s[NB][NB + 1];
    double *a = A + (int64_t)off * ld + off;
    for (int e = threadIdx.x; e < nb * nb; e += 256) { const int i = e % nb, j = e / nb; s[i][j] = a[i + (int64_t)j * ld]; }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const double djj = s[j][j];
        if (!(djj > 0.0)) { if (threadIdx.x == 0) atomicCAS(info, 0, off + j + 1); return; }
        const double l = sqrt(djj);
        __syncthreads();
        if (threadIdx.x == 0) s[j][j] = l;
        for (int i = j + 1 + threadIdx.x; i < nb; i += 256) s[i][j] /= l;
        __syncthreads();
        // trailing update of the lower triangle
        const int rem = nb - j - 1;
        for (int e = threadIdx.x; e < rem * rem; e += 256) {
            const int i = j + 1 + e % rem, c = j + 1 + e / rem;
            if (i >= c) s[i][c] -= s[i][j] * s[c][j];
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < nb * nb; e += 256) { const int i = e % nb, j = e / nb; if (i >= j) a[i + (int64_t)j * ld] = s[i][j]; }
}

// panel solve: X = A[off+nb:, off:off+nb] * L^-T, one thread per row
__global__ __launch_bounds__(256) void trsm_kernel(double *__restrict__ A, int64_t ld, int off, int nb,
                                                   int t) {
    __shared__ double L[NB][NB + 1];
    const double *d = A + (int64_t)off * ld + off;
    for (int e = threadIdx.x; e < nb * nb; e += 256) { const int i = e % nb, j = e / nb; L[i][j] = (i >= j) ? d[i + (int64_t)j * ld] : 0.0; }
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= t) return;
    double *row = A + (int64_t)off * ld + off + nb + r;
    double x[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if (j < nb) {
            double s = row[(int64_t)j * ld];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= x[k] * L[j][k];
            x[j] = s / L[j][j];
            row[(int64_t)j * ld] = x[j];
        }
    }
}

// trailing update C -= P P' on the lower triangle with v_mfma_f64_16x16x4_f64.
// P = A[off2:off2+t, pc:pc+nb] (t x nb), C = A[off2:, off2:].  Block tile 64x64, 4 waves as
// 2x2 of 32x32; the two 64 x nb panel tiles are staged in LDS with a row stride of 80
// doubles so the four k-rows of one ds_read_b64 wave access land on disjoint banks.
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int SYRK_LDS_STRIDE = 80;

__global__ __launch_bounds__(256) void syrk_mfma_kernel(double *__restrict__ A, int64_t ld, int off2,
                                                        int pc, int nb, int t) {
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi) return;  // lower triangle of tiles only
    __shared__ double sI[NB * SYRK_LDS_STRIDE];
    __shared__ double sJ[NB * SYRK_LDS_STRIDE];
    const int i0 = bi * 64, j0 = bj * 64;
    // stage P[i0:i0+64, 0:nb] and P[j0:j0+64, 0:nb]; element (r, k) at s[k*STRIDE + r]
    for (int e = threadIdx.x; e < 64 * nb; e += 256) {
        const int r = e & 63, k = e >> 6;
        const double *pcol = A + (int64_t)(pc + k) * ld + off2;
        sI[k * SYRK_LDS_STRIDE + r] = (i0 + r < t) ? pcol[i0 + r] : 0.0;
        sJ[k * SYRK_LDS_STRIDE + r] = (j0 + r < t) ? pcol[j0 + r] : 0.0;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;  // wave's 32x32 sub-tile
    const int l15 = lane & 15, l4 = lane >> 4;
    // acc[a][b]: D'[row = j][col = i] tile, i-sub-block a, j-sub-block b
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < nb; k0 += 4) {
        double fi[2], fj[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            fi[a] = sI[(k0 + l4) * SYRK_LDS_STRIDE + wi + a * 16 + l15];
            fj[a] = sJ[(k0 + l4) * SYRK_LDS_STRIDE + wj + a * 16 + l15];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                // A-operand rows = j (P_j), B-operand cols = i (P_i): D'[j][i] += sum_k P[j][k] P[i][k]
                acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[b], fi[a], acc[a][b], 0, 0, 0);
    }
    // D' layout: lane holds D'[row = l4 + 4 r][col = l15]  ->  C[i = col][j = row]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi + a * 16 + l15;
                const int j = j0 + wj + b * 16 + l4 + 4 * r;
                if (i < t && j < t && i >= j) {
                    double *c = A + (int64_t)(off2 + j) * ld + off2 + i;
                    *c -= acc[a][b][r];
                }
            }
}

// single block: solve L L' x = b in place (L = lower triangle of A[off:off+m, off:off+m])
__global__ __launch_bounds__(1024) void potrs_kernel(const double *__restrict__ A, int64_t ld, int off,
                                                     int m, double *__restrict__ x) {
    __shared__ double scratch[17];
    __shared__ double piv;
    const double *L = A + (int64_t)off * ld + off;
    // forward, column oriented: x_j /= L_jj ; x[j+1:] -= x_j L[j+1:, j]
    for (int j = 0; j < m; ++j) {
        if (threadIdx.x == 0) { x[j] /= L[j + (int64_t)j * ld]; piv = x[j]; }
        __syncthreads();
        const double xj = piv;
        const double *c = L + (int64_t)j * ld;
        for (int i = j + 1 + threadIdx.x; i < m; i += blockDim.x) x[i] -= xj * c[i];
        __syncthreads();
    }
    // backward, dot oriented on columns of L: x_j = (x_j - sum_{i>j} L[i][j] x_i) / L_jj
    for (int j = m - 1; j >= 0; --j) {
        const double *c = L + (int64_t)j * ld;
        double s = 0.0;
        for (int i = j + 1 + threadIdx.x; i < m; i += blockDim.x) s = fma(c[i], x[i], s);
        s = block_sum(s, scratch);
        if (threadIdx.x == 0) x[j] = (x[j] - s) / c[j];
        __syncthreads();
    }
}

static int cholesky_solve(double *A, int64_t ld, int off, int m, double *x_dev, hipStream_t s) {
    DevBuf<int> info;
    MHS_HIP(info.alloc(1));
    MHS_HIP(hipMemsetAsync(info.p, 0, sizeof(int), s));
    for (int j = 0; j < m; j += NB) {
        const int nb = std::min(NB, m - j);
        const int t = m - j - nb;
        hipLaunchKernelGGL(potf2_kernel, dim3(1), dim3(256), 0, s, A, ld, off + j, nb, info.p);
        if (t > 0) {
            hipLaunchKernelGGL(trsm_kernel, dim3((t + 255) / 256), dim3(256), 0, s, A, ld, off + j, nb, t);
            const int nt = (t + 63) / 64;
            hipLaunchKernelGGL(syrk_mfma_kernel, dim3(nt, nt), dim3(256), 0, s, A, ld, off + j + nb,
                               off + j, nb, t);
        }
    }
    hipLaunchKernelGGL(potrs_kernel, dim3(1), dim3(1024), 0, s, A, ld, off, m, x_dev);
    MHS_HIP(hipGetLastError());
    int h_info = 0;
    MHS_HIP(hipMemcpyAsync(&h_info, info.p, sizeof(int), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    if (h_info != 0) {
        set_error("mhs_tps_fit: Q2'KQ2 + lambda I is not positive definite (pivot %d)", h_info);
        return MHS_ERR_NUMERIC;
    }
    return MHS_OK;
}

// fields' Krig.replicates: unique locations (first-appearance order), means, counts
static void collapse_replicates(const double *xy, const double *y, int64_t N, std::vector<double> &xm,
                                std::vector<double> &ym_out, std::vector<double> &w, double &pure_ss) {
    std::map<std::pair<double, double>, int64_t> seen;
    std::vector<int64_t> gid((size_t)N);
    std::vector<double> ux, uy, sum, cnt;
    for (int64_t i = 0; i < N; ++i) {
        const auto key = std::make_pair(xy[i], xy[N + i]);
        auto it = seen.find(key);
        if (it == seen.end()) {
            it = seen.emplace(key, (int64_t)ux.size()).first;
            ux.push_back(key.first); uy.push_back(key.second); sum.push_back(0.0); cnt.push_back(0.0);
        }
        gid[i] = it->second;
        sum[it->second] += y[i];
        cnt[it->second] += 1.0;
    }
    const int64_t n = (int64_t)ux.size();
    xm.resize(2 * n);
    ym_out.resize(n);
    w = cnt;
    for (int64_t k = 0; k < n; ++k) { xm[k] = ux[k]; xm[n + k] = uy[k]; ym_out[k] = sum[k] / cnt[k]; }
    pure_ss = 0.0;
    for (int64_t i = 0; i < N; ++i) { const double r = y[i] - ym_out[gid[i]]; pure_ss += r * r; }
}

}  // namespace mhs

using namespace mhs;

extern "C" int mhs_tps_fit(const double *xy, const double *y, int64_t N, double lambda, int gcv_mode,
                           mhs_tps **out) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(xy && y && out, "NULL argument");
    MHS_REQUIRE(N > 3 && N < (1LL << 30), "need more than 3 observations");
    MHS_REQUIRE(std::isnan(lambda) || lambda >= 0, "lambda must be >= 0 or NaN");
    MHS_REQUIRE(gcv_mode == MHS_GCV_FIELDS || gcv_mode == MHS_GCV_CONVERGED, "bad gcv_mode");
    for (int64_t i = 0; i < N; ++i)
        if (!std::isfinite(xy[i]) || !std::isfinite(xy[N + i]) || !std::isfinite(y[i])) {
            set_error("mhs_tps_fit: non-finite input at row %lld (drop NA rows first, V73:706)", (long long)i);
            return MHS_ERR_INVALID;
        }

    std::vector<double> xm, ym, w;
    double pure_ss = 0.0;
    collapse_replicates(xy, y, N, xm, ym, w, pure_ss);
    const int64_t n = (int64_t)ym.size();
    if (n <= 3) { set_error("mhs_tps_fit: need more than 3 distinct locations"); return MHS_ERR_NUMERIC; }
    const int m = (int)(n - 3);

    // range scaling (fields scale.type = "range")
    double center[2], scale[2];
    std::vector<double> uv(2 * n), sw(n);
    for (int d = 0; d < 2; ++d) {
        double lo = xm[d * n], hi = xm[d * n];
        for (int64_t i = 0; i < n; ++i) { lo = std::min(lo, xm[d * n + i]); hi = std::max(hi, xm[d * n + i]); }
        center[d] = lo; scale[d] = hi - lo;
        if (!(scale[d] > 0)) { set_error("mhs_tps_fit: degenerate station coordinates (zero range)"); return MHS_ERR_NUMERIC; }
        for (int64_t i = 0; i < n; ++i) uv[d * n + i] = (xm[d * n + i] - center[d]) / scale[d];
    }
    for (int64_t i = 0; i < n; ++i) sw[i] = sqrt(w[i]);

    // QR of T~ = W^1/2 [1 u v]
    std::vector<double> T(3 * n), hv[3];
    double htau[3], R[9];
    for (int64_t i = 0; i < n; ++i) { T[i] = sw[i]; T[n + i] = sw[i] * uv[i]; T[2 * n + i] = sw[i] * uv[n + i]; }
    qr_n3(T, n, hv, htau, R);
    if (fabs(R[8]) < 1e-10 * fabs(R[0]) || fabs(R[4]) < 1e-10 * fabs(R[0])) {
        set_error("mhs_tps_fit: collinear station coordinates");
        return MHS_ERR_NUMERIC;
    }
    std::vector<double> wv(n);  // Q' y~
    for (int64_t i = 0; i < n; ++i) wv[i] = sw[i] * ym[i];
    for (int k = 0; k < 3; ++k) apply_reflector(hv[k], htau[k], wv.data(), n);

    hipStream_t s = ctx().stream;
    const int64_t ld = (n + 15) & ~(int64_t)15;
    DevBuf<double> A, duv, dsw, vbuf, pbuf, wbuf, gbuf, tau, offd;
    MHS_HIP(A.alloc((size_t)(ld * n)));
    MHS_HIP(duv.alloc((size_t)(2 * n)));
    MHS_HIP(dsw.alloc((size_t)n));
    MHS_HIP(vbuf.alloc((size_t)n));
    MHS_HIP(pbuf.alloc((size_t)n));
    MHS_HIP(wbuf.alloc((size_t)n));
    MHS_HIP(gbuf.alloc((size_t)n));
    MHS_HIP(tau.alloc((size_t)n + 3));
    MHS_HIP(offd.alloc((size_t)n));
    MHS_HIP(hipMemcpyAsync(duv.p, uv.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemcpyAsync(dsw.p, sw.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));

    {
        dim3 grid((unsigned)((n + 63) / 64), (unsigned)((n + 63) / 64));
        hipLaunchKernelGGL(gram_kernel, grid, dim3(256), 0, s, duv.p, duv.p + n, dsw.p, (int)n, ld,
                           ctx().log_tab, A.p);
    }
    // A <- H3 H2 H1 A H1 H2 H3 (full-length reflectors, leading zeros)
    MHS_HIP(hipMemcpyAsync(tau.p + n, htau, sizeof(double) * 3, hipMemcpyHostToDevice, s));
    for (int k = 0; k < 3; ++k) {
        MHS_HIP(hipMemcpyAsync(vbuf.p, hv[k].data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(symv_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, A.p, ld, 0, (int)n, vbuf.p, pbuf.p);
        hipLaunchKernelGGL(house_w_kernel, dim3(1), dim3(1024), 0, s, pbuf.p, vbuf.p, tau.p + n + k, (int)n, wbuf.p, (double *)nullptr);
        dim3 grid((unsigned)((n + 63) / 64), (unsigned)((n + 63) / 64));
        hipLaunchKernelGGL(syr2_kernel, grid, dim3(256), 0, s, A.p, ld, 0, (int)n, vbuf.p, wbuf.p);
        MHS_HIP(hipStreamSynchronize(s));  // vbuf is reused by the next upload
    }
    MHS_HIP(hipGetLastError());
    // rows 0..2 of the projected matrix, columns 3..n-1 (by symmetry: columns 0..2, rows 3..)
    std::vector<double> Atop(3 * (size_t)m);
    for (int k = 0; k < 3; ++k)
        MHS_HIP(hipMemcpyAsync(&Atop[(size_t)k * m], A.p + (int64_t)k * ld + 3, sizeof(double) * m, hipMemcpyDeviceToHost, s));

    std::vector<double> c2((size_t)m);
    double lam = lambda, gcv = NAN, eff_df = NAN;
    if (!std::isnan(lambda)) {
        // fixed lambda: Cholesky of B + lambda I, solve for c2 = (B + lambda I)^-1 w2
        hipLaunchKernelGGL(add_diag_kernel, dim3((m + 255) / 256), dim3(256), 0, s, A.p, ld, 3, m, lam);
        MHS_HIP(hipMemcpyAsync(gbuf.p, wv.data() + 3, sizeof(double) * m, hipMemcpyHostToDevice, s));
        if (int rc = cholesky_solve(A.p, ld, 3, m, gbuf.p, s)) return rc;
        MHS_HIP(hipMemcpyAsync(c2.data(), gbuf.p, sizeof(double) * m, hipMemcpyDeviceToHost, s));
        MHS_HIP(hipStreamSynchronize(s));
    } else {
        // tridiagonalise B in place, rotating g = P' w2 along
        MHS_HIP(hipMemcpyAsync(gbuf.p, wv.data() + 3, sizeof(double) * m, hipMemcpyHostToDevice, s));
        // step 0: Householder vector of column 0 and p = A_trail v_0; then two launches per step
        DevBuf<double> vbuf2;
        MHS_HIP(vbuf2.alloc((size_t)n));
        double *vcur = vbuf.p, *vnext = vbuf2.p;
        if (m >= 2) {
            hipLaunchKernelGGL(house_vec_kernel, dim3(1), dim3(1024), 0, s, A.p, ld, 3, m - 1, vcur, tau.p, offd.p, 0);
            hipLaunchKernelGGL(symv_kernel, dim3((unsigned)((m - 1 + 3) / 4)), dim3(256), 0, s, A.p, ld, 4, m - 1, vcur, pbuf.p);
        }
        for (int k = 0; k + 1 < m; ++k) {
            const int t = m - k - 1;
            hipLaunchKernelGGL(tri_step_kernel, dim3(1), dim3(1024), 0, s, A.p, ld, 3 + k + 1, t, pbuf.p, vcur, vnext,
                               wbuf.p, gbuf.p + k + 1, tau.p, offd.p, k);
            if (t >= 2)
                hipLaunchKernelGGL(tri_fused_kernel, dim3((unsigned)((t - 1 + 3) / 4)), dim3(256), 0, s, A.p, ld,
                                   3 + k + 1, t, vcur, wbuf.p, vnext, pbuf.p);
            std::swap(vcur, vnext);
        }
        MHS_HIP(hipGetLastError());
        std::vector<double> diag((size_t)m), off((size_t)std::max(m - 1, 1)), g((size_t)m), q((size_t)m);
        MHS_HIP(hipMemcpy2DAsync(diag.data(), sizeof(double), A.p + 3 * ld + 3, sizeof(double) * (ld + 1),
                                 sizeof(double), (size_t)m, hipMemcpyDeviceToHost, s));
        if (m > 1) MHS_HIP(hipMemcpyAsync(off.data(), offd.p, sizeof(double) * (m - 1), hipMemcpyDeviceToHost, s));
        MHS_HIP(hipMemcpyAsync(g.data(), gbuf.p, sizeof(double) * m, hipMemcpyDeviceToHost, s));
        MHS_HIP(hipStreamSynchronize(s));
        TridiagGcv tg;
        tg.a = diag.data(); tg.b = off.data(); tg.g = g.data(); tg.m = m; tg.n = n; tg.N = N; tg.pure_ss = pure_ss;
        lam = tg.find_lambda(gcv_mode);
        if (std::isnan(lam)) { set_error("mhs_tps_fit: GCV search failed"); return MHS_ERR_NUMERIC; }
        tg.eval(lam, &gcv, &eff_df, q.data());
        MHS_HIP(hipMemcpyAsync(gbuf.p, q.data(), sizeof(double) * m, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(backtransform_kernel, dim3(1), dim3(1024), 0, s, A.p, ld, 3, m, tau.p, gbuf.p);
        MHS_HIP(hipGetLastError());
        MHS_HIP(hipMemcpyAsync(c2.data(), gbuf.p, sizeof(double) * m, hipMemcpyDeviceToHost, s));
        MHS_HIP(hipStreamSynchronize(s));
    }

    // d = R^-1 (w1 - Atop c2) ; c~ = Q [0; c2] ; c = W^1/2 c~
    double rhs[3];
    for (int k = 0; k < 3; ++k) {
        double sdot = 0.0;
        for (int j = 0; j < m; ++j) sdot += Atop[(size_t)k * m + j] * c2[j];
        rhs[k] = wv[k] - sdot;
    }
    double dd[3];
    dd[2] = rhs[2] / R[8];
    dd[1] = (rhs[1] - R[1 + 3 * 2] * dd[2]) / R[4];
    dd[0] = (rhs[0] - R[0 + 3 * 1] * dd[1] - R[0 + 3 * 2] * dd[2]) / R[0];
    std::vector<double> ct((size_t)n, 0.0);
    for (int j = 0; j < m; ++j) ct[3 + j] = c2[j];
    for (int k = 2; k >= 0; --k) apply_reflector(hv[k], htau[k], ct.data(), n);

    mhs_tps *t = new mhs_tps();
    t->n = n;
    t->lambda = lam; t->eff_df = eff_df; t->gcv = gcv;
    memcpy(t->center, center, sizeof(center));
    memcpy(t->scale, scale, sizeof(scale));
    memcpy(t->d, dd, sizeof(dd));
    t->c.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) t->c[i] = sw[i] * ct[i];
    t->knots_uv = uv;
    if (int rc = upload_knots(t)) { mhs_tps_free(t); return rc; }
    *out = t;
    return MHS_OK;
}
