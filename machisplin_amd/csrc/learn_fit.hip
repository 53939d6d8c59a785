// Fitting of the two cheap learners of SURVEY.md section 8(f) rank 4 that are iterative optimisations rather than
// one linear solve (lm_fit.hip has the linear member):
//
//   * kernlab::ksvm(mod.form, data)  (V73:251 in the CV loop, V73:560 final): eps-SVR with the RBF kernel on scaled
//     data, C = 1, epsilon = 0.1, tol = 0.001.  kernlab's solver is the libsvm SMO (second-order working-set
//     selection, Fan / Chen / Lin 2005); here the Gram matrix is built once in HBM (n^2 doubles: 200 MB for 5 000
//     stations, 3.2 GB for 20 000) and the whole SMO runs in ONE resident kernel: a thread owns up to 8 stations'
//     (K beta, alpha, alpha*, y) in registers, an iteration is two arg-reductions (block tree + one slot per block
//     in global memory + a grid barrier when n needs more than one block of 1 024 threads) and two coalesced row
//     reads of K.  No host round trip per iteration (a launch per SMO step would be ~10^5 launches).
//   * nnet::nnet(size = 10, linout = TRUE, maxit = 10000)  (V73:249, V73:463): sum-of-squares objective minimised by
//     R's optim "BFGS" (vmmin, src/appl/optim.c).  One resident block: the rows are dealt over 256 threads, each
//     keeps its partial gradient in registers, a fixed-order tree adds them (the iteration path of a quasi-Newton
//     method amplifies any run-to-run difference, so no atomics), the BFGS matrix lives in LDS (packed lower
//     triangle) and the line search / update logic of vmmin is evaluated redundantly by every thread.
#include <algorithm>
#include <cmath>
#include <vector>
#include "common.h"

namespace mhs {

// ------------------------------------------------------------------------------------------------------- eps-SVR --
constexpr int SMO_T = 1024;     // threads per block
constexpr int SMO_E = 8;        // stations per thread
constexpr int SMO_MAXB = 64;    // blocks (n <= 524 288; the n^2 Gram matrix gives out long before)

__global__ __launch_bounds__(256) void rbf_gram_kernel(const double *__restrict__ Z, int n, int p, double sigma,
                                                       double *__restrict__ K) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)n * n) return;
    const int i = (int)(e / n), j = (int)(e - (int64_t)i * n);
    double d2 = 0.0;
    for (int k = 0; k < p; ++k) { const double d = Z[(int64_t)i * p + k] - Z[(int64_t)j * p + k]; d2 = fma(d, d, d2); }
    K[e] = exp(-sigma * d2);
}

__device__ __forceinline__ void st_u64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_f64(unsigned long long *p, double v) { st_u64(p, (unsigned long long)__double_as_longlong(v)); }
__device__ __forceinline__ double ld_f64(const unsigned long long *p) { return __longlong_as_double((long long)ld_u64(p)); }

// All blocks of the (cooperatively launched) grid meet here.  The slots the blocks exchange are written and read
// with agent-scope atomics, which are coherent across the XCDs' L2s by themselves: waiting for this thread's
// stores (vmcnt) before the arrival is all the ordering needed -- no L2 write-back.
__device__ __forceinline__ void grid_barrier(unsigned long long *counter, unsigned nblocks, unsigned long long &target) {
    if (nblocks > 1) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this thread's slot stores have been performed
    __syncthreads();
    if (nblocks > 1) {
        target += nblocks;
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

// arg-max of (key, idx) over the block, ties to the LARGER idx (libsvm's select_working_set scans t = 0 .. 2n-1 with
// '>=' / '<=': the last of equal candidates wins; SMO_NONE = -1 loses every tie); every thread returns the winner
__device__ __forceinline__ void block_argmax(double &key, int &idx, double *skey, int *sidx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double k2 = __shfl_xor(key, o);
        const int i2 = __shfl_xor(idx, o);
        if (k2 > key || (k2 == key && i2 > idx)) { key = k2; idx = i2; }
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { skey[wave] = key; sidx[wave] = idx; }
    __syncthreads();
    key = skey[0]; idx = sidx[0];
#pragma unroll
    for (int w = 1; w < SMO_T / 64; ++w) {
        const double k2 = skey[w];
        const int i2 = sidx[w];
        if (k2 > key || (k2 == key && i2 > idx)) { key = k2; idx = i2; }
    }
}

// slots (unsigned long long words, per parity and block): I: {gmax, a_i, idx, gmax2}; J: {-obj, a_j, G_j, K_ij, idx}
constexpr int SLOT_W = 8;
constexpr int SMO_NONE = -1;      // "no candidate": below every variable index, so it never wins a tie
struct SmoOut { double rho; long long iters; double violation; int status; };

__global__ __launch_bounds__(SMO_T) void svr_smo_kernel(const double *__restrict__ K, const double *__restrict__ y, int n,
                                                        double C, double eps, double tol, long long max_iter,
                                                        unsigned long long *slots, unsigned long long *counter,
                                                        double *__restrict__ al_out, double *__restrict__ as_out,
                                                        double *__restrict__ kb_out, SmoOut *out) {
    __shared__ double skey[SMO_T / 64], spay[4];
    __shared__ int sidx[SMO_T / 64];
    const unsigned nb = gridDim.x;
    const int stride = (int)nb * SMO_T;
    const int g0 = (int)blockIdx.x * SMO_T + (int)threadIdx.x;
    const double TAU = 1e-12, NEG = -INFINITY;
    double kb[SMO_E], al[SMO_E], as[SMO_E], yk[SMO_E];
#pragma unroll
    for (int r = 0; r < SMO_E; ++r) {
        const int k = g0 + r * stride;
        kb[r] = 0.0; al[r] = 0.0; as[r] = 0.0;
        yk[r] = k < n ? y[k] : 0.0;
    }
    unsigned long long target = 0;
    long long it = 0;
    double viol = INFINITY;
    int status = 0;
    unsigned long long *slotI = slots, *slotJ = slots + 2 * SMO_MAXB * SLOT_W;
    for (;; ++it) {
        const int par = (int)(it & 1);
        // ---- i = argmax over I_up of -s G ; gmax2 = max over I_low of s G
        double best = NEG, low = NEG;
        int bi = SMO_NONE;
#pragma unroll
        for (int r = 0; r < SMO_E; ++r) {
            const int k = g0 + r * stride;
            if (k < n) {
                const double gu = kb[r] + eps - yk[r], gd = -kb[r] + eps + yk[r];     // gradients of alpha_k, alpha*_k
                if (al[r] < C && (-gu > best || (-gu == best && k > bi))) { best = -gu; bi = k; }
                if (as[r] > 0.0 && (gd > best || (gd == best && k + n > bi))) { best = gd; bi = k + n; }
                if (al[r] > 0.0) low = fmax(low, gu);
                if (as[r] < C) low = fmax(low, -gd);
            }
        }
        block_argmax(best, bi, skey, sidx);
        {
            int dummy = 0;
            block_argmax(low, dummy, skey, sidx);
        }
        if (nb > 1) {
            unsigned long long *s = slotI + ((size_t)par * SMO_MAXB + blockIdx.x) * SLOT_W;
            if (threadIdx.x == 0) { st_f64(s + 0, best); st_u64(s + 2, (unsigned long long)(unsigned)bi); st_f64(s + 3, low); }
        }
        // the owner of the block's candidate publishes its alpha (one block: through LDS)
        {
            const int st = bi < n ? bi : bi - n;
#pragma unroll
            for (int r = 0; r < SMO_E; ++r)
                if (g0 + r * stride == st && bi != SMO_NONE) {
                    const double ai_local = bi < n ? al[r] : as[r];
                    if (nb > 1) st_f64(slotI + ((size_t)par * SMO_MAXB + blockIdx.x) * SLOT_W + 1, ai_local);
                    else spay[0] = ai_local;
                }
        }
        grid_barrier(counter, nb, target);
        double gmax = best, gmax2 = low, a_i;
        int i = bi;
        if (nb > 1) {
            gmax = NEG; gmax2 = NEG; i = SMO_NONE; a_i = 0.0;
            for (unsigned b = 0; b < nb; ++b) {
                const unsigned long long *s = slotI + ((size_t)par * SMO_MAXB + b) * SLOT_W;
                const double v = ld_f64(s + 0);
                const int id = (int)(unsigned)ld_u64(s + 2);
                gmax2 = fmax(gmax2, ld_f64(s + 3));
                if (id != SMO_NONE && (v > gmax || (v == gmax && id > i))) { gmax = v; i = id; a_i = ld_f64(s + 1); }
            }
        } else {
            a_i = spay[0];
        }
        viol = gmax + gmax2;
        if (i == SMO_NONE || !(gmax2 > NEG) || viol < tol) break;
        if (it >= max_iter) { status = 1; break; }
        const int si = i < n ? 1 : -1, ist = i < n ? i : i - n;
        // ---- j = argmin over I_low, -s G < gmax, of -(gmax + s G)^2 / (2 - 2 K_ij)
        const double *Ki = K + (int64_t)ist * n;
        double ki[SMO_E];
        double jb = NEG, jg = 0.0, ja = 0.0, jk = 0.0;
        int jx = SMO_NONE;
#pragma unroll
        for (int r = 0; r < SMO_E; ++r) {
            const int k = g0 + r * stride;
            ki[r] = k < n ? Ki[k] : 0.0;
            if (k < n) {
                const double gu = kb[r] + eps - yk[r], gd = -kb[r] + eps + yk[r];
                double q = 2.0 - 2.0 * ki[r];
                if (!(q > 0.0)) q = TAU;
                if (al[r] > 0.0 && -gu < gmax) {          // alpha_k in I_low
                    const double b = gmax + gu, o = (b * b) / q;
                    if (o > jb || (o == jb && k > jx)) { jb = o; jx = k; jg = gu; ja = al[r]; jk = ki[r]; }
                }
                if (as[r] < C && gd < gmax) {             // alpha*_k in I_low
                    const double b = gmax - gd, o = (b * b) / q;
                    if (o > jb || (o == jb && k + n > jx)) { jb = o; jx = k + n; jg = gd; ja = as[r]; jk = ki[r]; }
                }
            }
        }
        {
            const int mine = jx;
            block_argmax(jb, jx, skey, sidx);
            // the thread that holds the winner publishes its payload
            unsigned long long *s = slotJ + ((size_t)par * SMO_MAXB + blockIdx.x) * SLOT_W;
            if (mine == jx && jx != SMO_NONE) {
                if (nb > 1) { st_f64(s + 0, jb); st_f64(s + 1, ja); st_f64(s + 2, jg); st_f64(s + 3, jk); st_u64(s + 4, (unsigned long long)(unsigned)jx); }
                else { spay[1] = ja; spay[2] = jg; spay[3] = jk; }
            }
            if (nb > 1 && jx == SMO_NONE && threadIdx.x == 0) { st_f64(s + 0, NEG); st_u64(s + 4, (unsigned long long)(unsigned)SMO_NONE); }
        }
        grid_barrier(counter, nb, target);
        double a_j, G_j, K_ij;
        int j = jx;
        if (nb > 1) {
            double ob = NEG;
            j = SMO_NONE; a_j = G_j = K_ij = 0.0;
            for (unsigned b = 0; b < nb; ++b) {
                const unsigned long long *s = slotJ + ((size_t)par * SMO_MAXB + b) * SLOT_W;
                const double v = ld_f64(s + 0);
                const int id = (int)(unsigned)ld_u64(s + 4);
                if (id != SMO_NONE && (v > ob || (v == ob && id > j))) { ob = v; j = id; a_j = ld_f64(s + 1); G_j = ld_f64(s + 2); K_ij = ld_f64(s + 3); }
            }
        } else {
            a_j = spay[1]; G_j = spay[2]; K_ij = spay[3];
        }
        if (j == SMO_NONE) break;                        // no feasible direction left (libsvm: j == -1)
        const int sj = j < n ? 1 : -1, jst = j < n ? j : j - n;
        // ---- the two-variable subproblem (libsvm Solver::Solve), evaluated by every thread alike
        const double G_i = -(double)si * gmax;
        double quad = 2.0 - 2.0 * K_ij;
        if (!(quad > 0.0)) quad = TAU;
        double ni = a_i, nj = a_j;
        if (si != sj) {
            const double delta = (-G_i - G_j) / quad, diff = a_i - a_j;
            ni += delta; nj += delta;
            if (diff > 0.0) { if (nj < 0.0) { nj = 0.0; ni = diff; } }
            else { if (ni < 0.0) { ni = 0.0; nj = -diff; } }
            if (diff > 0.0) { if (ni > C) { ni = C; nj = C - diff; } }
            else { if (nj > C) { nj = C; ni = C + diff; } }
        } else {
            const double delta = (G_i - G_j) / quad, sum = a_i + a_j;
            ni -= delta; nj += delta;
            if (sum > C) { if (ni > C) { ni = C; nj = sum - C; } }
            else { if (nj < 0.0) { nj = 0.0; ni = sum; } }
            if (sum > C) { if (nj > C) { nj = C; ni = sum - C; } }
            else { if (ni < 0.0) { ni = 0.0; nj = sum; } }
        }
        const double dbi = (double)si * (ni - a_i), dbj = (double)sj * (nj - a_j);   // changes of beta at the two stations
        const double *Kj = K + (int64_t)jst * n;
#pragma unroll
        for (int r = 0; r < SMO_E; ++r) {
            const int k = g0 + r * stride;
            if (k < n) {
                kb[r] = kb[r] + ki[r] * dbi + Kj[k] * dbj;
                if (k == ist) { if (si > 0) al[r] = ni; else as[r] = ni; }
                if (k == jst) { if (sj > 0) al[r] = nj; else as[r] = nj; }
            }
        }
        __syncthreads();        // spay[] is rewritten by the next iteration
    }
#pragma unroll
    for (int r = 0; r < SMO_E; ++r) {
        const int k = g0 + r * stride;
        if (k < n) { al_out[k] = al[r]; as_out[k] = as[r]; kb_out[k] = kb[r]; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { out->iters = it; out->violation = viol; out->status = status; }
}

// libsvm Solver::calculate_rho on the 2n variables, one block, fixed order
__global__ __launch_bounds__(256) void svr_rho_kernel(const double *__restrict__ alpha, const double *__restrict__ alpha_s,
                                                      const double *__restrict__ kb, const double *__restrict__ y, int n,
                                                      double C, double eps, double *__restrict__ beta, SmoOut *out) {
    __shared__ double ssum[256], sub[256], slb[256];
    __shared__ int scnt[256];
    double sum = 0.0, ub = INFINITY, lb = -INFINITY;
    int cnt = 0;
    for (int k = threadIdx.x; k < n; k += 256) {
        const double al = alpha[k], as = alpha_s[k];
        beta[k] = al - as;
        const double gu = kb[k] + eps - y[k], gd = -kb[k] + eps + y[k];
        // alpha_k (s = +1): s G = gu ; alpha*_k (s = -1): s G = -gd
        if (al > 0.0 && al < C) { sum += gu; ++cnt; }
        else if (al <= 0.0) ub = fmin(ub, gu); else lb = fmax(lb, gu);
        if (as > 0.0 && as < C) { sum += -gd; ++cnt; }
        else if (as >= C) ub = fmin(ub, -gd); else lb = fmax(lb, -gd);
    }
    ssum[threadIdx.x] = sum; scnt[threadIdx.x] = cnt; sub[threadIdx.x] = ub; slb[threadIdx.x] = lb;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int t = 1; t < 256; ++t) { sum += ssum[t]; cnt += scnt[t]; ub = fmin(ub, sub[t]); lb = fmax(lb, slb[t]); }
        out->rho = cnt > 0 ? sum / (double)cnt : 0.5 * (ub + lb);
    }
}

// ------------------------------------------------------------------------------------------------ nnet (vmmin) --
constexpr int NN_T = 256;
constexpr int NN_H = 10;           // size = 10 (V73:249, V73:463)

__device__ __forceinline__ double nn_sigmoid(double z) {  // nnet.c sigmoid()
    if (z < -15.0) return 0.0;
    if (z > 15.0) return 1.0;
    return 1.0 / (1.0 + exp(-z));
}

struct NnOut { double value; int fncount, grcount, fail, pad; };

// value (and gradient when GRAD) of sum_k (yhat_k - y_k)^2 at the weights in LDS array w; the result is left in
// *sval / g[] (LDS) for every thread to read after the trailing barrier
template <int P, bool GRAD>
__device__ __forceinline__ void nn_eval(const double *__restrict__ X, const double *__restrict__ y, int n, const double *w,
                                        double *g, double *part, double *sval) {
    constexpr int NW = (P + 1) * NN_H + NN_H + 1;
    double ga[GRAD ? NW : 1];
    if (GRAD) {
#pragma unroll
        for (int q = 0; q < NW; ++q) ga[q] = 0.0;
    }
    double val = 0.0;
    for (int k = threadIdx.x; k < n; k += NN_T) {
        double x[P], h[NN_H];
#pragma unroll
        for (int j = 0; j < P; ++j) x[j] = X[(int64_t)k * P + j];
        double o = w[(P + 1) * NN_H];
#pragma unroll
        for (int u = 0; u < NN_H; ++u) {
            double z = w[u * (P + 1)];
#pragma unroll
            for (int j = 0; j < P; ++j) z = z + w[u * (P + 1) + 1 + j] * x[j];
            h[u] = nn_sigmoid(z);
            o = o + w[(P + 1) * NN_H + 1 + u] * h[u];
        }
        const double err = o - y[k];
        val = val + err * err;
        if (GRAD) {
            const double d = 2.0 * err;
            ga[(P + 1) * NN_H] += d;
#pragma unroll
            for (int u = 0; u < NN_H; ++u) {
                ga[(P + 1) * NN_H + 1 + u] += d * h[u];
                const double dz = d * w[(P + 1) * NN_H + 1 + u] * h[u] * (1.0 - h[u]);
                ga[u * (P + 1)] += dz;
#pragma unroll
                for (int j = 0; j < P; ++j) ga[u * (P + 1) + 1 + j] += dz * x[j];
            }
        }
    }
    // fixed-order sums: butterfly inside a wave, then the four waves in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) val = val + __shfl_xor(val, o);
    __syncthreads();
    if (lane == 0) part[wave] = val;
    if (GRAD) {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            double v = ga[q];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o);
            if (lane == 0) part[4 + wave * NW + q] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *sval = ((part[0] + part[1]) + part[2]) + part[3];
    if (GRAD && threadIdx.x < NW) {
        const int q = threadIdx.x;
        g[q] = ((part[4 + q] + part[4 + NW + q]) + part[4 + 2 * NW + q]) + part[4 + 3 * NW + q];
    }
    __syncthreads();
}

// R's vmmin (src/appl/optim.c), one block; B = packed lower triangle in LDS
template <int P>
__global__ __launch_bounds__(NN_T) void nnet_bfgs_kernel(const double *__restrict__ X, const double *__restrict__ y, int n,
                                                         double *__restrict__ wts, int maxit, double abstol, double reltol, NnOut *out) {
    constexpr int NW = (P + 1) * NN_H + NN_H + 1;
    static_assert(NW <= NN_T, "one thread per weight");
    extern __shared__ double sm[];
    double *b = sm, *g = b + NW, *t = g + NW, *c = t + NW, *Xs = c + NW, *Xc = Xs + NW, *part = Xc + NW, *sval = part + 4 + 4 * NW, *B = sval + 2;
    const double stepredn = 0.2, acctol = 1e-4, reltest = 10.0;
    const int q = threadIdx.x;
    if (q < NW) b[q] = wts[q];
    __syncthreads();
    nn_eval<P, true>(X, y, n, b, g, part, sval);
    double f = *sval, fmin = f;
    int funcount = 1, gradcount = 1, iter = 1, ilast = 1, count = 0, fail = 0;
    if (!(fabs(f) <= 1.79769313486231570815e308)) { if (q == 0) { out->value = f; out->fncount = 1; out->grcount = 1; out->fail = 2; } return; }
    if (maxit <= 0) { if (q == 0) { out->value = f; out->fncount = 0; out->grcount = 0; out->fail = 0; } return; }
    auto Bat = [&](int i, int j) -> double & { return i >= j ? B[i * (i + 1) / 2 + j] : B[j * (j + 1) / 2 + i]; };
    for (;;) {
        if (ilast == gradcount) {
            for (int e = q; e < NW * (NW + 1) / 2; e += NN_T) B[e] = 0.0;
            __syncthreads();
            if (q < NW) B[q * (q + 1) / 2 + q] = 1.0;
        }
        __syncthreads();
        if (q < NW) {
            Xs[q] = b[q]; c[q] = g[q];
            double s = 0.0;
            for (int j = 0; j < NW; ++j) s -= Bat(q, j) * g[j];
            t[q] = s;
        }
        __syncthreads();
        double gradproj = 0.0;
        for (int i = 0; i < NW; ++i) gradproj += t[i] * g[i];
        if (gradproj < 0.0) {                      // a descent direction
            double steplength = 1.0;
            bool accpoint = false;
            do {
                __syncthreads();
                if (q < NW) b[q] = Xs[q] + steplength * t[q];
                __syncthreads();
                count = 0;
                for (int i = 0; i < NW; ++i) if (reltest + Xs[i] == reltest + b[i]) ++count;
                if (count < NW) {
                    nn_eval<P, false>(X, y, n, b, g, part, sval);
                    f = *sval;
                    ++funcount;
                    accpoint = (fabs(f) <= 1.79769313486231570815e308) && (f <= fmin + gradproj * steplength * acctol);
                    if (!accpoint) steplength *= stepredn;
                }
            } while (!(count == NW || accpoint));
            const bool enough = (f > abstol) && fabs(f - fmin) > reltol * (fabs(fmin) + reltol);
            if (!enough) { count = NW; fmin = f; }
            if (count < NW) {
                fmin = f;
                nn_eval<P, true>(X, y, n, b, g, part, sval);
                ++gradcount; ++iter;
                if (q < NW) { t[q] = steplength * t[q]; c[q] = g[q] - c[q]; }
                __syncthreads();
                double D1 = 0.0;
                for (int i = 0; i < NW; ++i) D1 += t[i] * c[i];
                if (D1 > 0.0) {
                    if (q < NW) {
                        double s = 0.0;
                        for (int j = 0; j < NW; ++j) s += Bat(q, j) * c[j];
                        Xc[q] = s;
                    }
                    __syncthreads();
                    double D2 = 0.0;
                    for (int i = 0; i < NW; ++i) D2 += Xc[i] * c[i];
                    D2 = 1.0 + D2 / D1;
                    for (int e = q; e < NW * (NW + 1) / 2; e += NN_T) {
                        int i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
                        while (i * (i + 1) / 2 > e) --i;
                        while ((i + 1) * (i + 2) / 2 <= e) ++i;
                        const int j = e - i * (i + 1) / 2;
                        B[e] += (D2 * t[i] * t[j] - Xc[i] * t[j] - t[i] * Xc[j]) / D1;
                    }
                } else {
                    ilast = gradcount;             // D1 <= 0: restart with the identity
                }
            } else {                               // no progress
                if (ilast < gradcount) { count = 0; ilast = gradcount; }
            }
        } else {                                   // uphill search
            count = 0;
            if (ilast == gradcount) count = NW; else ilast = gradcount;
        }
        __syncthreads();
        if (iter >= maxit) break;
        if (gradcount - ilast > 2 * NW) ilast = gradcount;      // periodic restart
        if (count == NW && ilast == gradcount) break;
    }
    if (iter >= maxit) fail = 1;
    __syncthreads();
    if (q < NW) wts[q] = b[q];
    if (q == 0) { out->value = fmin; out->fncount = funcount; out->grcount = gradcount; out->fail = fail; }
}

template <int P>
static int launch_nnet_fit(const double *X, const double *y, int n, double *w, int maxit, double abstol, double reltol, NnOut *out, hipStream_t s) {
    constexpr int NW = (P + 1) * NN_H + NN_H + 1;
    const size_t bytes = sizeof(double) * (size_t)(6 * NW + 4 + 4 * NW + 2 + NW * (NW + 1) / 2);
    MHS_HIP(hipFuncSetAttribute((const void *)nnet_bfgs_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL((nnet_bfgs_kernel<P>), dim3(1), dim3(NN_T), bytes, s, X, y, n, w, maxit, abstol, reltol, out);
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

}  // namespace mhs

using namespace mhs;

extern "C" {

int mhs_svr_fit(const double *X, const double *y, int64_t n, int p, double sigma, double C, double epsilon, double tol,
                int64_t max_iter, double *beta, double *b, double *x_center, double *x_scale, double *y_center,
                double *y_scale, int64_t *n_iter) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(X && y && beta && b && x_center && x_scale && y_center && y_scale, "NULL argument");
    MHS_REQUIRE(n >= 2 && n <= (int64_t)SMO_T * SMO_E * SMO_MAXB && p >= 1 && p <= 64, "n or p out of range");
    MHS_REQUIRE(sigma > 0 && C > 0 && epsilon >= 0 && tol > 0, "sigma, C and tol must be positive, epsilon non-negative");
    // scaled = TRUE: columns and response to zero mean, unit standard deviation (n - 1)
    std::vector<double> Z((size_t)n * p), t((size_t)n);
    for (int j = 0; j < p; ++j) {
        const double *col = X + (size_t)j * n;
        double m = 0.0;
        for (int64_t k = 0; k < n; ++k) { MHS_REQUIRE(std::isfinite(col[k]), "non-finite predictor"); m += col[k]; }
        m /= (double)n;
        double ss = 0.0;
        for (int64_t k = 0; k < n; ++k) ss += (col[k] - m) * (col[k] - m);
        const double sd = sqrt(ss / (double)(n - 1));
        MHS_REQUIRE(sd > 0, "a predictor is constant");
        x_center[j] = m; x_scale[j] = sd;
        for (int64_t k = 0; k < n; ++k) Z[(size_t)k * p + j] = (col[k] - m) / sd;
    }
    {
        double m = 0.0;
        for (int64_t k = 0; k < n; ++k) { MHS_REQUIRE(std::isfinite(y[k]), "non-finite response"); m += y[k]; }
        m /= (double)n;
        double ss = 0.0;
        for (int64_t k = 0; k < n; ++k) ss += (y[k] - m) * (y[k] - m);
        const double sd = sqrt(ss / (double)(n - 1));
        MHS_REQUIRE(sd > 0, "the response is constant");
        *y_center = m; *y_scale = sd;
        for (int64_t k = 0; k < n; ++k) t[(size_t)k] = (y[k] - m) / sd;
    }
    hipStream_t s = ctx().stream;
    DevBuf<double> dZ, dt, dK, dbeta, dkb, dal, das;
    DevBuf<unsigned long long> dslots;
    DevBuf<unsigned long long> dcount;
    DevBuf<SmoOut> dout;
    MHS_HIP(dZ.alloc((size_t)n * p)); MHS_HIP(dt.alloc((size_t)n)); MHS_HIP(dK.alloc((size_t)n * n));
    MHS_HIP(dbeta.alloc((size_t)n)); MHS_HIP(dkb.alloc((size_t)n)); MHS_HIP(dal.alloc((size_t)n)); MHS_HIP(das.alloc((size_t)n));
    MHS_HIP(dslots.alloc((size_t)4 * SMO_MAXB * SLOT_W)); MHS_HIP(dcount.alloc(1)); MHS_HIP(dout.alloc(1));
    MHS_HIP(hipMemcpyAsync(dZ.p, Z.data(), sizeof(double) * Z.size(), hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemcpyAsync(dt.p, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemsetAsync(dslots.p, 0, sizeof(unsigned long long) * 4 * SMO_MAXB * SLOT_W, s));
    MHS_HIP(hipMemsetAsync(dcount.p, 0, sizeof(unsigned long long), s));
    MHS_HIP(hipMemsetAsync(dout.p, 0, sizeof(SmoOut), s));
    hipLaunchKernelGGL(rbf_gram_kernel, dim3((unsigned)(((int64_t)n * n + 255) / 256)), dim3(256), 0, s, dZ.p, (int)n, p, sigma, dK.p);
    MHS_HIP(hipGetLastError());
    const unsigned nb = (unsigned)((n + (int64_t)SMO_T * SMO_E - 1) / ((int64_t)SMO_T * SMO_E));
    {
        const double *Kp = dK.p, *yp = dt.p;
        int nn = (int)n;
        long long mi = max_iter > 0 ? (long long)max_iter : std::max<long long>(10000000LL, 100LL * n);
        unsigned long long *sl = dslots.p;
        unsigned long long *cn = dcount.p;
        double *ao = dal.p, *so = das.p, *ko = dkb.p;
        SmoOut *oo = dout.p;
        void *args[] = {&Kp, &yp, &nn, &C, &epsilon, &tol, &mi, &sl, &cn, &ao, &so, &ko, &oo};
        if (nb > 1) MHS_HIP(hipLaunchCooperativeKernel((const void *)svr_smo_kernel, dim3(nb), dim3(SMO_T), args, 0, s));
        else MHS_HIP(hipLaunchKernel((const void *)svr_smo_kernel, dim3(1), dim3(SMO_T), args, 0, s));
    }
    hipLaunchKernelGGL(svr_rho_kernel, dim3(1), dim3(256), 0, s, dal.p, das.p, dkb.p, dt.p, (int)n, C, epsilon, dbeta.p, dout.p);
    MHS_HIP(hipGetLastError());
    SmoOut h;
    MHS_HIP(hipMemcpyAsync(beta, dbeta.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipMemcpyAsync(&h, dout.p, sizeof(SmoOut), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    *b = h.rho;
    if (n_iter) *n_iter = h.iters;
    if (h.status != 0) { set_error("mhs_svr_fit: no convergence within %lld iterations (violation %.3g)", h.iters, h.violation); return MHS_ERR_NUMERIC; }
    return MHS_OK;
}

int mhs_nnet_fit(const double *X, const double *y, int64_t n, int p, int size, double *wts, int maxit, double abstol,
                 double reltol, double *value, int *counts, int *fail) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(X && y && wts && n >= 1 && n < (1LL << 31), "bad arguments");
    MHS_REQUIRE(size == NN_H, "this build fits nnet(size = 10) only (V73:249, V73:463)");
    MHS_REQUIRE(p >= 1 && p <= 12, "p must be between 1 and 12");
    const int NW = (p + 1) * NN_H + NN_H + 1;
    // rows in the kernel's order (row-major) from R's column-major matrix
    std::vector<double> Xr((size_t)n * p);
    for (int j = 0; j < p; ++j)
        for (int64_t k = 0; k < n; ++k) {
            const double v = X[(size_t)j * n + k];
            MHS_REQUIRE(std::isfinite(v), "non-finite predictor");
            Xr[(size_t)k * p + j] = v;
        }
    for (int64_t k = 0; k < n; ++k) MHS_REQUIRE(std::isfinite(y[k]), "non-finite response");
    hipStream_t s = ctx().stream;
    DevBuf<double> dX, dy, dw;
    DevBuf<NnOut> dout;
    MHS_HIP(dX.alloc(Xr.size())); MHS_HIP(dy.alloc((size_t)n)); MHS_HIP(dw.alloc((size_t)NW)); MHS_HIP(dout.alloc(1));
    MHS_HIP(hipMemcpyAsync(dX.p, Xr.data(), sizeof(double) * Xr.size(), hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemcpyAsync(dy.p, y, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemcpyAsync(dw.p, wts, sizeof(double) * (size_t)NW, hipMemcpyHostToDevice, s));
    int rc = MHS_OK;
    switch (p) {
#define MHS_NNF(P_) case P_: rc = launch_nnet_fit<P_>(dX.p, dy.p, (int)n, dw.p, maxit, abstol, reltol, dout.p, s); break;
        MHS_NNF(1) MHS_NNF(2) MHS_NNF(3) MHS_NNF(4) MHS_NNF(5) MHS_NNF(6) MHS_NNF(7) MHS_NNF(8) MHS_NNF(9) MHS_NNF(10) MHS_NNF(11) MHS_NNF(12)
#undef MHS_NNF
    }
    if (rc) return rc;
    NnOut h;
    MHS_HIP(hipMemcpyAsync(wts, dw.p, sizeof(double) * (size_t)NW, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipMemcpyAsync(&h, dout.p, sizeof(NnOut), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    if (h.fail == 2) { set_error("mhs_nnet_fit: the initial value is not finite"); return MHS_ERR_NUMERIC; }
    if (value) *value = h.value;
    if (counts) { counts[0] = h.fncount; counts[1] = h.grcount; }
    if (fail) *fail = h.fail;
    return MHS_OK;
}

}  // extern "C"
