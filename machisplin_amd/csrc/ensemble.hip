// Per-cell evaluation of the six ensemble members on gfx950 and their weighted sum:
// the terra::predict(rast_stack, model) loop of machisplin.mltps Step 2 (V73:447-619).
//
// Every kernel reads the C covariate planes once per cell (coalesced along columns),
// generates LONG/LAT from the grid affine, and writes  out (+)= pred * weight.
// Regimes: lm / nnet / earth are HBM-bound (a few dozen flops per cell);
// ksvm is FP64-VALU bound (nSV exp-pairs per cell, support vectors through the scalar
// cache, exp from a 64-entry 2^(j/64) table in LDS); gbm / randomForest are LDS-latency
// bound tree walks (node records staged chunk-wise in LDS, predictors parked in LDS so a
// lane can index them by the node's split variable).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"
#include "devmath.h"

#include "ensemble_int.h"

namespace mhs {

// ------------------------------------------------------------------------- lm --
__global__ __launch_bounds__(256) void lm_kernel(const double *__restrict__ coef, int p, StackDev s,
                                                 PredGeom g, double weight, int accumulate,
                                                 double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)g.nr * g.nc) return;
    const int row = (int)(i / g.nc), col = (int)(i - (int64_t)row * g.nc);
    double acc = coef[0];
    for (int j = 0; j < p; ++j) acc = acc + coef[j + 1] * predictor(s, g, j, row, col);
    emit(out, (int64_t)row * g.ld_out + col, acc, weight, accumulate);
}

// ----------------------------------------------------------------------- nnet --
__device__ __forceinline__ double nnet_sigmoid(double z) {  // nnet.c sigmoid()
    if (z < -15.0) return 0.0;
    if (z > 15.0) return 1.0;
    return 1.0 / (1.0 + exp(-z));
}

template <int P>
__global__ __launch_bounds__(256) void nnet_kernel(const double *__restrict__ w, int H, double y_scale,
                                                   double y_shift, StackDev s, PredGeom g,
                                                   double weight, int accumulate,
                                                   double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)g.nr * g.nc) return;
    const int row = (int)(i / g.nc), col = (int)(i - (int64_t)row * g.nc);
    double x[P];
    bool na = false;
#pragma unroll
    for (int j = 0; j < P; ++j) { x[j] = predictor(s, g, j, row, col); na |= isnan(x[j]); }
    double acc = w[(P + 1) * H];
    for (int h = 0; h < H; ++h) {
        const double *wh = w + h * (P + 1);
        double z = wh[0];
#pragma unroll
        for (int j = 0; j < P; ++j) z = z + wh[1 + j] * x[j];
        acc = acc + w[(P + 1) * H + 1 + h] * nnet_sigmoid(z);
    }
    acc = acc * y_scale + y_shift;
    emit(out, (int64_t)row * g.ld_out + col, na ? NAN : acc, weight, accumulate);
}

// ---------------------------------------------------------------------- earth --
// terms as factor lists: term k owns factors tstart[k] .. tstart[k+1]-1, each (var, dir, cut)
__global__ __launch_bounds__(256) void earth_kernel(const double *__restrict__ coef,
                                                    const double *__restrict__ fcut,
                                                    const int *__restrict__ tstart,
                                                    const int *__restrict__ fvar,
                                                    const int *__restrict__ fdir, int nterms, int p,
                                                    StackDev s, PredGeom g, double weight, int accumulate,
                                                    double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)g.nr * g.nc) return;
    const int row = (int)(i / g.nc), col = (int)(i - (int64_t)row * g.nc);
    bool na = false;
    for (int j = 0; j < p; ++j) na |= isnan(predictor(s, g, j, row, col));
    double acc = 0.0;
    for (int k = 0; k < nterms; ++k) {
        double term = 1.0;
        for (int f = tstart[k]; f < tstart[k + 1]; ++f) {
            const double xv = predictor(s, g, fvar[f], row, col);
            const int d = fdir[f];
            const double fac = d == 2 ? xv : (d == 1 ? fmax(0.0, xv - fcut[f]) : fmax(0.0, fcut[f] - xv));
            term = term * fac;
        }
        acc = acc + coef[k] * term;
    }
    emit(out, (int64_t)row * g.ld_out + col, na ? NAN : acc, weight, accumulate);
}

// ------------------------------------------------- lm + nnet + earth in ONE pass --
// The three HBM-bound members are consecutive in the reference's model order (g, n, m -- V73:340-362): evaluated in
// one kernel the planes are read once and the output plane is read and written once (28 B per cell instead of 3 x
// 28), with exactly the arithmetic and accumulation order of the three separate kernels (bit-identical planes).
struct SmallArgs {
    const double *lm_coef;                                             // NULL: member absent
    const double *nn_w; int nn_H; double nn_scale, nn_shift;           // nn_w NULL: absent
    const double *ea_coef; const double *ea_cut; const int *ea_tstart, *ea_fvar, *ea_fdir; int ea_nterms;   // ea_coef NULL: absent
    double w_lm, w_nn, w_ea;
};
template <int P>
__global__ __launch_bounds__(256) void small_members_kernel(SmallArgs a, StackDev s, PredGeom g, int accumulate,
                                                            double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)g.nr * g.nc) return;
    const int row = (int)(i / g.nc), col = (int)(i - (int64_t)row * g.nc);
    double x[P];
    bool na = false;
#pragma unroll
    for (int j = 0; j < P; ++j) { x[j] = predictor(s, g, j, row, col); na |= isnan(x[j]); }
    const int64_t o = (int64_t)row * g.ld_out + col;
    double v = accumulate ? out[o] : 0.0;
    bool first = !accumulate;
    auto add = [&](double pred, double w) { const double t = pred * w; v = first ? t : v + t; first = false; };
    if (a.lm_coef) {
        double acc = a.lm_coef[0];
#pragma unroll
        for (int j = 0; j < P; ++j) acc = acc + a.lm_coef[j + 1] * x[j];
        add(acc, a.w_lm);
    }
    if (a.nn_w) {
        double acc = a.nn_w[(P + 1) * a.nn_H];
        for (int h = 0; h < a.nn_H; ++h) {
            const double *wh = a.nn_w + h * (P + 1);
            double z = wh[0];
#pragma unroll
            for (int j = 0; j < P; ++j) z = z + wh[1 + j] * x[j];
            acc = acc + a.nn_w[(P + 1) * a.nn_H + 1 + h] * nnet_sigmoid(z);
        }
        acc = acc * a.nn_scale + a.nn_shift;
        add(na ? NAN : acc, a.w_nn);
    }
    if (a.ea_coef) {
        double acc = 0.0;
        for (int k = 0; k < a.ea_nterms; ++k) {
            double term = 1.0;
            for (int f = a.ea_tstart[k]; f < a.ea_tstart[k + 1]; ++f) {
                const double xv = predictor(s, g, a.ea_fvar[f], row, col);
                const int d = a.ea_fdir[f];
                const double fac = d == 2 ? xv : (d == 1 ? fmax(0.0, xv - a.ea_cut[f]) : fmax(0.0, a.ea_cut[f] - xv));
                term = term * fac;
            }
            acc = acc + a.ea_coef[k] * term;
        }
        add(na ? NAN : acc, a.w_ea);
    }
    out[o] = v;
}

// ------------------------------------------------------------------------ svr --
// exp(-700 u) for u in [0, 1] (u = -x / 700: the factor is folded into the per-support-vector coefficients, and
// the range check is the free `clamp` output modifier of the fma that produces u).  In units of ln2/4096,
// y = -700 u 4096/ln2 = 4096 e + j + r, |r| <= 1/2; 2^(j/4096) from a 32 KB LDS table, exp(r ln2/4096) =
// 1 + c r + c^2 r^2 / 2 (truncation < 1e-13 relative).  The magic constant carries the exponent bias, so the low
// word of t is k = (e + 1023) 4096 + j and k << 8 has the finished exponent field on top and j in bits 8..19:
// one bit-field extract gives the table's byte offset and one and-or drops the exponent onto the table entry,
// which is stored as its mantissa bits only.  7 FP64-rate and 3 integer instructions (16 for a 64-entry
// table with a quartic).
constexpr int EXP_TAB_BITS = 12;
constexpr int EXP_TAB_N = 1 << EXP_TAB_BITS;
constexpr double EXP_SCALE = 4096.0 / 0.6931471805599453094;   // 4096 / ln 2
constexpr double EXP_RANGE = 700.0;                            // arguments below -700 count as -700 (1e-304)
__device__ __forceinline__ double table_exp_neg_acc(double u, const double *tab, double acc) {   // acc + exp(-700 u)
    // k = round(y) by the 1.5*2^52 trick: the integer lands in the low word of t (no v_rndne / v_cvt),
    // kd = t - magic is its exact double; y itself only ever exists inside the two fmas
    const double MAGIC = 0x1.8p52 + 1023.0 * EXP_TAB_N, NK = -EXP_RANGE * EXP_SCALE;
    const double t = fma(u, NK, MAGIC);
    const double kd = t - MAGIC;
    unsigned k8 = (unsigned)__double2loint(t) << 8;   // k >= 1023 * 4096 - 700 * 5910 > 0
    asm("" : "+v"(k8));   // or the extract below is rewritten as a shift and a mask of t's low word
    const double r = fma(u, NK, -kd);
    const double mj = *(const double *)((const char *)tab + __builtin_amdgcn_ubfe(k8, 5, EXP_TAB_BITS + 3));
    const double sj = __hiloint2double((int)((k8 & 0xfff00000u) | (unsigned)__double2hiint(mj)), __double2loint(mj));
    const double C1 = 1.0 / EXP_SCALE, C2 = 0.5 / (EXP_SCALE * EXP_SCALE);
    double q = fma(r, C2, C1);
    q = fma(q, r, 1.0);
    return fma(sj, q, acc);
}

// per support vector: [b_0 .. b_{P-1}, a], b_k = -2 sigma sv_k / 700, a = (sigma |sv|^2 - ln(|alpha| / amax)) / 700:
// u = a + sigma |x|^2 / 700 + b.x = (sigma |x - sv|^2 - ln(|alpha| / amax)) / 700 >= 0 and exp(-700 u) =
// |alpha| / amax K(x, sv).  With the coefficient inside the exponent the term is accumulated by the fma that
// finishes the exponential (acc += 2^e 2^(j/4096) * q); the support vectors with alpha > 0 come first, the
// others are summed separately and subtracted.
template <int P, int R>
__global__ __launch_bounds__(256) void svr_kernel(const double *__restrict__ svp, int nsv, int stride,
                                                  int npos, const double *__restrict__ xcs,
                                                  const double *__restrict__ gtab, double sigma, double b,
                                                  double amax,
                                                  double y_center, double y_scale, StackDev s, PredGeom g,
                                                  double weight, int accumulate, double *__restrict__ out) {
    __shared__ double etab[EXP_TAB_N];
    for (int i = threadIdx.x; i < EXP_TAB_N; i += 256) {   // mantissa bits of 2^(j/4096)
        const double e = gtab[i];
        etab[i] = __hiloint2double(__double2hiint(e) & 0x000fffff, __double2loint(e));
    }
    __syncthreads();
    const int64_t total = (int64_t)g.nr * g.nc;
    const int64_t half = (total + R - 1) / R;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i0 >= half) return;
    double x[R][P], q[R], acc[R];
    bool na[R];
    int row[R], col[R];
#pragma unroll
    for (int c = 0; c < R; ++c) {
        int64_t i = i0 + c * half;
        if (i >= total) i = total - 1;
        row[c] = (int)(i / g.nc); col[c] = (int)(i - (int64_t)row[c] * g.nc);
        na[c] = false; q[c] = 0.0; acc[c] = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double xv = predictor(s, g, j, row[c], col[c]);
            na[c] |= isnan(xv);
            x[c][j] = (xv - xcs[j]) / xcs[P + j];
            q[c] = fma(x[c][j], x[c][j], q[c]);
        }
        q[c] = (sigma / EXP_RANGE) * q[c];
    }
    double accn[R];
#pragma unroll
    for (int c = 0; c < R; ++c) accn[c] = 0.0;
    auto sum_range = [&](int v0, int v1, double (&a)[R]) {
        for (int v = v0; v < v1; ++v) {
            const double *sp = svp + (int64_t)v * stride;
#pragma unroll
            for (int c = 0; c < R; ++c) {
                // the LAST predictor first: in grid mode it is LAT, constant along a raster row -- svr_rt_kernel forms this
                // term once per row and support vector with the same fma, so the two kernels agree bit for bit
                double arg = q[c] + fma(sp[P - 1], x[c][P - 1], sp[P]);
#pragma unroll
                for (int j = 0; j < P - 1; ++j) arg = fma(sp[j], x[c][j], arg);
                arg = fmin(fmax(arg, 0.0), 1.0);   // folds into the clamp modifier of the last fma
                a[c] = table_exp_neg_acc(arg, etab, a[c]);
            }
        }
    };
    sum_range(0, npos, acc);
    sum_range(npos, nsv, accn);
#pragma unroll
    for (int c = 0; c < R; ++c) {
        const int64_t i = i0 + c * half;
        if (i < total) {
            const double pred = ((acc[c] - accn[c]) * amax - b) * y_scale + y_center;
            emit(out, (int64_t)row[c] * g.ld_out + col[c], na[c] ? NAN : pred, weight, accumulate);
        }
    }
}

// ROW-TILE form of the ksvm kernel (round 3): a wave = 64 R consecutive cells of ONE raster row.  LAT is the last
// predictor of every model (V73:127-138) and the same for all of the wave's cells, so its term b_LAT x_LAT + a of the
// exponent is formed ONCE per wave and support vector -- 128 vectors at a time, two per lane, parked in a wave-private
// 1 KB of LDS and read back as a broadcast -- instead of once per cell: 14 VALU instructions per (cell, SV) instead of
// 15.  A wave that does NOT fold its cells' q (below) adds exactly what svr_kernel adds, bit for bit; a folding wave's plane
// agrees with svr_kernel's to ~2e-13 of the prediction (tests: test_ksvm_row_tile_kernel_*).
template <int P, int R>
__global__ __launch_bounds__(256) void svr_rt_kernel(const double *__restrict__ svp, int nsv, int stride,
                                                     int npos, const double *__restrict__ xcs,
                                                     const double *__restrict__ gtab, double sigma, double b,
                                                     double amax, double y_center, double y_scale, StackDev s, PredGeom g,
                                                     int tiles_per_row, double weight, int accumulate, double *__restrict__ out) {
    constexpr int CH = 128;
    __shared__ double etab[EXP_TAB_N];
    __shared__ double aw[4][CH];
    for (int i = threadIdx.x; i < EXP_TAB_N; i += 256) {   // mantissa bits of 2^(j/4096)
        const double e = gtab[i];
        etab[i] = __hiloint2double(__double2hiint(e) & 0x000fffff, __double2loint(e));
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ntiles = (int64_t)g.nr * tiles_per_row;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const int row = (int)(tile / tiles_per_row), tcol = (int)(tile - (int64_t)row * tiles_per_row) * (64 * R);
    double x[R][P], q[R], acc[R], accn[R];
    bool na[R], ok[R];
    int col[R];
#pragma unroll
    for (int c = 0; c < R; ++c) {
        const int cc = tcol + c * 64 + lane;
        ok[c] = cc < g.nc;
        col[c] = min(cc, g.nc - 1);
        na[c] = false; q[c] = 0.0; acc[c] = 0.0; accn[c] = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const double xv = predictor(s, g, j, row, col[c]);
            na[c] |= isnan(xv);
            x[c][j] = (xv - xcs[j]) / xcs[P + j];
            q[c] = fma(x[c][j], x[c][j], q[c]);
        }
        q[c] = (sigma / EXP_RANGE) * q[c];
    }
    const double xlat = x[0][P - 1];                       // the row's scaled LAT: the same value in every lane and cell
    // The cell's own term q = sigma |x|^2 / 700 of the exponent is the one addition per (cell, SV) that is left beside the
    // P - 1 fmas.  With S the LARGEST q of the wave's cells, u + (S - q_c) is still >= 0 and is formed with S folded into the
    // per-wave term above; the cell's sum then comes out scaled by exp(-700 (S - q_c)), which ONE exact exponential per cell
    // undoes at the end -- 13 VALU instructions per pair instead of 14.  Neighbouring cells: S - q_c is a few hundredths.
    // Where the wave's q spread more than 0.5 (terms would be flushed that the cell's scale brings back) the wave keeps
    // the per-pair addition.
    double qhi = -1.0, qlo = 2e300;
#pragma unroll
    for (int c = 0; c < R; ++c) { qhi = fmax(qhi, q[c]); qlo = fmin(qlo, q[c]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qhi = fmax(qhi, __shfl_xor(qhi, o)); qlo = fmin(qlo, __shfl_xor(qlo, o)); }
    // (qhi, qlo are the same in every lane after the butterfly; readfirstlane tells the COMPILER so: with a wave-uniform
    // branch the support-vector loop keeps its counter and the LDS cursor on the scalar unit -- round 5: the loop spent 3 of
    // its 42 vector instructions per support vector on them, 14.0 -> 13.1 per (cell, SV))
    const bool fold = __builtin_amdgcn_readfirstlane((int)(qhi - qlo <= 0.5)) != 0;   // false too when every cell of the wave is NA (qhi = -1, qlo = 2e300)
    const double S = fold ? qhi : 0.0;
    auto sum_range = [&](int v0, int v1, double (&a)[R], auto folded) {
        constexpr bool FOLD = decltype(folded)::value;
        auto pair = [&](const double *sp, const double ap) {
#pragma unroll
            for (int c = 0; c < R; ++c) {
                double arg = FOLD ? ap : q[c] + ap;
#pragma unroll
                for (int j = 0; j < P - 1; ++j) arg = fma(sp[j], x[c][j], arg);
                arg = fmin(fmax(arg, 0.0), 1.0);
                a[c] = table_exp_neg_acc(arg, etab, a[c]);
            }
        };
        for (int vb = v0; vb < v1; vb += CH) {
            const int n = min(CH, v1 - vb);
            __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < n; e += 64) {
                const double *sp = svp + (int64_t)(vb + e) * stride;
                aw[wave][e] = fma(sp[P - 1], xlat, sp[P]) + S;
            }
            __builtin_amdgcn_wave_barrier();
            const double *sp = svp + (int64_t)vb * stride;
            const double *ar = aw[wave];
            int e = 0;
            for (; e + 4 <= n; e += 4) {          // four support vectors per trip: one cursor bump, immediate LDS offsets
#pragma unroll
                for (int k = 0; k < 4; ++k) pair(sp + (int64_t)k * stride, ar[e + k]);
                sp += (int64_t)4 * stride;
            }
            for (; e < n; ++e) { pair(sp, ar[e]); sp += stride; }
        }
    };
    if (fold) {
        sum_range(0, npos, acc, std::true_type());
        sum_range(npos, nsv, accn, std::true_type());
    } else {
        sum_range(0, npos, acc, std::false_type());
        sum_range(npos, nsv, accn, std::false_type());
    }
#pragma unroll
    for (int c = 0; c < R; ++c)
        if (ok[c]) {
            const double back = fold ? exp(EXP_RANGE * (S - q[c])) : 1.0;
            const double pred = ((acc[c] - accn[c]) * back * amax - b) * y_scale + y_center;
            emit(out, (int64_t)row * g.ld_out + col[c], na[c] ? NAN : pred, weight, accumulate);
        }
}

// ---------------------------------------------------------------------- trees --
// gbm_pred / predictRegTree walks.  Dynamic LDS: xs[p][R*256] predictors, then one chunk
// of node records (LDS_NODES) or nothing (nodes read from global, for trees too large).
// SPLIT (few cells, e.g. the station rows): blockIdx.y owns a contiguous share of the LDS chunks and writes its
// partial tree sum to out[blockIdx.y * total + cell] (NaN marks an NA row of a forest); tree_finalize_kernel adds
// the shares in order.  Without it the 5 000 station rows of a step occupy 10 blocks for 10 000 trees' worth of time.
template <bool GBM, bool LDS_NODES, int R, bool NA_ONLY, bool SPLIT = false>
__global__ __launch_bounds__(256) void tree_kernel(const Node *__restrict__ gnodes,
                                                   const int *__restrict__ tree_off,
                                                   const TreeChunk *__restrict__ chunks, int n_chunks,
                                                   int n_trees, double init_f, int p, StackDev s,
                                                   PredGeom g, double weight, int accumulate,
                                                   double *__restrict__ out, const unsigned *__restrict__ na_gate = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (NA_ONLY && na_gate && na_gate[1] == 0u) return;          // the compacted NA list held every NA cell: nothing left to do here
    double *xs = (double *)smem;                                  // [p][R*256]
    Node *lnodes = (Node *)(smem + (size_t)p * R * 256 * sizeof(double));
    const int64_t total = (int64_t)g.nr * g.nc;
    const int64_t half = (total + R - 1) / R;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int row[R], col[R];
    bool na[R];
    double acc[R];
#pragma unroll
    for (int c = 0; c < R; ++c) {
        int64_t i = i0 + c * half;
        if (i >= total) i = total - 1;
        if (i < 0) i = 0;
        row[c] = (int)(i / g.nc); col[c] = (int)(i - (int64_t)row[c] * g.nc);
        na[c] = false; acc[c] = 0.0;
        for (int j = 0; j < p; ++j) {
            const double xv = predictor(s, g, j, row[c], col[c]);
            na[c] |= isnan(xv);
            xs[(j * R + c) * 256 + threadIdx.x] = xv;
        }
    }
    if (NA_ONLY) {  // fix-up pass after gbm_lut_kernel: only cells holding an NA covariate are walked
        bool any = false;
#pragma unroll
        for (int c = 0; c < R; ++c) any |= na[c] && i0 < half && (i0 + c * half) < total;
        if (!__syncthreads_or(any)) return;
    }
    int ch_begin = 0, ch_end = n_chunks;
    if (SPLIT) {
        const int per = (n_chunks + (int)gridDim.y - 1) / (int)gridDim.y;
        ch_begin = min((int)blockIdx.y * per, n_chunks);
        ch_end = min(ch_begin + per, n_chunks);
    }
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const TreeChunk tc = chunks[ch];
        if (LDS_NODES) {
            __syncthreads();
            const int4 *src = (const int4 *)(gnodes + tc.node_begin);
            int4 *dst = (int4 *)lnodes;
            for (int e = threadIdx.x; e < tc.node_count; e += 256) dst[e] = src[e];
            __syncthreads();
        }
        for (int t = tc.first_tree; t < tc.first_tree + tc.n_trees; ++t) {
            const int tbase = LDS_NODES ? tree_off[t] - tc.node_begin : tree_off[t];
#pragma unroll
            for (int c = 0; c < R; ++c) {
                if (NA_ONLY && !na[c]) continue;
                Node nd;
                if constexpr (LDS_NODES) nd = lnodes[tbase]; else nd = gnodes[tbase];
                while (nd.var >= 0) {
                    const double xv = xs[(nd.var * R + c) * 256 + threadIdx.x];
                    unsigned nxt;
                    if (GBM) nxt = isnan(xv) ? nd.missing : (xv < nd.val ? nd.left : nd.right);
                    else nxt = (xv <= nd.val) ? nd.left : nd.right;
                    if constexpr (LDS_NODES) nd = lnodes[tbase + nxt]; else nd = gnodes[tbase + nxt];
                }
                acc[c] = acc[c] + nd.val;
            }
        }
    }
    if (SPLIT) {
#pragma unroll
        for (int c = 0; c < R; ++c) {
            const int64_t i = i0 + c * half;
            if (i0 < half && i < total) out[(int64_t)blockIdx.y * total + i] = (!GBM && na[c]) ? NAN : acc[c];
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < R; ++c) {
        const int64_t i = i0 + c * half;
        if (i0 < half && i < total && (!NA_ONLY || na[c])) {
            double pred;
            if (GBM) pred = init_f + acc[c];
            else pred = na[c] ? NAN : acc[c] / (double)n_trees;
            emit(out, (int64_t)row[c] * g.ld_out + col[c], pred, weight, accumulate);
        }
    }
}


// NA cells of a gbm window (round 4).  The predicate-LUT kernels skip cells with an NA covariate; gbm routes those through
// its MissingNode children (gbm_pred).  The strided NA-only pass above walks all 10 000 trees in every BLOCK that holds a
// single NA cell among its 1 024 -- on the reference's bundled rasters (0.09 % NoData along the coast) that was 214 of
// 401 ms per 1e8 cells.  Now: one pass over the planes appends the NA cells to a list (na_collect_kernel), and the walk
// runs over the list, 256 NA cells per block (gbm_na_list_kernel).  A window with more NA cells than the list holds sets
// the overflow word and the strided pass takes over.
__global__ __launch_bounds__(256) void na_collect_kernel(StackDev s, PredGeom g, unsigned cap, unsigned *__restrict__ list) {
    const int64_t total = (int64_t)g.nr * g.nc;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int row = (int)(i / g.nc), col = (int)(i - (int64_t)row * g.nc);
    bool na = false;
    for (int j = 0; j < s.C; ++j) na |= isnan(predictor(s, g, j, row, col));
    if (na) {
        const unsigned slot = atomicAdd(&list[0], 1u);
        if (slot < cap) list[2 + slot] = (unsigned)i;
        else list[1] = 1u;
    }
}
template <bool LDS_NODES>
__global__ __launch_bounds__(256) void gbm_na_list_kernel(const Node *__restrict__ gnodes, const int *__restrict__ tree_off,
                                                          const TreeChunk *__restrict__ chunks, int n_chunks, double init_f, int p,
                                                          StackDev s, PredGeom g, double weight, int accumulate,
                                                          double *__restrict__ out, const unsigned *__restrict__ list, unsigned cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (list[1]) return;                                          // overflow: the strided pass does the work
    double *xs = (double *)smem;                                  // [p][256]
    Node *lnodes = (Node *)(smem + (size_t)p * 256 * sizeof(double));
    const unsigned count = min(list[0], cap);
    for (unsigned base = blockIdx.x * 256u; base < count; base += gridDim.x * 256u) {
        const bool live = base + threadIdx.x < count;
        const int64_t i = list[2 + (live ? base + threadIdx.x : base)];
        const int row = (int)(i / g.nc), col = (int)(i - (int64_t)row * g.nc);
        for (int j = 0; j < p; ++j) xs[j * 256 + threadIdx.x] = predictor(s, g, j, row, col);
        double acc = 0.0;
        for (int ch = 0; ch < n_chunks; ++ch) {
            const TreeChunk tc = chunks[ch];
            if (LDS_NODES) {
                __syncthreads();
                const int4 *src = (const int4 *)(gnodes + tc.node_begin);
                int4 *dst = (int4 *)lnodes;
                for (int e = threadIdx.x; e < tc.node_count; e += 256) dst[e] = src[e];
                __syncthreads();
            }
            for (int t = tc.first_tree; t < tc.first_tree + tc.n_trees; ++t) {
                const int tbase = LDS_NODES ? tree_off[t] - tc.node_begin : tree_off[t];
                Node nd;
                if constexpr (LDS_NODES) nd = lnodes[tbase]; else nd = gnodes[tbase];
                while (nd.var >= 0) {
                    const double xv = xs[nd.var * 256 + threadIdx.x];
                    const unsigned nxt = isnan(xv) ? nd.missing : (xv < nd.val ? nd.left : nd.right);
                    if constexpr (LDS_NODES) nd = lnodes[tbase + nxt]; else nd = gnodes[tbase + nxt];
                }
                acc = acc + nd.val;
            }
        }
        if (live) emit(out, (int64_t)row * g.ld_out + col, init_f + acc, weight, accumulate);
        __syncthreads();
    }
}

// ---------------------------------------------------------- gbm: predicate LUT --
// Trees grown with interaction.depth = 5 (V73:493) have <= 5 splits, so a tree is a function
// of its S <= 6 split predicates: leaf = LUT[b], b = the S predicate bits.  Per tree a wave
// evaluates S wave-uniform predicates (threshold and variable arrive through the scalar
// cache) instead of walking nodes lane by lane:
//   * every predictor is mapped to an order-preserving float KEY once per cell, thresholds
//     to key space on the host, so that  x < split  <=>  key < tkey  EXACTLY:
//       float32/int16 planes: key = the value, tkey = smallest float >= split;
//       LONG: key = column index, tkey = #columns whose centre is < split (same double formula);
//       LAT : key = -row index,   tkey = 0.5 - (first row whose centre is < split);
//   * the keys are then replaced by their RANK among the model's sorted distinct tkeys of that
//     predictor (rank = #{tkeys <= key}, found once per cell by a coarse LDS + fine global binary
//     search), and a split on the j-th tkey carries c = j + 1:  key < tkey_j  <=>  rank <= j
//     <=>  clamp(c - rank, 0, 1) = 1.  Ranks are small integers, exact in float32, so a
//     predicate and its accumulation into the leaf index are two PACKED-f32 instructions for
//     two cells (v_pk_add_f32 with the clamp modifier, v_pk_fma_f32 acc = 2 acc + bit) instead
//     of compare + add-with-carry per cell: half the VALU issue slots of the kernel's hot loop;
//   * -rank is parked in LDS as [var][lane][4 cells] so one ds_read_b128 at a wave-uniform var
//     offset fetches the keys of the lane's 4 cells.  The accumulator starts at 2^(23-S) + t, so
//     after S doublings its bit pattern is 0x4B000000 + (t << S) + b: the tree's LUT slot, turned
//     into an LDS address by one shift-add; the leaf comes from the tree's 2^S-entry LUT (one
//     256-byte LDS row for S = 5, conflict-free).
// Cells with an NA covariate are skipped here and walked through their MissingNode
// children by tree_kernel<GBM, .., NA_ONLY> afterwards (gbm_pred's NA routing).

// bit[c] = clamp(c_q - rank[c], 0, 1) for the lane's 4 cells (k holds -rank); HI selects which
// dword of the SGPR pair {c_q, c_q+1} is broadcast to both halves of the packed add
template <bool HI>
__device__ __forceinline__ void pred_bits(float2v &b01, float2v &b23, const float2v k01, const float2v k23,
                                          const unsigned long long cpair) {
    if (!HI)
        asm("v_pk_add_f32 %0, %2, %4 op_sel_hi:[1,0] clamp\n\t"
            "v_pk_add_f32 %1, %3, %4 op_sel_hi:[1,0] clamp"
            : "=&v"(b01), "=&v"(b23) : "v"(k01), "v"(k23), "s"(cpair));
    else
        asm("v_pk_add_f32 %0, %2, %4 op_sel:[0,1] op_sel_hi:[1,1] clamp\n\t"
            "v_pk_add_f32 %1, %3, %4 op_sel:[0,1] op_sel_hi:[1,1] clamp"
            : "=&v"(b01), "=&v"(b23) : "v"(k01), "v"(k23), "s"(cpair));
}


template <int S>
__global__ __launch_bounds__(256) void gbm_lut_kernel(const double *__restrict__ lut,
                                                      const int *__restrict__ meta,
                                                      const void *__restrict__ sorted, int key64,
                                                      const int *__restrict__ sorted_off, int n_trees_padded,
                                                      double init_f, int p, StackDev s, PredGeom g,
                                                      double weight, int accumulate,
                                                      double *__restrict__ out) {
    static_assert((LUT_CHUNK << S) * sizeof(double) >= LUT_COARSE * sizeof(float), "coarse table must fit");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *keys = (float *)smem;                                              // [p][256][4]
    double *slut = (double *)(smem + (size_t)p * 256 * LUT_R * sizeof(float));  // [LUT_CHUNK << S]
    float *coarse = (float *)slut;
    const int64_t total = (int64_t)g.nr * g.nc;
    const int64_t quarter = (total + LUT_R - 1) / LUT_R;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int row[LUT_R], col[LUT_R];
    bool na[LUT_R];
    double acc[LUT_R];
#pragma unroll
    for (int c = 0; c < LUT_R; ++c) {
        int64_t i = i0 + c * quarter;
        if (i >= total) i = total - 1;
        row[c] = (int)(i / g.nc); col[c] = (int)(i - (int64_t)row[c] * g.nc);
        na[c] = false; acc[c] = 0.0;
    }
    // keys -> ranks among the predictor's sorted distinct tkeys
    for (int j = 0; j < p; ++j) {
        float r[LUT_R];
        lut_ranks<LUT_R, 256>(j, sorted, key64, sorted_off, coarse, s, g, row, col, na, r);
#pragma unroll
        for (int c = 0; c < LUT_R; ++c) keys[(j * 256 + threadIdx.x) * LUT_R + c] = -r[c];
    }
    const char *kbase = (const char *)(keys + threadIdx.x * LUT_R);
    constexpr unsigned A0_BITS = (unsigned)(127 + 23 - S) << 23;   // float 2^(23-S)
    // byte offset of LUT slot 0 minus 8 * 0x4B000000 (mod 2^32), kept opaque so that slot -> address
    // stays one v_lshl_add_u32
    unsigned lut_base = (unsigned)p * 256u * LUT_R * (unsigned)sizeof(float) - 0x58000000u;
    asm volatile("" : "+s"(lut_base));
    for (int t0 = 0; t0 < n_trees_padded; t0 += LUT_CHUNK) {
        __syncthreads();
        for (int e = threadIdx.x; e < (LUT_CHUNK << S); e += 256) slut[e] = lut[((int64_t)t0 << S) + e];
        __syncthreads();
#pragma unroll 2
        for (int t = 0; t < LUT_CHUNK; ++t) {
            const int *m = meta + (int64_t)(t0 + t) * LUT_META_DW;
            const unsigned long long *mc = (const unsigned long long *)m;
            const unsigned long long a0 = A0_BITS + ((unsigned)t << S);      // float 2^(23-S) + t
            float2v a01, a23, b01, b23;
            {
                const float4 k = *(const float4 *)(kbase + m[6]);
                pred_bits<false>(b01, b23, float2v{k.x, k.y}, float2v{k.z, k.w}, mc[0]);
                asm("v_pk_fma_f32 %0, %2, 2.0, %3 op_sel_hi:[0,0,1]\n\t"
                    "v_pk_fma_f32 %1, %2, 2.0, %4 op_sel_hi:[0,0,1]"
                    : "=&v"(a01), "=&v"(a23) : "s"(a0), "v"(b01), "v"(b23));
            }
#pragma unroll
            for (int q = 1; q < S; ++q) {
                const float4 k = *(const float4 *)(kbase + m[6 + q]);
                if (q & 1) pred_bits<true>(b01, b23, float2v{k.x, k.y}, float2v{k.z, k.w}, mc[q >> 1]);
                else pred_bits<false>(b01, b23, float2v{k.x, k.y}, float2v{k.z, k.w}, mc[q >> 1]);
                asm("v_pk_fma_f32 %0, %0, 2.0, %2 op_sel_hi:[1,0,1]\n\t"
                    "v_pk_fma_f32 %1, %1, 2.0, %3 op_sel_hi:[1,0,1]"
                    : "+v"(a01), "+v"(a23) : "v"(b01), "v"(b23));
            }
            acc[0] = acc[0] + *(const double *)(smem + (__float_as_uint(a01.x) * 8u + lut_base));
            acc[1] = acc[1] + *(const double *)(smem + (__float_as_uint(a01.y) * 8u + lut_base));
            acc[2] = acc[2] + *(const double *)(smem + (__float_as_uint(a23.x) * 8u + lut_base));
            acc[3] = acc[3] + *(const double *)(smem + (__float_as_uint(a23.y) * 8u + lut_base));
        }
    }
#pragma unroll
    for (int c = 0; c < LUT_R; ++c) {
        const int64_t i = i0 + c * quarter;
        if (i0 < quarter && i < total && !na[c])
            emit(out, (int64_t)row[c] * g.ld_out + col[c], init_f + acc[c], weight, accumulate);
    }
}

// Same evaluation with the -rank keys held in 32 VGPRs (v[64:95] = [8 predictors][4 cells]) instead
// of LDS, for models with <= 8 predictors: a level's predictor is selected with the VGPR index
// mode (s_set_gpr_idx_on / _idx add the wave-uniform register offset in M0 to src0 of the packed
// adds), so a level costs no LDS read and no address arithmetic -- 28 VALU per tree and 4 cells
// (20 packed predicate ops, 4 shift-adds, 4 fp64 adds).  Inside the indexed region every VALU
// instruction keeps a constant in src0 except the two that are meant to be indexed.
typedef float float32v __attribute__((ext_vector_type(32)));
constexpr int LUT_REG_P = 8;

#define MHS_LUT_LEVEL(IDX, CP, SEL)                                                   \
    "s_set_gpr_idx_idx %[" IDX "]\n\t"                                               \
    "v_pk_add_f32 %[b01], v[64:65], %[" CP "] " SEL " clamp\n\t"                     \
    "v_pk_add_f32 %[b23], v[66:67], %[" CP "] " SEL " clamp\n\t"                     \
    "v_pk_fma_f32 %[a01], 2.0, %[a01], %[b01] op_sel_hi:[0,1,1]\n\t"                 \
    "v_pk_fma_f32 %[a23], 2.0, %[a23], %[b23] op_sel_hi:[0,1,1]\n\t"
#define MHS_SEL_LO "op_sel_hi:[1,0]"
#define MHS_SEL_HI "op_sel:[0,1] op_sel_hi:[1,1]"

template <int S>
__device__ __forceinline__ void lut_tree_reg(float2v &a01, float2v &a23, const float32v &keys, const int *m,
                                             const unsigned long long a0) {
    const unsigned long long *mc = (const unsigned long long *)m;
    const unsigned long long c01 = mc[0], c23 = mc[1], c45 = mc[2];
    const int i0 = m[6] >> 10, i1 = m[7] >> 10, i2 = m[8] >> 10, i3 = m[9] >> 10, i4 = m[10] >> 10;
    float2v b01, b23;
    if (S == 5) {
        asm("s_set_gpr_idx_on %[i0], 0x1\n\t"
            "v_pk_add_f32 %[b01], v[64:65], %[c01] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_add_f32 %[b23], v[66:67], %[c01] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_fma_f32 %[a01], 2.0, %[a0], %[b01] op_sel_hi:[0,0,1]\n\t"
            "v_pk_fma_f32 %[a23], 2.0, %[a0], %[b23] op_sel_hi:[0,0,1]\n\t"
            MHS_LUT_LEVEL("i1", "c01", MHS_SEL_HI)
            MHS_LUT_LEVEL("i2", "c23", MHS_SEL_LO)
            MHS_LUT_LEVEL("i3", "c23", MHS_SEL_HI)
            MHS_LUT_LEVEL("i4", "c45", MHS_SEL_LO)
            "s_set_gpr_idx_off"
            : [a01] "=&v"(a01), [a23] "=&v"(a23), [b01] "=&v"(b01), [b23] "=&v"(b23)
            : "{v[64:95]}"(keys), [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3), [i4] "s"(i4),
              [c01] "s"(c01), [c23] "s"(c23), [c45] "s"(c45), [a0] "s"(a0));
    } else {
        const int i5 = m[11] >> 10;
        asm("s_set_gpr_idx_on %[i0], 0x1\n\t"
            "v_pk_add_f32 %[b01], v[64:65], %[c01] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_add_f32 %[b23], v[66:67], %[c01] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_fma_f32 %[a01], 2.0, %[a0], %[b01] op_sel_hi:[0,0,1]\n\t"
            "v_pk_fma_f32 %[a23], 2.0, %[a0], %[b23] op_sel_hi:[0,0,1]\n\t"
            MHS_LUT_LEVEL("i1", "c01", MHS_SEL_HI)
            MHS_LUT_LEVEL("i2", "c23", MHS_SEL_LO)
            MHS_LUT_LEVEL("i3", "c23", MHS_SEL_HI)
            MHS_LUT_LEVEL("i4", "c45", MHS_SEL_LO)
            MHS_LUT_LEVEL("i5", "c45", MHS_SEL_HI)
            "s_set_gpr_idx_off"
            : [a01] "=&v"(a01), [a23] "=&v"(a23), [b01] "=&v"(b01), [b23] "=&v"(b23)
            : "{v[64:95]}"(keys), [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3), [i4] "s"(i4), [i5] "s"(i5),
              [c01] "s"(c01), [c23] "s"(c23), [c45] "s"(c45), [a0] "s"(a0));
    }
}

// K64: float64 planes (rank search in double).  A separate instantiation, so that the float-key kernel -- the one the
// resident-float32 bench runs -- keeps its 96 registers without the double search's temporaries.
template <int S, bool K64>
__global__ __launch_bounds__(256, 5) void gbm_lutreg_kernel(const double *__restrict__ lut,
                                                         const int *__restrict__ meta,
                                                         const void *__restrict__ sorted, int key64,
                                                         const int *__restrict__ sorted_off, int n_trees_padded,
                                                         double init_f, int p, StackDev s, PredGeom g,
                                                         double weight, int accumulate,
                                                         double *__restrict__ out) {
    static_assert((LUT_CHUNK << S) * sizeof(double) >= LUT_COARSE * sizeof(float), "coarse table must fit");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *slut = (double *)smem;                                            // [LUT_CHUNK << S]
    float *coarse = (float *)smem;
    const int64_t total = (int64_t)g.nr * g.nc;
    const int64_t quarter = (total + LUT_R - 1) / LUT_R;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int row[LUT_R], col[LUT_R];
    bool na[LUT_R];
    double acc[LUT_R];
#pragma unroll
    for (int c = 0; c < LUT_R; ++c) {
        int64_t i = i0 + c * quarter;
        if (i >= total) i = total - 1;
        row[c] = (int)(i / g.nc); col[c] = (int)(i - (int64_t)row[c] * g.nc);
        na[c] = false; acc[c] = 0.0;
    }
    float32v keys;
#pragma unroll
    for (int j = 0; j < LUT_REG_P; ++j) {
        float r[LUT_R] = {0.f, 0.f, 0.f, 0.f};
        if (j < p) {
            if constexpr (K64) lut_ranks_t<LUT_R, 256, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, r);
            else lut_ranks_t<LUT_R, 256, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, r);
        }
#pragma unroll
        for (int c = 0; c < LUT_R; ++c) keys[j * LUT_R + c] = -r[c];
    }
    constexpr unsigned A0_BITS = (unsigned)(127 + 23 - S) << 23;   // float 2^(23-S)
    unsigned lut_base = 0u - 0x58000000u;                          // see gbm_lut_kernel
    asm volatile("" : "+s"(lut_base));
    for (int t0 = 0; t0 < n_trees_padded; t0 += LUT_CHUNK) {
        __syncthreads();
        for (int e = threadIdx.x; e < (LUT_CHUNK << S); e += 256) slut[e] = lut[((int64_t)t0 << S) + e];
        __syncthreads();
#pragma unroll 2
        for (int t = 0; t < LUT_CHUNK; ++t) {
            float2v a01, a23;
            lut_tree_reg<S>(a01, a23, keys, meta + (int64_t)(t0 + t) * LUT_META_DW, A0_BITS + ((unsigned)t << S));
            acc[0] = acc[0] + *(const double *)(smem + (__float_as_uint(a01.x) * 8u + lut_base));
            acc[1] = acc[1] + *(const double *)(smem + (__float_as_uint(a01.y) * 8u + lut_base));
            acc[2] = acc[2] + *(const double *)(smem + (__float_as_uint(a23.x) * 8u + lut_base));
            acc[3] = acc[3] + *(const double *)(smem + (__float_as_uint(a23.y) * 8u + lut_base));
        }
    }
#pragma unroll
    for (int c = 0; c < LUT_R; ++c) {
        const int64_t i = i0 + c * quarter;
        if (i0 < quarter && i < total && !na[c])
            emit(out, (int64_t)row[c] * g.ld_out + col[c], init_f + acc[c], weight, accumulate);
    }
}

// ROW-TILE form of the register kernel (round 3; S = 5, the reference's interaction depth, V73:493).  A wave owns 256
// consecutive cells of ONE raster row (lane l: columns tile + l + 64 c, c = 0..3), so every split on LAT -- a
// predictor of every model, V73:127-138 -- is WAVE-UNIFORM for all four cells of all 64 lanes, and so is every
// padding level of a tree with fewer than 5 splits (its predicate is 0).  The geometry-dependent tables order each
// tree's levels uniform-first (up to two of them; the leaf LUT is permuted to match, per tree): the K uniform
// predicates are evaluated on the SCALAR unit (compare + select on the row's LAT rank, which sits in an SGPR) and
// folded into the accumulator's start value, and only the NV = 5 - K per-cell levels run as packed vector
// instructions -- 4 NV + 8 VALU per tree and 4 cells instead of 28.  The tree loop branches on NV (wave-uniform, scalar
// branches); the 64-byte record of the next tree is fetched into SGPRs while this one is evaluated.  Same leaf values
// added in the same tree order as gbm_lutreg_kernel and the generic walk: bit-identical results.
// Record (16 dwords): c[5] float rank thresholds of the vector levels | NV | idx[5] VGPR offsets of their predictors |
// uthr[5] int rank thresholds of the uniform levels (0 = never: a padding level).
constexpr int LUT_RT_DW = 16;
constexpr int LUT_RT_MAXK = 2;          // more uniform levels than this run as vector levels (their keys exist too)
typedef unsigned long long u64x8 __attribute__((ext_vector_type(8)));   // one record

// Which kernel for this window: PROBE instantiation of gbm_coherent_kernel (below) = its classification alone, on 64 tiles spread
// over the window and the first 256 trees, summing what the cell loops would cost (in hundredths of what the tree-order
// kernel spends per tree and wave; measured, profiles/r03_gbm_coherent.txt: 12 fixed, 85 + 26 n for a tree with n >= 1
// straddling splits).  Both product kernels are then launched and read the four blocks' sums: the one that loses returns
// at once.  No host synchronisation; smooth rasters run the coherent kernel (4x faster), white noise the tree-order one.
constexpr int GBC_PROBE_BLOCKS = 4, GBC_PROBE_CHUNKS = 4, GBC_PROBE_SLOTS = 1024;
__device__ __forceinline__ bool gbc_coherent_pays(const int *__restrict__ probe) {
    long long cost = 0, cnt = 0;
#pragma unroll
    for (int b = 0; b < GBC_PROBE_BLOCKS; ++b) { cost += probe[2 * b]; cnt += probe[2 * b + 1]; }
    return cost < 83 * cnt;
}
template <bool K64>
__global__ __launch_bounds__(256, 5) void gbm_lutreg_rt_kernel(const double *__restrict__ lut,
                                                            const u64x8 *__restrict__ meta,
                                                            const void *__restrict__ sorted,
                                                            const int *__restrict__ sorted_off, int n_trees_padded,
                                                            double init_f, int p, StackDev s, PredGeom g, int tiles_per_row,
                                                            double weight, int accumulate,
                                                            double *__restrict__ out, const int *__restrict__ probe) {
    constexpr int S = 5;
    static_assert((LUT_CHUNK << S) * sizeof(double) >= LUT_COARSE * sizeof(float), "coarse table must fit");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (probe && gbc_coherent_pays(probe)) return;                            // gbm_coherent_kernel takes this window
    double *slut = (double *)smem;                                            // [LUT_CHUNK << S]
    float *coarse = (float *)smem;
    const int lane = threadIdx.x & 63;
    const int64_t ntiles = (int64_t)g.nr * tiles_per_row;
    int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = tile < ntiles;
    if (!live) tile = ntiles - 1;
    const int trow = (int)(tile / tiles_per_row), tcol = (int)(tile - (int64_t)trow * tiles_per_row) * (64 * LUT_R);
    int row[LUT_R], col[LUT_R];
    bool na[LUT_R], ok[LUT_R];
    double acc[LUT_R];
#pragma unroll
    for (int c = 0; c < LUT_R; ++c) {
        const int cc = tcol + c * 64 + lane;
        ok[c] = live && cc < g.nc;
        row[c] = trow; col[c] = min(cc, g.nc - 1);
        na[c] = false; acc[c] = 0.0;
    }
    float32v keys;
    int latv = 0;
#pragma unroll
    for (int j = 0; j < LUT_REG_P; ++j) {
        float r[LUT_R] = {0.f, 0.f, 0.f, 0.f};
        if (j < p) {
            if constexpr (K64) lut_ranks_t<LUT_R, 256, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, r);
            else lut_ranks_t<LUT_R, 256, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, r);
            if (j == s.C + 1) latv = (int)r[0];
        }
#pragma unroll
        for (int c = 0; c < LUT_R; ++c) keys[j * LUT_R + c] = -r[c];
    }
    const int latrank = __builtin_amdgcn_readfirstlane(latv);      // the wave's row: one rank for all its cells
    unsigned lut_base = 0u - 0x58000000u;                          // see gbm_lut_kernel
    asm volatile("" : "+s"(lut_base));
    for (int t0 = 0; t0 < n_trees_padded; t0 += LUT_CHUNK) {
        __syncthreads();
        for (int e = threadIdx.x; e < (LUT_CHUNK << S); e += 256) slut[e] = lut[((int64_t)t0 << S) + e];
        __syncthreads();
        // the chunk's 64 trees: hand-scheduled loop (tools/gen_gbm_rt_asm.py; the table carries one record past the end)
        const u64x8 *mp = meta + t0;
        asm volatile(
#include "gbm_rt_loop.inc"
            : [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [acc3] "+v"(acc[3])
            : "{v[64:95]}"(keys), [mp] "s"(mp), [lat] "s"(latrank), [lb] "s"(lut_base)
            : "memory", "scc",
              "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
              "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
              "s68", "s69", "s70", "s71", "s72", "s73", "s74",
              "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
              "v56", "v57", "v58", "v59");
    }
#pragma unroll
    for (int c = 0; c < LUT_R; ++c)
        if (ok[c] && !na[c]) emit(out, (int64_t)row[c] * g.ld_out + col[c], init_f + acc[c], weight, accumulate);
}


// COHERENT form of the gbm evaluation (round 3; S = 5).  Neighbouring cells have neighbouring covariates, and a split is a
// threshold: over the 64 x 4 cells of a wave's tile most splits of most trees have the SAME outcome for
// every cell (cfg3's synthetic rasters: 67 % of the trees have no split at all whose threshold lies inside the wave's range
// of that predictor, and the others 1.2 such splits on average).  So, per wave and chunk of 64 trees, with LANE = TREE:
//   * from the wave's [min, max] rank of every predictor (one reduction at the start) each lane classifies the five splits
//     of its tree: all cells left, all cells right, or straddling;
//   * no straddling split: the tree's leaf is the same for all the wave's cells -- the lane adds it to ITS sum of such
//     leaves (the 64 sums are added at the very end and go to every cell);
//   * one straddling split (predictor v, rank threshold c): the cells take one of two leaves, B (bit 0) or A (bit 1); the
//     lane adds B to its sum and queues (v, c, A - B) in a wave-private list; then the wave runs down the list with
//     LANE = CELL: acc += rank_v < c ? A - B : 0 -- one compare, two selects and one fp64 add per cell;
//   * two or more: the tree is queued for the full five-level evaluation (lane = cell, class words broadcast from LDS).
// The same leaves reach every cell's sum; their ORDER differs from the other gbm kernels' (tree order), and A enters as
// B + (A - B): planes agree to ~1e-15 of the prediction, not bitwise.  Incoherent rasters (every tree straddling several
// times) fall to the last case for every tree, which is slower than gbm_lutreg_rt_kernel: launch_gbm_lut samples the
// grid first and takes this kernel only where it pays.
// Block = 16 waves; the chunk's leaf LUT (16 KB) and class words (1.25 KB) are staged in LDS for all of them, double-buffered.
constexpr int GBC_WAVES = 16;
constexpr int BAND_ALIGN = 4 * LUT_R;   // row bands of a window are cut at multiples of this many grid rows (a multiple of LUT_R)
__device__ __forceinline__ double gbc_lds_f64(unsigned a) { return *(__attribute__((address_space(3))) const double *)(uintptr_t)a; }
typedef float float4v __attribute__((ext_vector_type(4)));
template <bool K64, bool PROBE = false>
__global__ __launch_bounds__(1024) void gbm_coherent_kernel(const double *__restrict__ lut, const unsigned *__restrict__ cls,
                                                            const void *__restrict__ sorted, const int *__restrict__ sorted_off,
                                                            int n_trees_padded, double init_f, int p, StackDev s, PredGeom g,
                                                            int tiles_per_row, double weight, int accumulate,
                                                            double *__restrict__ out, int *__restrict__ probe, int tile16,
                                                            const int *__restrict__ axis_rank, int axis_ncol) {
    constexpr int S = 5, CH = LUT_CHUNK, CLS_STRIDE = 384, R = LUT_R;
    if (!PROBE && probe && !gbc_coherent_pays(probe)) return;
    if (PROBE) n_trees_padded = min(n_trees_padded, GBC_PROBE_CHUNKS * CH);
    static_assert(R == 4, "four cells per lane: two packed predicate instructions");
    static_assert((CH << S) * sizeof(double) >= LUT_COARSE * sizeof(float), "coarse table must fit");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *slut = (double *)smem;                                             // 2 x [CH << S], from LDS address 0
    unsigned *scls = (unsigned *)(smem + 2 * (CH << S) * sizeof(double));      // 2 x CLS_STRIDE
    int2 *srange = (int2 *)(scls + 2 * CLS_STRIDE);                            // [waves][8] (min, max) rank
    float *coarse = (float *)smem;
    if ((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem != 0u) __builtin_trap();
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // a wave = 64 columns x R adjacent rows (lane = column, a lane's R cells one below the other): the most compact
    // footprint a wave can have, hence the narrowest rank ranges.  The tiles are anchored to the GRID (rows at multiples of
    // R, columns at multiples of 64 of the grid, clipped by the window), so that a cell meets the same companions -- and
    // its sum the same order -- whether the window is evaluated whole or in row bands cut at multiples of R
    // Round 4 (tile16): 16 columns x 4 R rows per wave (lane = column + 16 x row group, a lane's R cells one below the other)
    // instead of 64 columns x R rows -- the forest's tile (rf_walk_ld_kernel): narrower rank ranges on smooth rasters, more
    // trees without a straddling split.  Anchored to the grid at multiples of 16 rows and 16 columns (BAND_ALIGN = 16 rows).
    const int TW = tile16 ? 16 : 64, TH = tile16 ? 4 * R : R;
    const int roff = (int)(g.r0 % TH), coff = (int)(g.c0 % TW);
    const int64_t ntiles = (int64_t)((g.nr + roff + TH - 1) / TH) * tiles_per_row;
    int64_t tile = (int64_t)blockIdx.x * GBC_WAVES + wave;
    if (PROBE) tile = ntiles >= GBC_PROBE_BLOCKS * GBC_WAVES ? tile * (ntiles / (GBC_PROBE_BLOCKS * GBC_WAVES)) : tile % ntiles;
    const bool live = tile < ntiles;
    if (!live) tile = ntiles - 1;
    const int trow = (int)(tile / tiles_per_row) * TH - roff + (tile16 ? (lane >> 4) * R : 0);
    const int tcol = (int)(tile % tiles_per_row) * TW - coff + (tile16 ? (lane & 15) : lane);
    int row[R], col[R];
    bool na[R], ok[R];
    double acc[R];
#pragma unroll
    for (int c = 0; c < R; ++c) {
        const int rr = trow + c, cc = tcol;
        ok[c] = live && cc >= 0 && cc < g.nc && rr >= 0 && rr < g.nr;
        row[c] = min(max(rr, 0), g.nr - 1); col[c] = min(max(cc, 0), g.nc - 1);
        na[c] = false; acc[c] = 0.0;
    }
    // MINUS the ranks (the packed predicate adds them to c), [predictor][cell] in 32 registers that the several-splits
    // loop addresses through the VGPR index mode, as gbm_lutreg_kernel does
    float32v keys;
#pragma unroll
    for (int j = 0; j < LUT_REG_P; ++j) {
        float r[R] = {0.f, 0.f, 0.f, 0.f};
        if (j < p) {
            if (axis_rank && j >= s.C && !s.all_from_planes) {      // LONG / LAT ranks by table (publish_axis_ranks)
#pragma unroll
                for (int c = 0; c < R; ++c)
                    r[c] = (float)(j == s.C ? axis_rank[g.c0 + col[c]] : axis_rank[(int64_t)axis_ncol + g.r0 + row[c]]);
            } else if constexpr (K64) lut_ranks_t<R, 1024, double>(j, (const double *)sorted, sorted_off, (double *)coarse, s, g, row, col, na, r);
            else lut_ranks_t<R, 1024, float>(j, (const float *)sorted, sorted_off, coarse, s, g, row, col, na, r);
        }
#pragma unroll
        for (int c = 0; c < R; ++c) keys[j * R + c] = -r[c];
        // the wave's rank range of predictor j, NA cells left out (they are walked separately)
        int mn = 0x7fffffff, mx = -1;
#pragma unroll
        for (int c = 0; c < R; ++c)
            if (!na[c]) { mn = min(mn, (int)r[c]); mx = max(mx, (int)r[c]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o)); mx = max(mx, __shfl_xor(mx, o)); }
        if (lane == 0) srange[wave * 8 + j] = make_int2(mn, mx);
    }
    __syncthreads();                                               // the coarse table (it aliases the LUT buffers) is done with
    for (int e = tid; e < (CH << S); e += 1024) slut[e] = lut[e];
    if (tid < CH * S) scls[tid] = cls[tid];
    __syncthreads();
    double usum = 0.0;
    int pcost = 0;
    for (int t0 = 0, buf = 0; t0 < n_trees_padded; t0 += CH, buf ^= 1) {
        const bool more = t0 + CH < n_trees_padded;
        double pre0 = 0.0, pre1 = 0.0;
        unsigned prec = 0u;
        if (more) {
            const double *ln = lut + ((int64_t)(t0 + CH) << S);
            pre0 = ln[tid]; pre1 = ln[tid + 1024];
            if (tid < CH * S) prec = cls[(int64_t)(t0 + CH) * S + tid];
        }
        // ---- lane = tree: classify the five splits against the wave's rank ranges
        const unsigned *cw = scls + buf * CLS_STRIDE + lane * S;
        const double *L = slut + buf * (CH << S) + (lane << S);
        unsigned w[S];
        int2 mm[S];
#pragma unroll
        for (int q = 0; q < S; ++q) w[q] = cw[q];
#pragma unroll
        for (int q = 0; q < S; ++q) mm[q] = srange[wave * 8 + (int)(w[q] & 7u)];
        // (round 5: the straddling levels are first collected as a 5-bit mask and their (c, predictor, bit position) lists are
        // filled from it only in lanes that have any -- one at a time, most significant bit = first level -- instead of 25
        // predicated moves per lane and chunk: the classification is the whole price of a tree without a straddling split)
        unsigned known = 0u, smask = 0u;
#pragma unroll
        for (int q = 0; q < S; ++q) {
            const int c = (int)(w[q] >> 3);
            const bool all1 = mm[q].y < c, all0 = mm[q].x >= c;
            if (all1) known |= 16u >> q;
            if (!(all1 || all0)) smask |= 16u >> q;
        }
        const unsigned nstr = (unsigned)__popc(smask);
        unsigned cl[S], vl[S];     // the straddling levels in level order: c as a float's bits; 4 * predictor | bit position << 8
#pragma unroll
        for (int q = 0; q < S; ++q) { cl[q] = 0u; vl[q] = 0u; }
        {
            unsigned sm = smask;
#pragma unroll
            for (int l = 0; l < S; ++l) {
                if (sm) {                                              // (nested: level l + 1 is looked at only behind level l)
                    const int pos = 31 - __clz((int)sm);               // bit position of the level in the leaf index; level q = 4 - pos
                    unsigned wq = w[S - 1];
#pragma unroll
                    for (int k = 0; k < S - 1; ++k) if (pos == S - 1 - k) wq = w[k];
                    cl[l] = __float_as_uint((float)(int)(wq >> 3));
                    vl[l] = ((wq & 7u) << 2) | ((unsigned)pos << 8);
                    sm &= ~(1u << pos);
                } else break;
            }
        }
        if (PROBE) {
            if (nstr) pcost += 85 + 26 * (int)nstr;
            if (more) {
                double *ld = slut + (buf ^ 1) * (CH << S);
                ld[tid] = pre0; ld[tid + 1024] = pre1;
                if (tid < CH * S) scls[(buf ^ 1) * CLS_STRIDE + tid] = prec;
            }
            __syncthreads();
            continue;
        }
        const double B = L[known];
        const bool single = nstr == 1u, pair = nstr == 2u, multi = nstr >= 3u;
        if (!multi) usum = usum + B;
        // One or two straddling splits: with b1, b2 their predicates the leaf is
        //     B + b1 (A1 - B) + b2 ((A2 - B) + b1 (A12 - A1 - A2 + B)),     A1, A2, A12 = the leaves with bit 1 / bit 2 / both set.
        // The cell loops below form a predicate as the float clamp(c - rank) written into the HIGH word of a register pair
        // whose low word is 0: read as a double that is 0 or 2^-7, so the differences are scaled by 2^7 (2^14) here -- exact.
        double d1 = 0.0, d2 = 0.0, d12 = 0.0;
        if (single || pair) {
            const unsigned p1 = 1u << (vl[0] >> 8), p2 = 1u << (vl[1] >> 8);
            const double A1 = L[known + p1];
            d1 = (A1 - B) * 128.0;
            if (pair) {
                const double A2 = L[known + p2], A12 = L[known + p1 + p2];
                d2 = (A2 - B) * 128.0;
                d12 = (((A12 - A1) - A2) + B) * 16384.0;
            }
        }
        // ---- lane = cell.  What a tree needs sits in the registers of the lane that classified it and is fetched from
        // there with v_readlane (no list in LDS, no wait); the predictor's ranks are picked with the VGPR index mode.
#define MHS_GBC_RL(x) __builtin_amdgcn_readlane((int)(x), t)
#define MHS_GBC_RLD(x) __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), t), __builtin_amdgcn_readlane(__double2loint(x), t))
#define MHS_GBC_BITS(h0, h1, h2, h3, vo, cc)                                                                 \
        asm volatile("s_set_gpr_idx_on %[v], 0x1\n\t"                                                        \
                     "v_add_f32_e64 %[a], v64, %[c] clamp\n\t"                                               \
                     "v_add_f32_e64 %[b], v65, %[c] clamp\n\t"                                               \
                     "v_add_f32_e64 %[d], v66, %[c] clamp\n\t"                                               \
                     "v_add_f32_e64 %[e], v67, %[c] clamp\n\t"                                               \
                     "s_set_gpr_idx_off"                                                                     \
                     : [a] "=&v"(h0), [b] "=&v"(h1), [d] "=&v"(h2), [e] "=&v"(h3)                            \
                     : "{v[64:95]}"(keys), [v] "s"(vo), [c] "s"(cc));
        for (unsigned long long m1 = __builtin_amdgcn_ballot_w64(single); m1; m1 &= m1 - 1ull) {
            const int t = (int)__builtin_ctzll(m1);
            const int cc = MHS_GBC_RL(cl[0]), vo = MHS_GBC_RL(vl[0]) & 0xFF;
            const double e1 = MHS_GBC_RLD(d1);
            int h[R];
            MHS_GBC_BITS(h[0], h[1], h[2], h[3], vo, cc)
#pragma unroll
            for (int c = 0; c < R; ++c) acc[c] = fma(__hiloint2double(h[c], 0), e1, acc[c]);
        }
        for (unsigned long long mp = __builtin_amdgcn_ballot_w64(pair); mp; mp &= mp - 1ull) {
            const int t = (int)__builtin_ctzll(mp);
            const int c1 = MHS_GBC_RL(cl[0]), v1 = MHS_GBC_RL(vl[0]) & 0xFF, c2 = MHS_GBC_RL(cl[1]), v2 = MHS_GBC_RL(vl[1]) & 0xFF;
            const double e1 = MHS_GBC_RLD(d1), e2 = MHS_GBC_RLD(d2), e12 = MHS_GBC_RLD(d12);
            int h[R], k[R];
            MHS_GBC_BITS(h[0], h[1], h[2], h[3], v1, c1)
            MHS_GBC_BITS(k[0], k[1], k[2], k[3], v2, c2)
#pragma unroll
            for (int c = 0; c < R; ++c) {
                const double b1 = __hiloint2double(h[c], 0), b2 = __hiloint2double(k[c], 0);
                acc[c] = fma(b1, e1, acc[c]);
                acc[c] = fma(b2, fma(b1, e12, e2), acc[c]);
            }
        }
#undef MHS_GBC_BITS
#undef MHS_GBC_RLD
#undef MHS_GBC_RL
        // Three or more: their straddling levels only (the others' bits are in `known`), each a packed predicate and a packed
        // multiply-add that drops the bit into the slot, which is kept as a float 2^23 + slot (its bit pattern is the LUT's
        // dword index plus a constant)
        const unsigned abase = (unsigned)buf * (unsigned)((CH << S) * sizeof(double)) - 0x58000000u;
        for (unsigned long long m2 = __builtin_amdgcn_ballot_w64(multi); m2; m2 &= m2 - 1ull) {
            const int t = (int)__builtin_ctzll(m2);
            const int n = __builtin_amdgcn_readlane((int)nstr, t);
            const float a0 = __uint_as_float(0x4B000000u + ((unsigned)t << S) + (unsigned)__builtin_amdgcn_readlane((int)known, t));
            float2v s01 = {a0, a0}, s23 = {a0, a0};
#pragma unroll
            for (int l = 0; l < S; ++l) {
                if (l < n) {
                    const unsigned long long cc = (unsigned)__builtin_amdgcn_readlane((int)cl[l], t);
                    const unsigned vw = (unsigned)__builtin_amdgcn_readlane((int)vl[l], t);
                    const int vo = (int)(vw & 0xFFu);
                    const unsigned long long wb = (127u + (vw >> 8)) << 23;
                    float2v b01, b23;
                    asm volatile("s_set_gpr_idx_on %[vo], 0x1\n\t"
                                 "v_pk_add_f32 %[b01], v[64:65], %[cc] op_sel_hi:[1,0] clamp\n\t"
                                 "v_pk_add_f32 %[b23], v[66:67], %[cc] op_sel_hi:[1,0] clamp\n\t"
                                 "s_set_gpr_idx_off\n\t"
                                 "v_pk_fma_f32 %[s01], %[b01], %[wb], %[s01] op_sel_hi:[1,0,1]\n\t"
                                 "v_pk_fma_f32 %[s23], %[b23], %[wb], %[s23] op_sel_hi:[1,0,1]"
                                 : [s01] "+v"(s01), [s23] "+v"(s23), [b01] "=&v"(b01), [b23] "=&v"(b23)
                                 : "{v[64:95]}"(keys), [vo] "s"(vo), [cc] "s"(cc), [wb] "s"(wb));
                }
            }
            acc[0] = acc[0] + gbc_lds_f64(__float_as_uint(s01.x) * 8u + abase);
            acc[1] = acc[1] + gbc_lds_f64(__float_as_uint(s01.y) * 8u + abase);
            acc[2] = acc[2] + gbc_lds_f64(__float_as_uint(s23.x) * 8u + abase);
            acc[3] = acc[3] + gbc_lds_f64(__float_as_uint(s23.y) * 8u + abase);
        }
        // ---- the next chunk into the other buffer
        if (more) {
            double *ld = slut + (buf ^ 1) * (CH << S);
            ld[tid] = pre0; ld[tid + 1024] = pre1;
            if (tid < CH * S) scls[(buf ^ 1) * CLS_STRIDE + tid] = prec;
        }
        __syncthreads();
    }
    if (PROBE) {
        int *tot = (int *)srange;                                   // the ranges are done with
        __syncthreads();
        if (tid == 0) tot[0] = 0;
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pcost += __shfl_xor(pcost, o);
        if (lane == 0) atomicAdd(tot, pcost);
        __syncthreads();
        if (tid == 0) { probe[2 * blockIdx.x] = tot[0]; probe[2 * blockIdx.x + 1] = GBC_WAVES * n_trees_padded; }
        return;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) usum = usum + __shfl_xor(usum, o);
#pragma unroll
    for (int c = 0; c < R; ++c)
        if (ok[c] && !na[c]) emit(out, (int64_t)row[c] * g.ld_out + col[c], init_f + (acc[c] + usum), weight, accumulate);
}
static size_t gbc_lds_bytes() {
    return 2 * ((size_t)LUT_CHUNK << 5) * sizeof(double) + 2 * 384 * sizeof(unsigned) + GBC_WAVES * 8 * sizeof(int2);
}

// predict.gbm(model, newdata, n.trees = step, 2 step, ...) at a table of points: what machisplin.gbm.step evaluates on
// every fold's hold-out rows after every gbm.more (V73:1843, 1919) -- here in one walk over the trees, the running sum
// written out every `step` trees (same additions in the same order as a model cut at that tree count)
__global__ __launch_bounds__(256) void gbm_staged_points_kernel(const Node *__restrict__ nodes, const int *__restrict__ tree_off,
                                                                int n_trees, double init_f, const double *__restrict__ X,
                                                                int64_t n, int step, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int t = 0; t < n_trees; ++t) {
        const int tb = tree_off[t];
        Node nd = nodes[tb];
        while (nd.var >= 0) {
            const double xv = X[(int64_t)nd.var * n + i];
            const unsigned nxt = isnan(xv) ? nd.missing : (xv < nd.val ? nd.left : nd.right);
            nd = nodes[tb + nxt];
        }
        acc = acc + nd.val;
        if ((t + 1) % step == 0) out[(int64_t)((t + 1) / step - 1) * n + i] = init_f + acc;
    }
}

// res.FINAL at the stations (V73:477-482 ... 608-611, 620): ((resp - pred_1) w_1 + (resp - pred_2) w_2 + ...) / wt.tot,
// member after member as the reference accumulates it
struct ResidualArgs { double w[8]; };
__global__ __launch_bounds__(256) void residual_points_kernel(const double *__restrict__ pred, const double *__restrict__ resp,
                                                              int64_t n, int n_models, ResidualArgs a, double wt_total,
                                                              double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double y = resp[i];
    double res = (y - pred[i]) * a.w[0];
    for (int k = 1; k < n_models; ++k) res = res + (y - pred[(int64_t)k * n + i]) * a.w[k];
    out[i] = res / wt_total;
}

__global__ __launch_bounds__(256) void scale_add_kernel(const double *__restrict__ a, double divisor,
                                                        const double *__restrict__ b,
                                                        double *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = a[i] / divisor;
    if (b) v = v + b[i];
    out[i] = v;
}

__global__ __launch_bounds__(256) void scale_window_kernel(double *__restrict__ out, int nr, int nc,
                                                           int64_t ld, double divisor) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)nr * nc) return;
    const int row = (int)(i / nc), col = (int)(i - (int64_t)row * nc);
    out[(int64_t)row * ld + col] = out[(int64_t)row * ld + col] / divisor;
}

// ------------------------------------------------------------------ host side --

int finish_trees(mhs_model *m, const std::vector<Node> &nodes, const std::vector<int> &off) {
    const int nt = m->n_trees;
    const size_t xs_bytes = (size_t)m->p * TREE_R * 256 * sizeof(double);
    int biggest = 0;
    for (int t = 0; t < nt; ++t) biggest = std::max(biggest, off[t + 1] - off[t]);
    int cap = (m->kind == K_GBM) ? std::max(GBM_CHUNK_NODES, biggest) : biggest;
    m->lds_ok = xs_bytes + (size_t)cap * sizeof(Node) <= LDS_LIMIT;
    std::vector<TreeChunk> chunks;
    if (m->lds_ok) {
        int t = 0;
        while (t < nt) {
            TreeChunk c{t, 0, off[t], 0};
            while (t < nt && off[t + 1] - c.node_begin <= cap) { ++t; }
            c.n_trees = t - c.first_tree;
            c.node_count = off[t] - c.node_begin;
            chunks.push_back(c);
        }
        m->max_chunk_nodes = cap;
    } else {
        chunks.push_back(TreeChunk{0, nt, 0, off[nt]});
        m->max_chunk_nodes = 0;
    }
    m->n_chunks = (int)chunks.size();
    m->n_nodes = (int64_t)nodes.size();
    if (int rc = to_device(nodes.data(), nodes.size(), &m->nodes)) return rc;
    if (int rc = to_device(off.data(), off.size(), &m->tree_off)) return rc;
    return to_device(chunks.data(), chunks.size(), &m->chunks);
}

int check_common(int p, mhs_model **out) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(out != nullptr, "out is NULL");
    MHS_REQUIRE(p >= 2 && p <= 64, "p (covariates + LONG + LAT) out of range");
    return MHS_OK;
}

template <int P>
static void launch_nnet(const mhs_model *m, const StackDev &s, const PredGeom &g, double w, int acc,
                        double *out, hipStream_t st, unsigned blocks) {
    hipLaunchKernelGGL((nnet_kernel<P>), dim3(blocks), dim3(256), 0, st, m->dpar, m->n0, m->s0, m->s1, s, g, w, acc, out);
}
template <int P>
static void launch_svr(const mhs_model *m, const StackDev &s, const PredGeom &g, double w, int acc,
                       double *out, hipStream_t st, int64_t total) {
    constexpr int R = 3;      // cells per lane (2: 83.5, 3: 81.0, 4: 82.4 ms on 8000^2 cells x 3000 SVs)
    // grid mode with long rows: a wave = 192 cells of one row, the LAT term of the exponent once per wave (svr_rt_kernel)
    const int tpr = (g.nc + 64 * R - 1) / (64 * R);
    if (!s.all_from_planes && P == s.C + 2 && P >= 3 && !getenv("MHS_SVR_NO_ROWTILE") && (double)g.nc >= 0.93 * (double)tpr * (64 * R)) {
        const int64_t ntiles = (int64_t)g.nr * tpr;
        hipLaunchKernelGGL((svr_rt_kernel<P, R>), dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st,
                           m->dpar, m->n0, m->n1, m->n2, m->dpar + (size_t)m->n0 * m->n1, ctx().exp_tab, m->s1, m->s0, m->s4,
                           m->s2, m->s3, s, g, tpr, w, acc, out);
        return;
    }
    const int64_t half = (total + R - 1) / R;
    hipLaunchKernelGGL((svr_kernel<P, R>), dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st,
                       m->dpar, m->n0, m->n1, m->n2, m->dpar + (size_t)m->n0 * m->n1, ctx().exp_tab, m->s1, m->s0, m->s4,
                       m->s2, m->s3,
                       s, g, w, acc, out);
}

#define MHS_DISPATCH_P(P_, CALL)                                                     \
    switch (P_) {                                                                    \
        case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;      \
        case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break;      \
        case 8: CALL(8); break; case 9: CALL(9); break; case 10: CALL(10); break;    \
        case 11: CALL(11); break; case 12: CALL(12); break;                          \
        default: set_error("predict: p = %d exceeds the %d predictors this build supports for nnet/ksvm", P_, PMAX); \
                 return MHS_ERR_INVALID;                                             \
    }

template <bool GBM>
__global__ __launch_bounds__(256) void tree_finalize_kernel(const double *__restrict__ part, int shares, int64_t total,
                                                            double init_f, int n_trees, PredGeom g, double weight,
                                                            int accumulate, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    double acc = 0.0;
    for (int y = 0; y < shares; ++y) acc = acc + part[(int64_t)y * total + i];
    const int row = (int)(i / g.nc), col = (int)(i - (int64_t)row * g.nc);
    emit(out, (int64_t)row * g.ld_out + col, GBM ? init_f + acc : acc / (double)n_trees, weight, accumulate);
}

constexpr int64_t TREE_SPLIT_MAX_CELLS = 16384;   // below this the tree loop is shared out over blockIdx.y
constexpr int TREE_SPLIT_SHARES = 64;

template <bool GBM, bool NA_ONLY>
static int launch_trees(const mhs_model *m, const StackDev &s, const PredGeom &g, double w, int acc,
                        double *out, hipStream_t st, int64_t total) {
    const int64_t half = (total + TREE_R - 1) / TREE_R;
    const unsigned blocks = (unsigned)((half + 255) / 256);
    const size_t xs_bytes = (size_t)m->p * TREE_R * 256 * sizeof(double);
    if (!NA_ONLY && m->lds_ok && total <= TREE_SPLIT_MAX_CELLS && m->n_chunks >= 4) {
        const int shares = std::min(m->n_chunks, TREE_SPLIT_SHARES);
        mhs_model *mm = const_cast<mhs_model *>(m);
        if (!mm->split_scratch) MHS_HIP(hipMalloc((void **)&mm->split_scratch, sizeof(double) * TREE_SPLIT_MAX_CELLS * TREE_SPLIT_SHARES));
        const size_t bytes = xs_bytes + (size_t)m->max_chunk_nodes * sizeof(Node);
        auto kern = tree_kernel<GBM, true, TREE_R, false, true>;
        MHS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(kern, dim3(blocks, (unsigned)shares), dim3(256), bytes, st, m->nodes, m->tree_off, m->chunks,
                           m->n_chunks, m->n_trees, m->init_f, m->p, s, g, w, acc, mm->split_scratch, (const unsigned *)nullptr);
        hipLaunchKernelGGL(tree_finalize_kernel<GBM>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           mm->split_scratch, shares, total, m->init_f, m->n_trees, g, w, acc, out);
        return MHS_OK;
    }
    if (m->lds_ok) {
        const size_t bytes = xs_bytes + (size_t)m->max_chunk_nodes * sizeof(Node);
        auto kern = tree_kernel<GBM, true, TREE_R, NA_ONLY>;
        MHS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), bytes, st, m->nodes, m->tree_off, m->chunks,
                           m->n_chunks, m->n_trees, m->init_f, m->p, s, g, w, acc, out, (const unsigned *)nullptr);
    } else {
        auto kern = tree_kernel<GBM, false, TREE_R, NA_ONLY>;
        MHS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs_bytes));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), xs_bytes, st, m->nodes, m->tree_off, m->chunks,
                           m->n_chunks, m->n_trees, m->init_f, m->p, s, g, w, acc, out, (const unsigned *)nullptr);
    }
    return MHS_OK;
}

// key-space thresholds of every split for this grid, the sorted distinct thresholds of each
// predictor and every split's rank among them (see gbm_lut_kernel); cached per geometry and key type
template <typename KT>
static int build_lut_meta_t(mhs_model *m, const mhs_grid &grid, int C) {
    const int S = m->lut_S;
    std::vector<KT> tkey((size_t)m->n_trees * S, (KT)0);
    std::vector<std::vector<KT>> sorted((size_t)m->p);
    for (int t = 0; t < m->n_trees; ++t) {
        for (int q = 0; q < S; ++q) {
            const int v = m->lut_var[(size_t)t * S + q];
            if (v < 0) continue;
            const KT tk = split_tkey<KT, false>(v, C, m->lut_thr[(size_t)t * S + q], grid);
            tkey[(size_t)t * S + q] = tk;
            sorted[(size_t)v].push_back(tk);
        }
    }
    std::vector<int> off;
    std::vector<KT> flat;
    sort_unique(sorted, off, flat);
    for (int v = 0; v < m->p; ++v)
        if (sorted[(size_t)v].size() >= ((size_t)1 << 24)) { set_error("gbm: too many distinct split values"); return MHS_ERR_INVALID; }
    if (int rc = publish_axis_ranks(m, sorted, C, grid)) return rc;
    std::vector<int> meta((size_t)m->n_trees_padded * LUT_META_DW, 0);
    const float never = -33554432.f;   // c - rank <= 0 for every rank: the padded predicates read 0
    for (int t = 0; t < m->n_trees_padded; ++t) {
        int *mt = &meta[(size_t)t * LUT_META_DW];
        for (int q = 0; q < 6; ++q) { memcpy(&mt[q], &never, 4); mt[6 + q] = 0; }
        if (t >= m->n_trees) continue;
        for (int q = 0; q < S; ++q) {
            const int v = m->lut_var[(size_t)t * S + q];
            if (v < 0) continue;
            const std::vector<KT> &sv = sorted[(size_t)v];
            const KT tk = tkey[(size_t)t * S + q];
            const float c = (float)((std::lower_bound(sv.begin(), sv.end(), tk) - sv.begin()) + 1);
            memcpy(&mt[q], &c, 4);
            mt[6 + q] = v * 256 * LUT_R * (int)sizeof(float);
        }
    }
    // row-tile tables (gbm_lutreg_rt_kernel, S = 5): per tree, up to LUT_RT_MAXK of the levels that are uniform along
    // a raster row -- splits on LAT (predictor C + 1) and padding levels -- first, the other levels after them, the leaf
    // LUT permuted to that order; one all-padding record past the last tree (the kernel fetches a tree ahead)
    if (m->p <= LUT_REG_P && S == 5 && !m->lut_host.empty()) {
        const int lat = C + 1 < m->p ? C + 1 : -2;
        std::vector<int> rmeta(((size_t)m->n_trees_padded + 1) * LUT_RT_DW, 0);
        std::vector<double> rlut(m->lut_host.size(), 0.0);
        for (int t = 0; t <= m->n_trees_padded; ++t) {
            int *mt = &rmeta[(size_t)t * LUT_RT_DW];
            for (int q = 0; q < 5; ++q) { memcpy(&mt[q], &never, 4); mt[6 + q] = 0; mt[11 + q] = 0; }
            mt[5] = S - LUT_RT_MAXK;            // padded trees: two uniform padding levels, three vector ones that read 0
            if (t >= m->n_trees) continue;
            int perm[6], K = 0, n = 0;
            bool first[6] = {false, false, false, false, false, false};
            for (int q = 0; q < S && K < LUT_RT_MAXK; ++q) {
                const int v = m->lut_var[(size_t)t * S + q];
                if (v < 0 || v == lat) { perm[n++] = q; first[q] = true; ++K; }
            }
            for (int q = 0; q < S; ++q) if (!first[q]) perm[n++] = q;
            for (int i = 0; i < S; ++i) {
                const int q = perm[i], v = m->lut_var[(size_t)t * S + q];
                if (v < 0) continue;            // padding: uthr 0 / c = never
                const std::vector<KT> &sv = sorted[(size_t)v];
                const int c = (int)(std::lower_bound(sv.begin(), sv.end(), tkey[(size_t)t * S + q]) - sv.begin()) + 1;
                if (i < K) mt[11 + i] = c;
                else { const float cf = (float)c; memcpy(&mt[i - K], &cf, 4); mt[6 + i - K] = v * LUT_R; }
            }
            mt[5] = S - K;
            for (int b = 0; b < (1 << S); ++b) {        // b: slot in the new order (level i <-> bit S-1-i)
                int bo = 0;
                for (int i = 0; i < S; ++i) bo |= ((b >> (S - 1 - i)) & 1) << (S - 1 - perm[i]);
                rlut[((size_t)t << S) + b] = m->lut_host[((size_t)t << S) + bo];
            }
        }
        if (int rc = publish(m, rlut, &m->lut_rt)) return rc;
        if (int rc = publish(m, rmeta, &m->lut_rt_meta)) return rc;
        // class words (gbm_coherent_kernel): level q of tree t = rank threshold << 3 | predictor, 0 for a padding level
        std::vector<unsigned> cw((size_t)m->n_trees_padded * 5, 0u);
        for (int t = 0; t < m->n_trees; ++t)
            for (int q = 0; q < S; ++q) {
                const int v = m->lut_var[(size_t)t * S + q];
                if (v < 0) continue;
                const std::vector<KT> &sv = sorted[(size_t)v];
                const unsigned c = (unsigned)(std::lower_bound(sv.begin(), sv.end(), tkey[(size_t)t * S + q]) - sv.begin()) + 1u;
                cw[(size_t)t * 5 + q] = (c << 3) | (unsigned)v;
            }
        if (int rc = publish(m, cw, &m->lut_cls)) return rc;
    }
    if (int rc = publish(m, flat, (KT **)&m->lut_sorted)) return rc;
    if (int rc = publish(m, off, &m->lut_sorted_off)) return rc;
    return publish(m, meta, &m->lut_meta);
}

static int build_lut_meta(mhs_model *m, const mhs_grid &grid, int C, int key64, TreeTables *tt) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (!(m->lut_meta && same_meta(m, grid, C, key64))) {
        if (int rc = key64 ? build_lut_meta_t<double>(m, grid, C) : build_lut_meta_t<float>(m, grid, C)) return rc;
        m->meta_grid = grid; m->meta_C = C; m->meta_key64 = key64;
    }
    *tt = TreeTables{m->lut_sorted, m->lut_sorted_off, m->lut_meta, nullptr, nullptr, m->lut_rt, m->lut_rt_meta, m->lut_cls, m->axis_rank, m->axis_ncol};
    return MHS_OK;
}

// the MissingNode walk of a window's NA cells behind a predicate-LUT kernel: compacted list, strided pass on overflow
static int launch_gbm_na(const mhs_model *m, const StackDev &s, const PredGeom &g, double w, int acc, double *out, hipStream_t st,
                         int64_t total) {
    if (total >= (1LL << 32)) return launch_trees<true, true>(m, s, g, w, acc, out, st, total);
    mhs_model *mm = const_cast<mhs_model *>(m);
    const unsigned slot = mm->na_next.fetch_add(1) % 4;
    const size_t want = (size_t)std::max<int64_t>(4096, total / 8);      // up to 12.5 % NA cells; beyond: the strided pass
    unsigned *list = nullptr;
    unsigned cap = 0;
    {
        // One model predicted from several streams or host threads (round-4 advisor finding): launch k + 4 reuses launch k's
        // buffer, so it first waits -- on the device -- for the event recorded behind launch k's last reader, whichever stream
        // that ran on; list / cap are read under the lock that guards their reallocation.
        std::lock_guard<std::mutex> lk(mm->mu);
        if (!mm->na_done[slot]) MHS_HIP(hipEventCreateWithFlags(&mm->na_done[slot], hipEventDisableTiming));
        else if (mm->na_stream[slot] != st) MHS_HIP(hipStreamWaitEvent(st, mm->na_done[slot], 0));
        if (mm->na_cap[slot] < want) {
            if (mm->na_list[slot]) {
                MHS_HIP(hipEventSynchronize(mm->na_done[slot]));
                MHS_HIP(hipStreamSynchronize(st));
                (void)hipFree(mm->na_list[slot]); mm->na_list[slot] = nullptr; mm->na_cap[slot] = 0;
            }
            MHS_HIP(hipMalloc((void **)&mm->na_list[slot], sizeof(unsigned) * (want + 2)));
            mm->na_cap[slot] = want;
        }
        list = mm->na_list[slot];
        cap = (unsigned)mm->na_cap[slot];
        mm->na_stream[slot] = st;
    }
    MHS_HIP(hipMemsetAsync(list, 0, 2 * sizeof(unsigned), st));
    hipLaunchKernelGGL(na_collect_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s, g, cap, list);
    const unsigned lblocks = (unsigned)std::min<int64_t>(2048, (total / 8 + 255) / 256 + 1);
    const size_t xs_bytes = (size_t)m->p * 256 * sizeof(double);
    if (m->lds_ok) {
        const size_t bytes = xs_bytes + (size_t)m->max_chunk_nodes * sizeof(Node);
        MHS_HIP(hipFuncSetAttribute((const void *)gbm_na_list_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(gbm_na_list_kernel<true>, dim3(lblocks), dim3(256), bytes, st, m->nodes, m->tree_off, m->chunks, m->n_chunks, m->init_f,
                           m->p, s, g, w, acc, out, list, cap);
    } else {
        MHS_HIP(hipFuncSetAttribute((const void *)gbm_na_list_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs_bytes));
        hipLaunchKernelGGL(gbm_na_list_kernel<false>, dim3(lblocks), dim3(256), xs_bytes, st, m->nodes, m->tree_off, m->chunks, m->n_chunks,
                           m->init_f, m->p, s, g, w, acc, out, list, cap);
    }
    // the strided NA-only pass, gated on the overflow word (its blocks return at once otherwise)
    const int64_t half = (total + TREE_R - 1) / TREE_R;
    const unsigned blocks = (unsigned)((half + 255) / 256);
    const size_t xs4 = (size_t)m->p * TREE_R * 256 * sizeof(double);
    if (m->lds_ok) {
        const size_t bytes = xs4 + (size_t)m->max_chunk_nodes * sizeof(Node);
        auto kern = tree_kernel<true, true, TREE_R, true>;
        MHS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), bytes, st, m->nodes, m->tree_off, m->chunks, m->n_chunks, m->n_trees, m->init_f, m->p,
                           s, g, w, acc, out, (const unsigned *)list);
    } else {
        auto kern = tree_kernel<true, false, TREE_R, true>;
        MHS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs4));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), xs4, st, m->nodes, m->tree_off, m->chunks, m->n_chunks, m->n_trees, m->init_f, m->p,
                           s, g, w, acc, out, (const unsigned *)list);
    }
    MHS_HIP(hipGetLastError());
    {
        std::lock_guard<std::mutex> lk(mm->mu);
        MHS_HIP(hipEventRecord(mm->na_done[slot], st));
    }
    return MHS_OK;
}

static int launch_gbm_lut(const mhs_model *m, const StackDev &s, const PredGeom &g, const mhs_grid &grid,
                          double w, int acc, double *out, hipStream_t st, int64_t total) {
    const int key64 = s.dtype == MHS_F64;
    TreeTables tt;
    if (int rc = build_lut_meta(const_cast<mhs_model *>(m), grid, s.C, key64, &tt)) return rc;
    const int64_t quarter = (total + LUT_R - 1) / LUT_R;
    const unsigned blocks = (unsigned)((quarter + 255) / 256);
    const size_t lut_bytes = ((size_t)LUT_CHUNK << m->lut_S) * sizeof(double);
    const bool in_regs = m->p <= LUT_REG_P;
    // row-tile form: a wave = 256 consecutive cells of one row (LAT splits and padding levels on the scalar unit); taken
    // when the rows are long enough that the ragged last tile of a row wastes little
    const int tpr = (g.nc + 64 * LUT_R - 1) / (64 * LUT_R);
    const int ctw = 16, cth = 4 * LUT_R;                                   // the coherent kernel's 16 x 16-cell wave tiles
    const int ctpr = (int)((g.nc + g.c0 % ctw + ctw - 1) / ctw);
    const bool rt_ok = in_regs && m->lut_S == 5 && tt.lut_rt && (double)g.nc >= 0.93 * (double)tpr * (64 * LUT_R);
    const bool coh_ok = in_regs && m->lut_S == 5 && tt.lut_cls && g.nr >= 2 * LUT_R && (double)g.nc >= 0.8 * (double)ctpr * ctw &&
                        !getenv("MHS_GBM_NO_COHERENT");
    // Grids: the coherent kernel where neighbouring cells share their trees' outcomes, the tree-order row-tile kernel where
    // they do not (white-noise rasters) -- decided on the device by a probe of 64 tiles (large windows; small ones are not
    // worth two more launches and take the coherent kernel; MHS_GBM_FORCE_COHERENT: no probe)
    int *probe = nullptr;
    if (coh_ok && rt_ok && total >= (1 << 20) && !getenv("MHS_GBM_FORCE_COHERENT")) {
        mhs_model *mm = const_cast<mhs_model *>(m);
        {
            std::lock_guard<std::mutex> lk(mm->mu);
            if (!mm->gbm_probe) MHS_HIP(hipMalloc(&mm->gbm_probe, sizeof(int) * 2 * GBC_PROBE_BLOCKS * GBC_PROBE_SLOTS));
        }
        probe = mm->gbm_probe + (size_t)(mm->gbm_probe_next.fetch_add(1) % GBC_PROBE_SLOTS) * 2 * GBC_PROBE_BLOCKS;
    }
    if (coh_ok) {
        const int64_t ntiles = (int64_t)((g.nr + g.r0 % cth + cth - 1) / cth) * ctpr;
        const unsigned cblocks = (unsigned)((ntiles + GBC_WAVES - 1) / GBC_WAVES);
        if (probe) {
            auto pk = key64 ? gbm_coherent_kernel<true, true> : gbm_coherent_kernel<false, true>;
            MHS_HIP(hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gbc_lds_bytes()));
            hipLaunchKernelGGL(pk, dim3(GBC_PROBE_BLOCKS), dim3(1024), gbc_lds_bytes(), st, m->lut, tt.lut_cls, tt.sorted, tt.sorted_off,
                               m->n_trees_padded, m->init_f, m->p, s, g, ctpr, w, acc, out, probe, 1, tt.axis_rank, tt.axis_ncol);
        }
        auto ck = key64 ? gbm_coherent_kernel<true> : gbm_coherent_kernel<false>;
        MHS_HIP(hipFuncSetAttribute((const void *)ck, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gbc_lds_bytes()));
        hipLaunchKernelGGL(ck, dim3(cblocks), dim3(1024), gbc_lds_bytes(), st, m->lut, tt.lut_cls, tt.sorted, tt.sorted_off,
                           m->n_trees_padded, m->init_f, m->p, s, g, ctpr, w, acc, out, probe, 1, tt.axis_rank, tt.axis_ncol);
        if (!probe) return launch_gbm_na(m, s, g, w, acc, out, st, total);
    }
    if (rt_ok) {
        const int64_t ntiles = (int64_t)g.nr * tpr;
        const unsigned rblocks = (unsigned)((ntiles + 3) / 4);
        auto rk = key64 ? gbm_lutreg_rt_kernel<true> : gbm_lutreg_rt_kernel<false>;
        MHS_HIP(hipFuncSetAttribute((const void *)rk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lut_bytes));
        hipLaunchKernelGGL(rk, dim3(rblocks), dim3(256), lut_bytes, st, tt.lut_rt, (const u64x8 *)tt.lut_rt_meta, tt.sorted, tt.sorted_off,
                           m->n_trees_padded, m->init_f, m->p, s, g, tpr, w, acc, out, (const int *)probe);
        return launch_gbm_na(m, s, g, w, acc, out, st, total);
    }
    const size_t bytes = in_regs ? lut_bytes : (size_t)m->p * 256 * LUT_R * sizeof(float) + lut_bytes;
    auto kern = in_regs ? (m->lut_S == 5 ? (key64 ? gbm_lutreg_kernel<5, true> : gbm_lutreg_kernel<5, false>)
                                         : (key64 ? gbm_lutreg_kernel<6, true> : gbm_lutreg_kernel<6, false>))
                        : (m->lut_S == 5 ? gbm_lut_kernel<5> : gbm_lut_kernel<6>);
    MHS_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), bytes, st, m->lut, tt.lut_meta, tt.sorted, key64, tt.sorted_off,
                       m->n_trees_padded, m->init_f, m->p, s, g, w, acc, out);
    // cells with an NA covariate: walked through their MissingNode children
    return launch_gbm_na(m, s, g, w, acc, out, st, total);
}


static int launch_model(const mhs_model *m, const StackDev &s, const PredGeom &g, double weight,
                        int accumulate, double *out, hipStream_t st, const mhs_grid *grid = nullptr) {
    const int64_t total = (int64_t)g.nr * g.nc;
    if (total == 0) return MHS_OK;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    switch (m->kind) {
        case K_LM:
            hipLaunchKernelGGL(lm_kernel, dim3(blocks), dim3(256), 0, st, m->dpar, m->p, s, g, weight, accumulate, out);
            break;
        case K_NNET: {
#define CALL_NNET(P) launch_nnet<P>(m, s, g, weight, accumulate, out, st, blocks)
            MHS_DISPATCH_P(m->p, CALL_NNET)
#undef CALL_NNET
            break;
        }
        case K_EARTH:
            hipLaunchKernelGGL(earth_kernel, dim3(blocks), dim3(256), 0, st, m->dpar, m->dpar + m->n0,
                               m->ipar, m->ipar + m->n0 + 1, m->ipar + m->n0 + 1 + m->n1, m->n0, m->p, s, g,
                               weight, accumulate, out);
            break;
        case K_SVR: {
#define CALL_SVR(P) launch_svr<P>(m, s, g, weight, accumulate, out, st, total)
            MHS_DISPATCH_P(m->p, CALL_SVR)
#undef CALL_SVR
            break;
        }
        case K_GBM:
            // fast path: grid mode (rank search in float for float32 / int16 planes, in double for float64 planes)
            if (grid && m->lut_S > 0 && !s.all_from_planes && !getenv("MHS_TREES_GENERIC") && (size_t)m->p * 256 * LUT_R * 4 + ((size_t)LUT_CHUNK << m->lut_S) * 8 <= LDS_LIMIT) {
                if (int rc = launch_gbm_lut(m, s, g, *grid, weight, accumulate, out, st, total)) return rc;
            } else if (int rc = launch_trees<true, false>(m, s, g, weight, accumulate, out, st, total)) return rc;
            break;
        case K_RF: {
            bool launched = false;
            if (int rc = launch_forest(m, s, g, grid, weight, accumulate, out, st, total, &launched)) return rc;
            if (!launched) if (int rc = launch_trees<false, false>(m, s, g, weight, accumulate, out, st, total)) return rc;
            break;
        }
        default:
            set_error("predict: unknown model kind %d", m->kind);
            return MHS_ERR_INVALID;
    }
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

template <int P>
static void launch_small(const SmallArgs &a, const StackDev &s, const PredGeom &g, int acc, double *out, hipStream_t st, unsigned blocks) {
    hipLaunchKernelGGL((small_members_kernel<P>), dim3(blocks), dim3(256), 0, st, a, s, g, acc, out);
}

// out (+)= sum_k w_k pred_k for the members in order (V73:471 ... 606); a run of two or three of (lm, nnet, earth) --
// consecutive in the reference's model order -- goes through one fused pass.  MHS_NO_FUSE=1 launches them one by one.
static int launch_members(const mhs_model *const *models, const double *weights, int n_models, const StackDev &s,
                          const PredGeom &g, int accumulate_first, double *out, hipStream_t st, const mhs_grid *grid) {
    constexpr bool fuse = true;
    const int64_t total = (int64_t)g.nr * g.nc;
    int k = 0;
    bool masked_done = false;
    // which member leaves the compute units free: randomForest if present -- LDS-bound, one block per CU, it loses the
    // least (+13 %; gbm +26 %, ksvm +50 % measured) -- else ksvm, else gbm
    int mask_kind = -1;
    for (int want : {K_RF, K_SVR, K_GBM})
        for (int q = 0; q < n_models && mask_kind < 0; ++q)
            if (models[q]->kind == want) mask_kind = want;
    // (round 4, measured and not kept: a SMALL reservation held for the forest's and ksvm's 190 ms instead of a large one for the
    // forest's 70 -- the 32-column fit wants the whole chip for its MFMA updates: confined to 8 / 16 / 32 units it ends after
    // 486 / 336 / 259 ms of a cfg3 step, so the calibration of sharded.py now reserves nothing)
    while (k < n_models) {
        const int acc = (k > 0 || accumulate_first) ? 1 : 0;
        // the longest run lm? nnet? earth? starting at k (each at most once, in that order)
        int e = k;
        const mhs_model *lm = nullptr, *nn = nullptr, *ea = nullptr;
        if (fuse && total > 0 && models[k]->p <= PMAX) {
            if (e < n_models && models[e]->kind == K_LM) lm = models[e++];
            if (e < n_models && models[e]->kind == K_NNET && models[e]->p == models[k]->p) nn = models[e++];
            if (e < n_models && models[e]->kind == K_EARTH && models[e]->p == models[k]->p) ea = models[e++];
        }
        if (e - k >= 2) {
            SmallArgs a{};
            if (lm) { a.lm_coef = lm->dpar; a.w_lm = weights[k]; }
            if (nn) { a.nn_w = nn->dpar; a.nn_H = nn->n0; a.nn_scale = nn->s0; a.nn_shift = nn->s1; a.w_nn = weights[k + (lm ? 1 : 0)]; }
            if (ea) {
                a.ea_coef = ea->dpar; a.ea_cut = ea->dpar + ea->n0; a.ea_tstart = ea->ipar; a.ea_fvar = ea->ipar + ea->n0 + 1;
                a.ea_fdir = ea->ipar + ea->n0 + 1 + ea->n1; a.ea_nterms = ea->n0; a.w_ea = weights[e - 1];
            }
            const unsigned blocks = (unsigned)((total + 255) / 256);
            const int P_ = models[k]->p;
#define CALL_SMALL(P) launch_small<P>(a, s, g, acc, out, st, blocks)
            MHS_DISPATCH_P(P_, CALL_SMALL)
#undef CALL_SMALL
            MHS_HIP(hipGetLastError());
            k = e;
            continue;
        }
        // mhs_fit_reserve_cus: the chosen long member runs on the masked stream, fenced by two events so that it keeps
        // its place in the caller's stream order
        const int kind = models[k]->kind;
        if (!masked_done && ctx().reserved_cus > 0 && grid && total >= (1 << 22) && kind == mask_kind) {
            Context &c = ctx();
            std::lock_guard<std::mutex> lk(mask_mutex());
            if (c.masked_stream) {
                masked_done = true;
                MHS_HIP(hipEventRecord(c.mask_ev0, st));
                MHS_HIP(hipStreamWaitEvent(c.masked_stream, c.mask_ev0, 0));
                if (int rc = launch_model(models[k], s, g, weights[k], acc, out, c.masked_stream, grid)) return rc;
                MHS_HIP(hipEventRecord(c.mask_ev1, c.masked_stream));
                MHS_HIP(hipStreamWaitEvent(st, c.mask_ev1, 0));
                ++k;
                continue;
            }
        }
        if (int rc = launch_model(models[k], s, g, weights[k], acc, out, st, grid)) return rc;
        ++k;
    }
    return MHS_OK;
}

static int make_stack(const mhs_model *m, const mhs_grid *g, const mhs_stack *c, StackDev *s) {
    MHS_REQUIRE(c && c->data, "covariate stack is NULL");
    MHS_REQUIRE(c->n_layers == m->p - 2, "stack has the wrong number of layers for this model (p = layers + 2)");
    MHS_REQUIRE(c->dtype == MHS_F64 || c->dtype == MHS_F32 || c->dtype == MHS_I16, "bad stack dtype");
    MHS_REQUIRE(c->ld >= g->ncol && c->plane_stride >= c->ld * g->nrow, "stack strides smaller than the grid");
    s->data = c->data; s->C = c->n_layers; s->dtype = c->dtype; s->plane_stride = c->plane_stride;
    s->ld = c->ld; s->nodata = c->nodata; s->has_nodata = !std::isnan(c->nodata); s->all_from_planes = 0;
    return MHS_OK;
}

static int make_geom(const mhs_grid *g, int64_t r0, int64_t r1, int64_t c0, int64_t c1, int64_t ld, PredGeom *pg) {
    MHS_REQUIRE(g && g->nrow > 0 && g->ncol > 0 && g->xres > 0 && g->yres > 0, "bad grid geometry");
    MHS_REQUIRE(0 <= r0 && r0 <= r1 && r1 <= g->nrow && 0 <= c0 && c0 <= c1 && c1 <= g->ncol, "window outside the grid");
    MHS_REQUIRE(ld >= c1 - c0, "ld smaller than the window width");
    MHS_REQUIRE((r1 - r0) * (c1 - c0) < (1LL << 40) && r1 - r0 < (1LL << 31) && c1 - c0 < (1LL << 31), "window too large");
    pg->xmin = g->xmin; pg->ymax = g->ymax; pg->xres = g->xres; pg->yres = g->yres;
    pg->r0 = r0; pg->c0 = c0; pg->nr = (int)(r1 - r0); pg->nc = (int)(c1 - c0); pg->ld_out = ld;
    return MHS_OK;
}

// Members k of `models` on rows [b0, b1) of the grid from a device buffer that holds rows [buf_r0, buf_r1) of every covariate
// plane (plane k at buf + k * (buf_r1 - buf_r0) * ld elements): the device copy is described with the PARENT grid's affine,
// so cell centres -- and with them every member's value -- are those of the one-piece evaluation (multi.hip: one row band
// per device, evaluated in sub-bands while the rest of the band still travels).  `accumulate`: add to out_dev instead of
// starting it; `scale`: divide by wt_total afterwards (the last members of a cell).  out_dev = row b0's first cell.
int members_rows_dev(const mhs_model *const *models, const double *weights, int n_models, int accumulate, int scale, double wt_total,
                     const mhs_grid *g, const void *buf, int64_t buf_r0, int64_t buf_r1, int n_layers, int dtype, int64_t ld,
                     double nodata, int64_t b0, int64_t b1, double *out_dev, int64_t ld_out, hipStream_t st) {
    MHS_REQUIRE(models && weights && n_models >= 0 && g && buf && out_dev, "bad ensemble arguments");
    MHS_REQUIRE(wt_total != 0.0 && !std::isnan(wt_total), "wt_total must be non-zero");
    MHS_REQUIRE(dtype == MHS_F64 || dtype == MHS_F32 || dtype == MHS_I16, "bad stack dtype");
    MHS_REQUIRE(buf_r0 <= b0 && b0 <= b1 && b1 <= buf_r1, "rows outside the buffer");
    for (int k = 0; k < n_models; ++k)
        MHS_REQUIRE(models[k] && n_layers == models[k]->p - 2, "stack has the wrong number of layers for a model");
    if (b1 == b0) return MHS_OK;
    const size_t esz = dtype == MHS_F64 ? 8 : dtype == MHS_F32 ? 4 : 2;
    PredGeom pg;
    if (int rc = make_geom(g, b0, b1, 0, g->ncol, ld_out, &pg)) return rc;
    StackDev sd;
    sd.data = (const char *)buf - (size_t)buf_r0 * ld * esz; sd.C = n_layers; sd.dtype = dtype;
    sd.plane_stride = (buf_r1 - buf_r0) * ld; sd.ld = ld; sd.nodata = nodata;
    sd.has_nodata = !std::isnan(nodata); sd.all_from_planes = 0;
    if (n_models > 0) if (int rc = launch_members(models, weights, n_models, sd, pg, accumulate, out_dev, st, g)) return rc;
    if (scale) {
        const int64_t total = (b1 - b0) * g->ncol;
        hipLaunchKernelGGL(scale_window_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out_dev, (int)(b1 - b0),
                           (int)g->ncol, ld_out, wt_total);
        MHS_HIP(hipGetLastError());
    }
    return MHS_OK;
}

// pred.elev on rows [b0, b1) from a buffer that holds exactly those rows
int ensemble_band_dev(const mhs_model *const *models, const double *weights, int n_models, double wt_total, const mhs_grid *g,
                      const void *band_data, int n_layers, int dtype, int64_t ld, double nodata, int64_t b0, int64_t b1,
                      double *out_dev, int64_t ld_out, hipStream_t st) {
    MHS_REQUIRE(n_models >= 1, "bad ensemble arguments");
    return members_rows_dev(models, weights, n_models, 0, 1, wt_total, g, band_data, b0, b1, n_layers, dtype, ld, nodata, b0, b1, out_dev,
                            ld_out, st);
}

// The handle's twin on another device slot, built on first use from the remembered loader call.
int model_on_slot(const mhs_model *m, int slot, const mhs_model **out) {
    MHS_REQUIRE(m && out && slot >= 0 && slot < MAX_SLOTS, "bad arguments");
    const int dev = ctx_slot(slot).device;
    if (m->slot == slot && m->device == dev) { *out = m; return MHS_OK; }
    mhs_model *w = const_cast<mhs_model *>(m);
    std::lock_guard<std::mutex> lk(w->mu);
    if (w->replica[slot] && w->replica[slot]->device != dev) {      // the slots were re-initialised on other devices
        mhs_model_free(w->replica[slot]);
        w->replica[slot] = nullptr;
    }
    if (!w->replica[slot]) {
        MHS_REQUIRE((bool)w->reload, "model handle cannot be replicated");
        SlotBind bind(slot);
        mhs_model *r = nullptr;
        if (int rc = w->reload(&r)) return rc;
        w->replica[slot] = r;
    }
    *out = w->replica[slot];
    return MHS_OK;
}

}  // namespace mhs

using namespace mhs;

extern "C" {

int mhs_model_free(mhs_model *m) {
    if (!m) return MHS_OK;
    for (mhs_model *&r : m->replica) if (r) { mhs_model_free(r); r = nullptr; }
    if (m->dpar) (void)hipFree(m->dpar);
    if (m->ipar) (void)hipFree(m->ipar);
    if (m->nodes) (void)hipFree(m->nodes);
    if (m->tree_off) (void)hipFree(m->tree_off);
    if (m->chunks) (void)hipFree(m->chunks);
    if (m->split_scratch) (void)hipFree(m->split_scratch);
    for (unsigned *q : m->na_list) if (q) (void)hipFree(q);
    for (hipEvent_t e : m->na_done) if (e) (void)hipEventDestroy(e);
    if (m->lut) (void)hipFree(m->lut);
    if (m->lut_meta) (void)hipFree(m->lut_meta);
    if (m->lut_rt) (void)hipFree(m->lut_rt);
    if (m->lut_rt_meta) (void)hipFree(m->lut_rt_meta);
    if (m->lut_cls) (void)hipFree(m->lut_cls);
    if (m->gbm_probe) (void)hipFree(m->gbm_probe);
    if (m->lut_sorted) (void)hipFree(m->lut_sorted);
    if (m->lut_sorted_off) (void)hipFree(m->lut_sorted_off);
    if (m->axis_rank) (void)hipFree(m->axis_rank);
    if (m->rf_nodes) (void)hipFree(m->rf_nodes);
    if (m->rf_lval) (void)hipFree(m->rf_lval);
    if (m->rf_depth) (void)hipFree(m->rf_depth);
    if (m->rf_dmin) (void)hipFree(m->rf_dmin);
    if (m->rf_coff) (void)hipFree(m->rf_coff);
    if (m->rf_csub) (void)hipFree(m->rf_csub);
    if (m->rf_clval) (void)hipFree(m->rf_clval);
    for (void *q : m->retired) (void)hipFree(q);
    delete m;
    return MHS_OK;
}

int mhs_lm_load(const double *coef, int p, mhs_model **out) {
    if (int rc = check_common(p, out)) return rc;
    MHS_REQUIRE(coef != nullptr, "coef is NULL");
    mhs_model *m = new mhs_model();
    m->kind = K_LM; m->p = p;
    if (int rc = to_device(coef, (size_t)p + 1, &m->dpar)) { mhs_model_free(m); return rc; }
    m->slot = current_slot(); m->device = ctx().device;
    m->reload = [v = std::vector<double>(coef, coef + p + 1), p](mhs_model **o) { return mhs_lm_load(v.data(), p, o); };
    *out = m;
    return MHS_OK;
}

int mhs_nnet_load(const double *wts, int p, int size, double y_scale, double y_shift, mhs_model **out) {
    if (int rc = check_common(p, out)) return rc;
    MHS_REQUIRE(wts != nullptr && size >= 1 && size <= 4096, "bad nnet arguments");
    MHS_REQUIRE(p <= PMAX, "p exceeds the predictors supported for nnet");
    mhs_model *m = new mhs_model();
    m->kind = K_NNET; m->p = p; m->n0 = size; m->s0 = y_scale; m->s1 = y_shift;
    if (int rc = to_device(wts, (size_t)(p + 1) * size + size + 1, &m->dpar)) { mhs_model_free(m); return rc; }
    m->slot = current_slot(); m->device = ctx().device;
    m->reload = [v = std::vector<double>(wts, wts + (size_t)(p + 1) * size + size + 1), p, size, y_scale, y_shift](mhs_model **o) {
        return mhs_nnet_load(v.data(), p, size, y_scale, y_shift, o);
    };
    *out = m;
    return MHS_OK;
}

int mhs_earth_load(const double *coef, const int32_t *dirs, const double *cuts, int nterms, int p,
                   mhs_model **out) {
    if (int rc = check_common(p, out)) return rc;
    MHS_REQUIRE(coef && dirs && cuts && nterms >= 1, "bad earth arguments");
    std::vector<int> tstart(1, 0), fvar, fdir;
    std::vector<double> fcut;
    for (int k = 0; k < nterms; ++k) {
        for (int v = 0; v < p; ++v) {
            const int d = dirs[(size_t)k * p + v];
            MHS_REQUIRE(d == 0 || d == 1 || d == -1 || d == 2, "earth dirs must be 0, 1, -1 or 2");
            if (d != 0) { fvar.push_back(v); fdir.push_back(d); fcut.push_back(cuts[(size_t)k * p + v]); }
        }
        tstart.push_back((int)fvar.size());
    }
    mhs_model *m = new mhs_model();
    m->kind = K_EARTH; m->p = p; m->n0 = nterms; m->n1 = (int)fvar.size();
    std::vector<double> dp(coef, coef + nterms);
    dp.insert(dp.end(), fcut.begin(), fcut.end());
    std::vector<int> ip(tstart);
    ip.insert(ip.end(), fvar.begin(), fvar.end());
    ip.insert(ip.end(), fdir.begin(), fdir.end());
    int rc = to_device(dp.data(), dp.size(), &m->dpar);
    if (!rc) rc = to_device(ip.data(), ip.size(), &m->ipar);
    if (rc) { mhs_model_free(m); return rc; }
    m->slot = current_slot(); m->device = ctx().device;
    m->reload = [c = std::vector<double>(coef, coef + nterms), d = std::vector<int32_t>(dirs, dirs + (size_t)nterms * p),
                 q = std::vector<double>(cuts, cuts + (size_t)nterms * p), nterms, p](mhs_model **o) {
        return mhs_earth_load(c.data(), d.data(), q.data(), nterms, p, o);
    };
    *out = m;
    return MHS_OK;
}

int mhs_svr_load(const double *alpha, const double *sv, int64_t nsv, int p, double b, double sigma,
                 const double *x_center, const double *x_scale, double y_center, double y_scale,
                 mhs_model **out) {
    if (int rc = check_common(p, out)) return rc;
    MHS_REQUIRE(alpha && sv && x_center && x_scale && nsv >= 1 && nsv < (1LL << 30), "bad ksvm arguments");
    MHS_REQUIRE(p <= PMAX, "p exceeds the predictors supported for ksvm");
    MHS_REQUIRE(sigma > 0, "sigma must be positive");
    for (int j = 0; j < p; ++j) MHS_REQUIRE(x_scale[j] != 0.0, "x_scale has a zero entry");
    const int stride = ((p + 1) + 3) & ~3;  // doubles per support vector, 32-byte multiple
    // support vectors with alpha > 0 first, then alpha < 0 (alpha = 0 contributes nothing), |alpha| / amax folded
    // into the exponent
    double amax = 0.0;
    std::vector<int64_t> order;
    for (int64_t v = 0; v < nsv; ++v) {
        MHS_REQUIRE(std::isfinite(alpha[v]), "non-finite alpha");
        amax = std::max(amax, fabs(alpha[v]));
        if (alpha[v] > 0) order.push_back(v);
    }
    const int npos = (int)order.size();
    for (int64_t v = 0; v < nsv; ++v) if (alpha[v] < 0) order.push_back(v);
    const int64_t nkeep = (int64_t)order.size();
    std::vector<double> h((size_t)nkeep * stride + 2 * p, 0.0);
    for (int64_t e = 0; e < nkeep; ++e) {
        const int64_t v = order[(size_t)e];
        double ss = 0.0;
        for (int j = 0; j < p; ++j) {
            const double x = sv[(size_t)v * p + j];
            h[(size_t)e * stride + j] = -2.0 * sigma * x / EXP_RANGE;
            ss += x * x;
        }
        h[(size_t)e * stride + p] = (sigma * ss - log(fabs(alpha[v]) / amax)) / EXP_RANGE;
    }
    for (int j = 0; j < p; ++j) { h[(size_t)nkeep * stride + j] = x_center[j]; h[(size_t)nkeep * stride + p + j] = x_scale[j]; }
    mhs_model *m = new mhs_model();
    m->kind = K_SVR; m->p = p; m->n0 = (int)nkeep; m->n1 = stride; m->n2 = npos; m->s4 = amax;
    m->s0 = b; m->s1 = sigma; m->s2 = y_center; m->s3 = y_scale;
    if (int rc = to_device(h.data(), h.size(), &m->dpar)) { mhs_model_free(m); return rc; }
    m->slot = current_slot(); m->device = ctx().device;
    m->reload = [a = std::vector<double>(alpha, alpha + nsv), v = std::vector<double>(sv, sv + (size_t)nsv * p), nsv, p, b, sigma,
                 xc = std::vector<double>(x_center, x_center + p), xs = std::vector<double>(x_scale, x_scale + p), y_center,
                 y_scale](mhs_model **o) {
        return mhs_svr_load(a.data(), v.data(), nsv, p, b, sigma, xc.data(), xs.data(), y_center, y_scale, o);
    };
    *out = m;
    return MHS_OK;
}

int mhs_gbm_load(double init_f, int64_t n_trees, const int64_t *tree_offsets, const int32_t *split_var,
                 const double *split_val, const int32_t *left, const int32_t *right,
                 const int32_t *missing, int p, mhs_model **out) {
    if (int rc = check_common(p, out)) return rc;
    MHS_REQUIRE(tree_offsets && split_var && split_val && left && right && missing, "NULL gbm array");
    MHS_REQUIRE(n_trees >= 0 && n_trees < (1LL << 30) && tree_offsets[0] == 0, "bad tree offsets");
    const int64_t nn = tree_offsets[n_trees];
    MHS_REQUIRE(nn < (1LL << 31), "too many nodes");
    std::vector<Node> nodes((size_t)nn);
    std::vector<int> off((size_t)n_trees + 1);
    for (int64_t t = 0; t <= n_trees; ++t) off[t] = (int)tree_offsets[t];
    for (int64_t t = 0; t < n_trees; ++t) {
        const int64_t o = tree_offsets[t], cnt = tree_offsets[t + 1] - o;
        MHS_REQUIRE(cnt >= 1 && cnt <= 65535, "a gbm tree must have 1..65535 nodes");
        for (int64_t k = 0; k < cnt; ++k) {
            Node &nd = nodes[(size_t)(o + k)];
            nd.val = split_val[o + k];
            nd.var = (short)split_var[o + k];
            if (split_var[o + k] >= 0) {
                MHS_REQUIRE(split_var[o + k] < p, "gbm SplitVar out of range");
                MHS_REQUIRE(left[o + k] >= 0 && left[o + k] < cnt && right[o + k] >= 0 && right[o + k] < cnt &&
                            missing[o + k] >= 0 && missing[o + k] < cnt, "gbm child index out of range");
                nd.left = (unsigned short)left[o + k]; nd.right = (unsigned short)right[o + k];
                nd.missing = (unsigned short)missing[o + k];
            } else { nd.var = -1; nd.left = nd.right = nd.missing = 0; }
        }
    }
    mhs_model *m = new mhs_model();
    m->kind = K_GBM; m->p = p; m->n_trees = (int)n_trees; m->init_f = init_f;
    if (int rc = finish_trees(m, nodes, off)) { mhs_model_free(m); return rc; }
    // predicate-LUT form (gbm_lut_kernel) when every tree has at most 6 splits
    int max_splits = 0;
    for (int64_t t = 0; t < n_trees; ++t) {
        int ns = 0;
        for (int k = off[t]; k < off[t + 1]; ++k) ns += nodes[(size_t)k].var >= 0;
        max_splits = std::max(max_splits, ns);
    }
    if (n_trees > 0 && max_splits <= 6) {
        const int S = max_splits <= 5 ? 5 : 6;
        m->lut_S = S;
        m->n_trees_padded = (int)((n_trees + LUT_CHUNK - 1) / LUT_CHUNK * LUT_CHUNK);
        m->lut_var.assign((size_t)n_trees * S, -1);
        m->lut_thr.assign((size_t)n_trees * S, 0.0);
        std::vector<double> lut(((size_t)m->n_trees_padded) << S, 0.0);
        std::vector<int> qmap;
        for (int64_t t = 0; t < n_trees; ++t) {
            const int o = off[t], cnt = off[t + 1] - off[t];
            qmap.assign((size_t)cnt, -1);
            int q = 0;
            for (int k = 0; k < cnt; ++k)
                if (nodes[(size_t)(o + k)].var >= 0) {
                    qmap[k] = q;
                    m->lut_var[(size_t)t * S + q] = nodes[(size_t)(o + k)].var;
                    m->lut_thr[(size_t)t * S + q] = nodes[(size_t)(o + k)].val;
                    ++q;
                }
            for (int b = 0; b < (1 << S); ++b) {
                int k = 0, guard = 0;
                while (nodes[(size_t)(o + k)].var >= 0 && guard++ <= cnt) {
                    const int bit = (b >> (S - 1 - qmap[k])) & 1;  // predicate 0 is the most significant bit
                    k = bit ? nodes[(size_t)(o + k)].left : nodes[(size_t)(o + k)].right;
                }
                lut[((size_t)t << S) + b] = nodes[(size_t)(o + k)].val;
            }
        }
        if (int rc = to_device(lut.data(), lut.size(), &m->lut)) { mhs_model_free(m); return rc; }
        m->lut_host = std::move(lut);
    }
    m->slot = current_slot(); m->device = ctx().device;
    m->reload = [init_f, n_trees, to = std::vector<int64_t>(tree_offsets, tree_offsets + n_trees + 1),
                 sv = std::vector<int32_t>(split_var, split_var + nn), sl = std::vector<double>(split_val, split_val + nn),
                 l = std::vector<int32_t>(left, left + nn), r = std::vector<int32_t>(right, right + nn),
                 ms = std::vector<int32_t>(missing, missing + nn), p](mhs_model **o) {
        return mhs_gbm_load(init_f, n_trees, to.data(), sv.data(), sl.data(), l.data(), r.data(), ms.data(), p, o);
    };
    *out = m;
    return MHS_OK;
}

int mhs_predict_dev(const mhs_model *m, const mhs_grid *g, const mhs_stack *covars, int64_t r0,
                    int64_t r1, int64_t c0, int64_t c1, double weight, int accumulate, double *out_dev,
                    int64_t ld, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(m && out_dev, "NULL argument");
    PredGeom pg;
    StackDev s;
    if (int rc = make_geom(g, r0, r1, c0, c1, ld, &pg)) return rc;
    if (int rc = make_stack(m, g, covars, &s)) return rc;
    return launch_model(m, s, pg, weight, accumulate, out_dev, pick_stream(stream), g);
}

int mhs_members_predict_dev(const mhs_model *const *models, const double *weights, int n_models, const mhs_grid *g,
                            const mhs_stack *covars, int64_t r0, int64_t r1, int64_t c0, int64_t c1, int accumulate,
                            double *out_dev, int64_t ld, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(models && weights && n_models >= 1 && out_dev, "bad arguments");
    PredGeom pg;
    StackDev s;
    if (int rc = make_geom(g, r0, r1, c0, c1, ld, &pg)) return rc;
    for (int k = 0; k < n_models; ++k) {
        MHS_REQUIRE(models[k] != nullptr, "NULL model");
        if (int rc = make_stack(models[k], g, covars, &s)) return rc;     // every member must fit the stack
    }
    return launch_members(models, weights, n_models, s, pg, accumulate, out_dev, pick_stream(stream), g);
}

int mhs_ensemble_predict_dev(const mhs_model *const *models, const double *weights, int n_models,
                             double wt_total, const mhs_grid *g, const mhs_stack *covars, int64_t r0,
                             int64_t r1, int64_t c0, int64_t c1, double *out_dev, int64_t ld, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(models && weights && n_models >= 1 && out_dev, "bad ensemble arguments");
    MHS_REQUIRE(wt_total != 0.0 && !std::isnan(wt_total), "wt_total must be non-zero");
    if (int rc = mhs_members_predict_dev(models, weights, n_models, g, covars, r0, r1, c0, c1, 0, out_dev, ld, stream)) return rc;
    const int64_t total = (r1 - r0) * (c1 - c0);
    if (total > 0) {
        hipLaunchKernelGGL(scale_window_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           pick_stream(stream), out_dev, (int)(r1 - r0), (int)(c1 - c0), ld, wt_total);
        MHS_HIP(hipGetLastError());
    }
    return MHS_OK;
}

// The host-pointer form -- what the R shim calls with terra's in-memory rasters (V73:468-606 reads, predicts and writes
// block by block) -- as a three-stream pipeline over ROW BANDS: while band k is predicted, band k + 1's covariate rows
// travel host -> device and band k - 1's result device -> host.  Buffers come from the library's persistent arena (two
// covariate bands + two result bands; no hipMalloc / hipFree per call).  The host side issues, in this order, "kernels of
// band k, upload of band k + 1, download of band k - 1": copies from / to pageable memory block the CALLING THREAD until
// they are staged, so the kernels must already be in the queue when the thread goes into them.  Cells are independent
// and a band is described with the parent grid's affine, so the plane equals the one-piece evaluation bit for bit.
// MHS_HOST_BANDS = n forces n equal bands (1 = the serial round-2 behaviour, minus the allocations).
// Host-pointer ensemble, large windows (round 3).  Cutting the WHOLE member sequence into row bands hides the copies but
// pays the partly filled last round of every member kernel once per band (measured: 3 bands +20 ms on a 497 ms pass).  Only
// two things have to be banded: the FIRST member launch, so that it can start on the rows that have arrived while the rest
// of the covariates still travels (bands of 4, 16, 40, 40 % of the rows: the exposed upload is the 4 %), and the LAST one, so
// that finished rows travel back under the rows still being computed (48, 30, 14, 6, 2 %: the exposed download is the 2 %; round 5: both band plans grow no faster than the copies outrun the kernels, see below).
// Everything between them runs once over the whole window.  Per cell the members are still accumulated in the caller's
// order, so the plane equals the one-piece evaluation bit for bit.  The window lives in the persistent arena.
// Error returns of the host-pointer pipelines: copies between the caller's pageable buffers and the arena may still be in flight on
// the three pipe streams when a later call fails; the caller is free to release its buffers once the entry point has returned,
// so every exit that is not the normal one drains the streams first (round-3 advisor finding).
struct PipeDrain {
    Context &c;
    bool done = false;
    explicit PipeDrain(Context &ctx_) : c(ctx_) {}
    ~PipeDrain() {
        if (done) return;
        (void)hipStreamSynchronize(c.pipe_h2d); (void)hipStreamSynchronize(c.pipe_comp); (void)hipStreamSynchronize(c.pipe_d2h);
    }
};

static int host_window_pipeline(const mhs_model *const *models, const double *weights, int n_models, int first_end, int last_start,
                                double wt_total, const mhs_grid *g, const mhs_stack *covars, int64_t r0, int64_t r1, int64_t c0,
                                int64_t c1, double *out_host) {
    const int64_t nr = r1 - r0, nc = c1 - c0;
    const size_t esz = covars->dtype == MHS_F64 ? 8 : covars->dtype == MHS_F32 ? 4 : 2;
    const size_t plane_bytes = (size_t)nr * covars->ld * esz;
    const size_t in_bytes = (plane_bytes * (size_t)covars->n_layers + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lk(pipe_mutex());
    if (int rc = host_pipe(in_bytes + (size_t)nr * nc * sizeof(double))) return rc;
    Context &c = ctx();
    PipeDrain drain(c);
    char *in = c.pipe_arena;
    double *outp = (double *)(c.pipe_arena + in_bytes);
    // Upload bands: band b + 1 must have arrived when band b's kernels end, and the copies (0.43 ms per % of cfg3's three
    // float64 planes) are only ~2.2 x faster than the members that run on them (gbm + forest: 0.93 ms per %): with 4, 16, 40,
    // 40 % the device waited 3 ms for the 16 % and 2 ms for the first 40 %.  Float64 planes therefore go in five bands that
    // grow by at most that factor (3, 6, 13, 28, 50 %); float32 / int16 planes (half / a quarter of the bytes) keep four.
    const bool wide = covars->dtype == MHS_F64;
    const int NU = wide ? 5 : 4;
    const int pct_up5[5] = {3, 6, 13, 28, 50}, pct_up4[4] = {4, 16, 40, 40};
    // ... and the last member (ksvm: 1.1 ms per %) runs ~2.3 x slower than its rows travel down (0.48 ms per %): the same rule mirrored
    constexpr int ND = 5;
    const int pct_down[ND] = {48, 30, 14, 6, 2};
    int64_t up[6], down[ND + 1];
    up[0] = down[0] = r0;
    // cut at multiples of BAND_ALIGN grid rows (gbm_coherent_kernel's tiles are anchored to the grid: a band sees whole tiles)
    for (int b = 0, a = 0; b < NU; ++b) {
        a += wide ? pct_up5[b] : pct_up4[b];
        up[b + 1] = b == NU - 1 ? r1 : std::min(r1, std::max(up[b], (r0 + nr * a / 100 + BAND_ALIGN / 2) / BAND_ALIGN * BAND_ALIGN));
    }
    for (int b = 0, d = 0; b < ND; ++b) {
        d += pct_down[b];
        down[b + 1] = b == ND - 1 ? r1 : std::min(r1, std::max(down[b], (r0 + nr * d / 100 + BAND_ALIGN / 2) / BAND_ALIGN * BAND_ALIGN));
    }
    StackDev sd;
    sd.data = in - (size_t)r0 * covars->ld * esz; sd.C = covars->n_layers; sd.dtype = covars->dtype;
    sd.plane_stride = nr * covars->ld; sd.ld = covars->ld; sd.nodata = covars->nodata;
    sd.has_nodata = !std::isnan(covars->nodata); sd.all_from_planes = 0;
    const bool timing = getenv("MHS_TIMING") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now_ms();
    auto upload = [&](int b) -> int {           // rows [up[b], up[b + 1]) of every layer; blocks the calling thread (pageable source)
        for (int k = 0; k < covars->n_layers; ++k)
            MHS_HIP(hipMemcpyAsync(in + plane_bytes * k + (size_t)(up[b] - r0) * covars->ld * esz,
                                   (const char *)covars->data + ((size_t)k * covars->plane_stride + (size_t)up[b] * covars->ld) * esz,
                                   (size_t)(up[b + 1] - up[b]) * covars->ld * esz, hipMemcpyHostToDevice, c.pipe_h2d));
        MHS_HIP(hipEventRecord(c.pipe_in[b], c.pipe_h2d));
        return MHS_OK;
    };
    auto members = [&](int k0, int k1, int64_t b0, int64_t b1, int acc) -> int {
        PredGeom pg;
        if (int rc = make_geom(g, b0, b1, c0, c1, nc, &pg)) return rc;
        return launch_members(models + k0, weights + k0, k1 - k0, sd, pg, acc, outp + (size_t)(b0 - r0) * nc, c.pipe_comp, g);
    };
    if (int rc = upload(0)) return rc;
    for (int b = 0; b < NU; ++b) {
        MHS_HIP(hipStreamWaitEvent(c.pipe_comp, c.pipe_in[b], 0));
        if (up[b + 1] > up[b]) if (int rc = members(0, first_end, up[b], up[b + 1], 0)) return rc;
        if (b + 1 < NU) if (int rc = upload(b + 1)) return rc;
    }
    const double t_up = now_ms();
    if (last_start > first_end) if (int rc = members(first_end, last_start, r0, r1, 1)) return rc;
    for (int b = 0; b < ND; ++b) {
        if (down[b + 1] > down[b]) {
            if (int rc = members(last_start, n_models, down[b], down[b + 1], 1)) return rc;
            hipLaunchKernelGGL(scale_window_kernel, dim3((unsigned)(((down[b + 1] - down[b]) * nc + 255) / 256)), dim3(256), 0, c.pipe_comp,
                               outp + (size_t)(down[b] - r0) * nc, (int)(down[b + 1] - down[b]), (int)nc, nc, wt_total);
            MHS_HIP(hipGetLastError());
        }
        MHS_HIP(hipEventRecord(c.pipe_done[b], c.pipe_comp));
    }
    for (int b = 0; b < ND; ++b) {
        if (down[b + 1] == down[b]) continue;
        MHS_HIP(hipStreamWaitEvent(c.pipe_d2h, c.pipe_done[b], 0));
        MHS_HIP(hipMemcpyAsync(out_host + (size_t)(down[b] - r0) * nc, outp + (size_t)(down[b] - r0) * nc,
                               sizeof(double) * (size_t)((down[b + 1] - down[b]) * nc), hipMemcpyDeviceToHost, c.pipe_d2h));
    }
    MHS_HIP(hipStreamSynchronize(c.pipe_d2h));
    MHS_HIP(hipStreamSynchronize(c.pipe_comp));
    drain.done = true;
    if (timing) fprintf(stderr, "[mhs_ensemble_predict] window pipeline: uploads issued by %.1f ms, all done at %.1f ms\n", t_up - t_start, now_ms() - t_start);
    return MHS_OK;
}

int mhs_ensemble_predict(const mhs_model *const *models, const double *weights, int n_models,
                         double wt_total, const mhs_grid *g, const mhs_stack *covars, int64_t r0,
                         int64_t r1, int64_t c0, int64_t c1, double *out_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(models && n_models >= 1 && g && covars && covars->data && out_host, "bad ensemble arguments");
    MHS_REQUIRE(0 <= r0 && r0 <= r1 && r1 <= g->nrow && 0 <= c0 && c0 <= c1 && c1 <= g->ncol, "window outside the grid");
    const int64_t nr = r1 - r0, nc = c1 - c0;
    if (nr == 0 || nc == 0) return MHS_OK;
    for (int k = 0; k < n_models; ++k)
        MHS_REQUIRE(models[k] && covars->n_layers == models[k]->p - 2, "stack has the wrong number of layers for a model");
    const size_t esz = covars->dtype == MHS_F64 ? 8 : covars->dtype == MHS_F32 ? 4 : 2;
    // Large windows with at least two member launches: bands only where bytes cross PCIe (host_window_pipeline)
    if (!getenv("MHS_HOST_BANDS") && nr * nc >= 16000000 && nr >= 64) {
        int first_end = 1, last_start = n_models - 1;
        auto small = [](const mhs_model *m) { return m->kind == K_LM || m->kind == K_NNET || m->kind == K_EARTH; };
        if (small(models[0])) while (first_end < n_models && small(models[first_end]) && models[first_end]->kind > models[first_end - 1]->kind) ++first_end;
        if (small(models[last_start])) while (last_start > first_end && small(models[last_start - 1]) && models[last_start - 1]->kind < models[last_start]->kind) --last_start;
        const size_t need = (size_t)nr * covars->ld * esz * (size_t)covars->n_layers + (size_t)nr * nc * sizeof(double) + 512;
        // Round 4: every member before the last group runs on the UPLOAD bands (no whole-window middle group): the coherent gbm
        // kernel (40 ms per 1e8 cells) is as short as the upload of three float64 planes (43 ms at 56 GB/s), so with gbm alone on
        // the upload bands the last band's gbm ran after the last upload, fully exposed (+22 ms on cfg3); the forest's bands
        // cover it.  A banded launch costs the partly filled last round of its blocks, < 1 ms per band and member.
        if (last_start > first_end) first_end = last_start;
        if (last_start >= first_end && need <= ((size_t)96 << 30))
            return host_window_pipeline(models, weights, n_models, first_end, last_start, wt_total, g, covars, r0, r1, c0, c1, out_host);
    }
    // Band plan.  Measured on cfg3 (tools/r03_host_abi.py): the copies do hide behind the kernels, what a band costs is the
    // partly filled last round of each member kernel, ~3-5 ms per band -- so FEW bands; and all that stays exposed is the
    // first band's upload and the last band's download -- so those two bands are SHORT (8 % of the rows each, at least one
    // round of blocks over the device) and the rows between them go in bands of at most ~100 M cells (they bound the
    // arena: two covariate bands + two result bands).  Small windows go in one piece.  MHS_HOST_BANDS = n: n equal bands.
    std::vector<int64_t> edge;       // band b = rows [edge[b], edge[b + 1])
    edge.push_back(r0);
    const int64_t cells = nr * nc;
    if (const char *e = getenv("MHS_HOST_BANDS")) {
        const int64_t n = std::max<int64_t>(1, std::min<int64_t>(nr, atoll(e))), rp = (nr + n - 1) / n;
        for (int64_t r = r0 + rp; r < r1; r += rp) edge.push_back(r);
    } else if (cells >= 16000000 && nr >= 8) {
        const int64_t ends = std::min<int64_t>(nr / 4, std::max<int64_t>((nr * 8 + 99) / 100, (1500000 + nc - 1) / nc));
        const int64_t mid = nr - 2 * ends, nmid = std::max<int64_t>(1, (mid * nc + 99999999) / 100000000), rp = (mid + nmid - 1) / nmid;
        for (int64_t r = r0 + ends; r < r1 - ends; r += rp) edge.push_back(r);
        edge.push_back(r1 - ends);
    }
    edge.push_back(r1);
    // cuts at multiples of BAND_ALIGN grid rows: gbm_coherent_kernel's tiles are anchored there, so a band sees whole tiles
    // and its cells the same sums as in a resident call
    for (size_t b = 1; b + 1 < edge.size(); ++b) edge[b] = std::min(r1, std::max(r0, (edge[b] + BAND_ALIGN / 2) / BAND_ALIGN * BAND_ALIGN));
    edge.erase(std::unique(edge.begin(), edge.end()), edge.end());
    const int64_t nb = (int64_t)edge.size() - 1;
    int64_t rows_per = 0;
    for (int64_t b = 0; b < nb; ++b) rows_per = std::max(rows_per, edge[(size_t)b + 1] - edge[(size_t)b]);
    const size_t in_bytes = ((size_t)rows_per * covars->ld * esz * (size_t)covars->n_layers + 255) & ~(size_t)255;
    const size_t out_bytes = ((size_t)rows_per * nc * sizeof(double) + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lk(pipe_mutex());
    if (int rc = host_pipe(2 * (in_bytes + out_bytes))) return rc;
    Context &c = ctx();
    PipeDrain drain(c);
    char *in[2] = {c.pipe_arena, c.pipe_arena + in_bytes};
    double *outb[2] = {(double *)(c.pipe_arena + 2 * in_bytes), (double *)(c.pipe_arena + 2 * in_bytes + out_bytes)};
    const bool timing = getenv("MHS_TIMING") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now_ms();
    auto band_rows = [&](int64_t b, int64_t *b0, int64_t *b1) { *b0 = edge[(size_t)b]; *b1 = edge[(size_t)b + 1]; };
    auto upload = [&](int64_t b) -> int {
        const int sl = (int)(b & 1);
        int64_t b0, b1;
        band_rows(b, &b0, &b1);
        if (b >= 2) MHS_HIP(hipStreamWaitEvent(c.pipe_h2d, c.pipe_done[sl], 0));      // band b - 2's kernels read this buffer
        const size_t plane_bytes = (size_t)(b1 - b0) * covars->ld * esz;
        for (int k = 0; k < covars->n_layers; ++k) {
            const char *src = (const char *)covars->data + ((size_t)k * covars->plane_stride + (size_t)b0 * covars->ld) * esz;
            MHS_HIP(hipMemcpyAsync(in[sl] + plane_bytes * k, src, plane_bytes, hipMemcpyHostToDevice, c.pipe_h2d));
        }
        MHS_HIP(hipEventRecord(c.pipe_in[sl], c.pipe_h2d));
        return MHS_OK;
    };
    auto download = [&](int64_t b) -> int {
        const int sl = (int)(b & 1);
        int64_t b0, b1;
        band_rows(b, &b0, &b1);
        MHS_HIP(hipStreamWaitEvent(c.pipe_d2h, c.pipe_done[sl], 0));
        MHS_HIP(hipMemcpyAsync(out_host + (size_t)(b0 - r0) * nc, outb[sl], sizeof(double) * (size_t)((b1 - b0) * nc), hipMemcpyDeviceToHost,
                               c.pipe_d2h));
        MHS_HIP(hipEventRecord(c.pipe_out[sl], c.pipe_d2h));
        return MHS_OK;
    };
    if (int rc = upload(0)) return rc;
    for (int64_t b = 0; b < nb; ++b) {
        const int sl = (int)(b & 1);
        int64_t b0, b1;
        band_rows(b, &b0, &b1);
        hipStream_t cs = c.pipe_comp;
        MHS_HIP(hipStreamWaitEvent(cs, c.pipe_in[sl], 0));
        if (b >= 2) MHS_HIP(hipStreamWaitEvent(cs, c.pipe_out[sl], 0));       // band b - 2's result has left this buffer
        // The device copy is described with the PARENT grid's affine (cell centres stay bit-identical): plane k, absolute
        // row r lives at base + (k * plane_stride + r * ld) * esz, so the base is shifted back by b0 rows and plane_stride
        // skips the rows that were shipped.
        PredGeom pg;
        StackDev sd;
        if (int rc = make_geom(g, b0, b1, c0, c1, nc, &pg)) return rc;
        sd.data = in[sl] - (size_t)b0 * covars->ld * esz; sd.C = covars->n_layers; sd.dtype = covars->dtype;
        sd.plane_stride = (b1 - b0) * covars->ld; sd.ld = covars->ld; sd.nodata = covars->nodata;
        sd.has_nodata = !std::isnan(covars->nodata); sd.all_from_planes = 0;
        if (int rc = launch_members(models, weights, n_models, sd, pg, 0, outb[sl], cs, g)) return rc;
        hipLaunchKernelGGL(scale_window_kernel, dim3((unsigned)(((b1 - b0) * nc + 255) / 256)), dim3(256), 0, cs,
                           outb[sl], (int)(b1 - b0), (int)nc, nc, wt_total);
        MHS_HIP(hipGetLastError());
        MHS_HIP(hipEventRecord(c.pipe_done[sl], cs));
        const double t0 = now_ms();
        if (b + 1 < nb) if (int rc = upload(b + 1)) return rc;
        const double t1 = now_ms();
        if (b >= 1) if (int rc = download(b - 1)) return rc;
        if (timing) fprintf(stderr, "[mhs_ensemble_predict] band %lld launched at %.1f ms: upload of the next %.1f ms, download of the previous %.1f ms\n",
                            (long long)b, t0 - t_start, t1 - t0, now_ms() - t1);
    }
    if (int rc = download(nb - 1)) return rc;
    MHS_HIP(hipStreamSynchronize(c.pipe_d2h));
    MHS_HIP(hipStreamSynchronize(c.pipe_comp));
    drain.done = true;
    return MHS_OK;
}

int mhs_predict_points(const mhs_model *m, const double *X, int64_t n, double *out_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(m && out_host && (X || n == 0) && n >= 0 && n < (1LL << 31), "bad arguments");
    if (n == 0) return MHS_OK;
    hipStream_t s = ctx().stream;
    DevBuf<double> dx, dout;
    MHS_HIP(dx.alloc((size_t)n * m->p));
    MHS_HIP(dout.alloc((size_t)n));
    MHS_HIP(hipMemcpyAsync(dx.p, X, sizeof(double) * (size_t)n * m->p, hipMemcpyHostToDevice, s));
    PredGeom pg;
    pg.xmin = pg.ymax = 0; pg.xres = pg.yres = 1; pg.r0 = pg.c0 = 0; pg.nr = 1; pg.nc = (int)n; pg.ld_out = n;
    StackDev sd;
    sd.data = dx.p; sd.C = m->p; sd.dtype = MHS_F64; sd.plane_stride = n; sd.ld = n; sd.nodata = NAN;
    sd.has_nodata = 0; sd.all_from_planes = 1;
    if (int rc = launch_model(m, sd, pg, 1.0, 0, dout.p, s)) return rc;
    MHS_HIP(hipMemcpyAsync(out_host, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

int mhs_gbm_probe_last(const mhs_model *m, int64_t *cost, int64_t *count) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(m && cost && count, "NULL argument");
    *cost = 0; *count = 0;
    if (m->kind != K_GBM || !m->gbm_probe || m->gbm_probe_next.load() == 0) return MHS_OK;
    int h[2 * GBC_PROBE_BLOCKS];
    MHS_HIP(hipDeviceSynchronize());
    MHS_HIP(hipMemcpy(h, m->gbm_probe + (size_t)((m->gbm_probe_next.load() - 1) % GBC_PROBE_SLOTS) * 2 * GBC_PROBE_BLOCKS, sizeof(h),
                      hipMemcpyDeviceToHost));
    for (int b = 0; b < GBC_PROBE_BLOCKS; ++b) { *cost += h[2 * b]; *count += h[2 * b + 1]; }
    return MHS_OK;
}

int mhs_model_info(const mhs_model *m, int *kind, int *p, int64_t *n_trees) {
    MHS_REQUIRE(m != nullptr, "NULL model");
    if (kind) *kind = m->kind;
    if (p) *p = m->p;
    if (n_trees) *n_trees = (m->kind == K_GBM || m->kind == K_RF) ? m->n_trees : 0;
    return MHS_OK;
}

int mhs_gbm_staged_points(const mhs_model *m, const double *X, int64_t n, int step, double *out_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(m && X && out_host && n >= 1 && n < (1LL << 31) && step >= 1, "bad arguments");
    MHS_REQUIRE(m->kind == K_GBM, "not a gbm model");
    const int stages = m->n_trees / step;
    if (stages == 0) return MHS_OK;
    hipStream_t s = ctx().stream;
    DevBuf<double> dx, dout;
    MHS_HIP(dx.alloc((size_t)n * m->p));
    MHS_HIP(dout.alloc((size_t)n * stages));
    MHS_HIP(hipMemcpyAsync(dx.p, X, sizeof(double) * (size_t)n * m->p, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gbm_staged_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, m->nodes, m->tree_off,
                       m->n_trees, m->init_f, dx.p, n, step, dout.p);
    MHS_HIP(hipGetLastError());
    MHS_HIP(hipMemcpyAsync(out_host, dout.p, sizeof(double) * (size_t)n * stages, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

int mhs_residual_points(const mhs_model *const *models, const double *weights, int n_models, double wt_total,
                        const double *X, const double *resp, int64_t n, double *out_host) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(models && weights && X && resp && out_host && n_models >= 1 && n >= 1 && n < (1LL << 31), "bad arguments");
    MHS_REQUIRE(n_models <= 8, "at most 8 members");
    const int p = models[0]->p;
    for (int k = 0; k < n_models; ++k) MHS_REQUIRE(models[k] && models[k]->p == p, "models disagree on the number of predictors");
    hipStream_t s = ctx().stream;
    // one grow-only scratch: X (n x p), the members' predictions (n x K), resp, out
    Context &c = ctx();
    const size_t need = (size_t)n * ((size_t)p + (size_t)n_models + 2);
    if (need > c.points_arena_cap) {
        if (c.points_arena) { (void)hipStreamSynchronize(s); (void)hipFree(c.points_arena); c.points_arena = nullptr; c.points_arena_cap = 0; }
        MHS_HIP(hipMalloc((void **)&c.points_arena, need * sizeof(double)));
        c.points_arena_cap = need;
    }
    double *dx = c.points_arena, *dpred = dx + (size_t)n * p, *dresp = dpred + (size_t)n * n_models, *dout = dresp + n;
    MHS_HIP(hipMemcpyAsync(dx, X, sizeof(double) * (size_t)n * p, hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemcpyAsync(dresp, resp, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    PredGeom pg;
    pg.xmin = pg.ymax = 0; pg.xres = pg.yres = 1; pg.r0 = pg.c0 = 0; pg.nr = 1; pg.nc = (int)n; pg.ld_out = n;
    StackDev sd;
    sd.data = dx; sd.C = p; sd.dtype = MHS_F64; sd.plane_stride = n; sd.ld = n; sd.nodata = NAN;
    sd.has_nodata = 0; sd.all_from_planes = 1;
    for (int k = 0; k < n_models; ++k)
        if (int rc = launch_model(models[k], sd, pg, 1.0, 0, dpred + (size_t)k * n, s)) return rc;
    ResidualArgs ra;
    for (int k = 0; k < 8; ++k) ra.w[k] = k < n_models ? weights[k] : 0.0;
    hipLaunchKernelGGL(residual_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dpred, dresp, n, n_models, ra,
                       wt_total, dout);
    MHS_HIP(hipGetLastError());
    MHS_HIP(hipMemcpyAsync(out_host, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

int mhs_scale_add_dev(const double *a, double divisor, const double *b, double *out, int64_t n, void *stream) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(a && out && n >= 0, "bad arguments");
    if (n == 0) return MHS_OK;
    hipLaunchKernelGGL(scale_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pick_stream(stream), a, divisor, b, out, n);
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

}  // extern "C"
