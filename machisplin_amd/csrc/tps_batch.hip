// Many small thin-plate-spline fits in ONE launch: fields::Tps on the 130-250 stations of every tile of the
// reference's tiled Step 3 (V73:690-738; SURVEY.md 8d "a different GPU regime -- tens to hundreds of small fits").
//
// One workgroup of 512 threads fits one spline, start to finish, without leaving the compute unit:
//
//   Gram matrix          K_rc = sw_r sw_c phi(|x_r - x_c|^2), computed straight into REGISTERS
//   null-space projection  A = H2 H1 H0 K H0 H1 H2: three two-sided Householder updates with the host's reflectors
//                          of [1 u v] (the same update the reduction uses, so one piece of code does both)
//   tridiagonalisation   classical Householder on B = A[3:, 3:], g <- Q'g carried along; the reflectors go to a
//                          global scratch (read back once, by one wave, for the back-transform)
//   GCV search           TridiagGcv::find_lambda's procedure (tps_gcv_host.hip) with its independent evaluations
//                          spread over the block's threads: extreme eigenvalues by 256-way section of Sturm counts,
//                          the bracket's 2 x 20 candidates at once, the 200-point grid at once, the golden section as
//                          rounds of 8 levels of its decision tree (255 evaluations in flight, the walk replays the
//                          sequential algorithm's comparisons exactly)
//   solve                q = (T + lambda I)^-1 g, c2 = Q q, d = R^-1 (w1 - A[0:3, 3:] c2), c = W^1/2 H0 H1 H2 [0; c2]
//   output               coefficients, and the evaluation's knot records written in the far-field plan's bin order
//
// The matrix in registers.  The 512 threads form a 32 x 16 grid (a, b); thread (a, b) holds the elements
// (a + 32 i, b + 16 j) of the symmetric matrix for the column blocks j <= 2 i + 1: the strict lower triangle of the
// 32 x 32 blocks (j <= 2 i - 1) and the diagonal blocks kept whole (j = 2 i, 2 i + 1: both mirror images of an
// off-diagonal element of a diagonal block are stored, by different threads) -- NS (NS + 1) doubles for n <= 32 NS,
// 72 at n = 256 (144 of the 256 registers a wave has at two waves per SIMD), half of what full storage needs and
// what makes 256 stations fit the register file.  A step is  p = tau A v  (row sums reduced over b: two lane halves +
// 8 waves through LDS; column sums over a: DPP), w = p - 1/2 tau (p'v) v,  A -= v w' + w v': three barriers, ~1 us,
// against 9 us for the one-block kernel of tps_fit.hip that keeps the matrix in L2.  Dead rows and columns (the reflectors' own) are handled by zeros in v
// and w, so the loops carry no per-element tests, only block-uniform skips of dead blocks.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "devmath.h"
#include "tps_batch.h"

namespace mhs {

constexpr int SB_THREADS = 512;
constexpr int SB_EV = 256;           // GCV evaluations in flight (their pivots and half-solved right-hand sides live in global scratch)
constexpr int SB_TREE_DEPTH = 8;     // golden-section levels per round: 2^8 - 1 = 255 evaluations

struct SbShared {
    double mx[16][SB_THREADS];        // n > 224 only: the matrix's first 32 columns (16 elements per thread; the registers hold 56, not 72)
    double u[SB_NMAX], v[SB_NMAX], sw[SB_NMAX];
    double vs[2][SB_NMAX];            // the step's Householder vector (double-buffered: the next owner writes while others still update)
    double ps[SB_NMAX];               // p = tau A v
    double pcs[SB_NMAX];              // column sums
    double part[16][SB_NMAX];         // row sums per wave (8 used); after the reduction: the search's tables (fs, xs, ...)
    double ta[SB_NMAX], tb[SB_NMAX], tg[SB_NMAX], ttau[SB_NMAX];   // tridiagonal, rotated data, reflector scalars -- by GLOBAL row
    double at[3][SB_NMAX];            // A[0:3, :] after the projection
    double tq[SB_NMAX];               // solution in the tridiagonal basis, then c2, then c~
    double red[4][2];
    double sc[2];                     // tau of the step (per buffer)
    double misc[32];
    unsigned long long stamp[8];      // 100 MHz counter at the phase boundaries
    int imisc[8];
    double2 tab[LOG_TAB_N];           // last: everything the step loop touches stays within the 64 KB an LDS instruction's offset field reaches
};

// ---------------------------------------------------------------------------------------------- wave helpers
// after this every lane of a 16-lane row holds the row's total
__device__ __forceinline__ double row16_sum(double x) {
    x += dpp_fetch<0xB1, 0xf>(x);
    x += dpp_fetch<0x4E, 0xf>(x);
    x += dpp_fetch<0x141, 0xf>(x);
    x += dpp_fetch<0x140, 0xf>(x);
    return x;
}
// total of each 32-lane half, valid in lanes 16-31 (lower half) and 48-63 (upper half)
__device__ __forceinline__ double half_sum_upper_row(double x) {
    x = row16_sum(x);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x142, 0xA, 0xf, false);   // row_bcast:15 into rows 1, 3
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x142, 0xA, 0xf, false);
    return x + __hiloint2double(hi, lo);
}
// total of the 32-lane half `half` (wave-uniform), in every lane
__device__ __forceinline__ double half_total(double x, int half) {
    x = row16_sum(x);
    return lane_value(x, 32 * half) + lane_value(x, 32 * half + 16);
}
// x of lane l + x of lane l ^ 32, in every lane
__device__ __forceinline__ double xor32_sum(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const auto r1 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(r2[0], r1[0]) + __hiloint2double(r2[1], r1[1]);
}

// The lane number, recomputed where it is used: a thread's few index values (a, b and the LDS addresses made of them) are
// loop invariants the register allocator otherwise parks in scratch memory and fetches back -- a round trip to memory
// each, a dozen per step -- to make room for the matrix; two instructions rebuild them.
__device__ __forceinline__ int sb_lane() {
    int x;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
    return x;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding GLOBAL stores
// (s_waitcnt vmcnt(0)) -- a round trip to L2 per step here, where the reflector just written out is not read back
// before the back-transform, behind a full barrier.
__device__ __forceinline__ void sb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 1 / x to double precision without the IEEE division sequence's scaling and fix-up (x is a pivot: finite, non-zero,
// far from the exponent range's ends): v_rcp_f64 + two Newton steps.  A pivot recurrence is a chain of dependent
// divisions -- the division IS its run time.
__device__ __forceinline__ double sb_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// x of the lane whose number differs in one bit: quad permutes (DPP) for bits 0 and 1, ds_swizzle's bit mode for bits 2-4
template <int CTRL>
__device__ __forceinline__ double sb_dpp(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int XOR>
__device__ __forceinline__ double sb_swz(double x) {
    constexpr int PAT = (XOR << 10) | 0x1f;     // bit mode: and_mask 0x1f, or_mask 0, xor_mask XOR
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), PAT);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), PAT);
    return __hiloint2double(hi, lo);
}
// Sixteen values per lane summed over the 32 lanes of each half-wave, by halving: at every stage a lane hands half of
// its partial sums to its partner and keeps the other half, so 8 + 4 + 2 + 1 + 1 values cross lanes instead of 16 x 5.
// Result: the total of value (lane & 15), in every lane.
__device__ __forceinline__ double sb_colsum16(const double (&x)[16], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
    double s1[8], s2[4], s3[2];
#pragma unroll
    for (int t = 0; t < 8; ++t) s1[t] = (b0 ? x[2 * t + 1] : x[2 * t]) + sb_dpp<0xB1>(b0 ? x[2 * t] : x[2 * t + 1]);
#pragma unroll
    for (int t = 0; t < 4; ++t) s2[t] = (b1 ? s1[2 * t + 1] : s1[2 * t]) + sb_dpp<0x4E>(b1 ? s1[2 * t] : s1[2 * t + 1]);
#pragma unroll
    for (int t = 0; t < 2; ++t) s3[t] = (b2 ? s2[2 * t + 1] : s2[2 * t]) + sb_swz<4>(b2 ? s2[2 * t] : s2[2 * t + 1]);
    const double s4 = (b3 ? s3[1] : s3[0]) + sb_swz<8>(b3 ? s3[0] : s3[1]);
    return s4 + sb_swz<16>(s4);
}

#define SB_IDX(i, j) ((i) * ((i) + 1) + (j))      // row block i (32 rows), column block j (16 columns), j <= 2 i + 1
// At NS = 8 (225..256 stations) the 72 doubles per thread leave the compiler no room and it parks a part of the matrix in
// scratch memory, ~100 round trips per step; the two column blocks that die first (columns 0..31: 16 elements per thread)
// live in LDS instead, thread-contiguous (no bank conflicts), and cost LDS bandwidth only during the first 64 steps.
#define SB_INLDS(j) (NS == 8 && (j) < 2)
#define SB_MLD(i, j) (SB_INLDS(j) ? S.mx[(i) * 2 + (j)][tid] : M[SB_IDX(i, j)])
#define SB_MST(i, j, val) do { if (SB_INLDS(j)) S.mx[(i) * 2 + (j)][tid] = (val); else M[SB_IDX(i, j)] = (val); } while (0)

// element (row block si, column block sj) of this thread, blocks given at run time (block-uniform)
template <int NS>
__device__ __forceinline__ double sb_get(const double (&M)[NS * (NS + 1)], const SbShared &S, int tid, int si, int sj) {
    double x = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j <= 2 * i + 1; ++j)
            if (i == si && j == sj) x = SB_MLD(i, j);
    return x;
}

// ---------------------------------------------------------------------------------------------- GCV criterion
// TridiagGcv::eval (tps_gcv_host.hip) for one lambda by one thread: the same recurrences (divisions as multiplications by
// a Newton-refined reciprocal, equal to rounding); the forward pivots and the half-solved right-hand side go through the
// thread's column of the global scratch.
// a, b, g: LDS (T's diagonal / off-diagonal, rotated data), m entries.  q_out: LDS or NULL.
__device__ __forceinline__ void sb_eval(const double *a, const double *b, const double *g, int m, int n, int N, double pure_ss,
                                        double lam, double *__restrict__ scr, int t, double *gcv_out, double *tr_out,
                                        double *q_out) {
    // forward: pivots dp_i = a_i + lam - b_{i-1}^2 / dp_{i-1} and the half-solved right-hand side; what the backward
    // pass needs of them -- 1 / dp_i, dp_i and the half-solved entry -- goes through the scratch column
    double dp = a[0] + lam, q = g[0];
    double ip = sb_rcp(dp);
    scr[t] = dp; scr[SB_EV + t] = q;
    for (int i = 1; i < m; ++i) {
        const double bi = b[i - 1];
        const double l = bi * ip;
        const double dn = a[i] + lam - bi * l;
        const double qn = g[i] - l * q;
        dp = dn; q = qn;
        ip = sb_rcp(dp);
        scr[(size_t)(2 * i) * SB_EV + t] = dp;
        scr[(size_t)(2 * i + 1) * SB_EV + t] = q;
    }
    constexpr int C = 8;
    double pd[C], pq[C];
    int top = m - 1;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = top - c;
        pd[c] = i >= 0 ? scr[(size_t)(2 * i) * SB_EV + t] : 1.0;
        pq[c] = i >= 0 ? scr[(size_t)(2 * i + 1) * SB_EV + t] : 0.0;
    }
    double dm = 0.0, im = 0.0, qn = 0.0, tr_inv = 0.0, qq = 0.0;
    while (top >= 0) {
        double cd[C], cq[C], ci[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { cd[c] = pd[c]; cq[c] = pq[c]; ci[c] = sb_rcp(pd[c]); }      // off the chain
#pragma unroll
        for (int c = 0; c < C; ++c) {      // next chunk's loads are in flight while this one's chain runs
            const int i = top - C - c;
            pd[c] = i >= 0 ? scr[(size_t)(2 * i) * SB_EV + t] : 1.0;
            pq[c] = i >= 0 ? scr[(size_t)(2 * i + 1) * SB_EV + t] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int i = top - c;
            if (i >= 0) {
                const double al = a[i] + lam;
                if (i == m - 1) { dm = al; qn = cq[c] * ci[c]; }
                else { const double bi = b[i]; dm = al - bi * bi * im; qn = (cq[c] - bi * qn) * ci[c]; }
                im = sb_rcp(dm);
                tr_inv += sb_rcp(cd[c] + dm - al);
                qq += qn * qn;
                if (q_out) q_out[i] = qn;
            }
        }
        top -= C;
    }
    const double rss = lam * lam * qq;
    const double tr = 3.0 + (double)m - lam * tr_inv;
    double mse = rss / (double)n;
    if (N - n > 0) mse += pure_ss / (double)(N - n);
    const double den = 1.0 - tr / (double)n;
    if (gcv_out) *gcv_out = den > 0 ? mse / (den * den) : NAN;
    if (tr_out) *tr_out = tr;
}

// eigenvalues of T strictly below x (Sturm sequence, as sturm_count of tps_gcv_host.hip)
__device__ __forceinline__ int sb_sturm(const double *a, const double *b, int m, double x) {
    int cnt = 0;
    double q = a[0] - x;
    if (q < 0) ++cnt;
    for (int i = 1; i < m; ++i) {
        const double den = (q != 0.0) ? q : 1e-300;
        q = a[i] - x - b[i - 1] * b[i - 1] * sb_rcp(den);
        if (q < 0) ++cnt;
    }
    return cnt;
}

// One step of golden_section() / of the converged search of TridiagGcv::find_lambda: the state after taking branch
// `d` (the comparison's outcome), and the point that has to be evaluated next.
struct GoldState { double x0, x1, x2, x3; };
__device__ __forceinline__ double gold_step_fields(GoldState &s, bool f2_lt_f1) {
    const double r = 0.61803399, con = 1.0 - r;
    if (f2_lt_f1) { s.x0 = s.x1; s.x1 = s.x2; s.x2 = r * s.x1 + con * s.x3; return s.x2; }
    s.x3 = s.x2; s.x2 = s.x1; s.x1 = r * s.x2 + con * s.x0; return s.x1;
}
// converged mode: x0 = lo, x3 = hi, x1 < x2 the interior points (in log lambda)
__device__ __forceinline__ double gold_step_conv(GoldState &s, bool f1_lt_f2) {
    const double r = 0.5 * (sqrt(5.0) - 1.0);
    if (f1_lt_f2) { s.x3 = s.x2; s.x2 = s.x1; s.x1 = s.x3 - r * (s.x3 - s.x0); return s.x1; }
    s.x0 = s.x1; s.x1 = s.x2; s.x2 = s.x0 + r * (s.x3 - s.x0); return s.x2;
}

// ---------------------------------------------------------------------------------------------- the fit
// Part 1 (matrix in registers): Gram matrix, projection, tridiagonalisation.  Leaves in LDS the tridiagonal (ta, tb),
// the rotated data (tg), the reflectors' scalars (ttau), the three projected rows (at); the reflectors in Vg.
template <int NS>
__device__ __forceinline__ void sb_reduce(const SmallJob *__restrict__ Jp, const double *__restrict__ in,
                                          double *__restrict__ Vg, int ldv, SbShared &S) {
    constexpr int NE = NS * (NS + 1);
    constexpr int NR = NS * 32;
    constexpr int NC = 2 * NS;                     // column blocks of 16
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = Jp->n;
#define SB_THREAD_IDS() const int lane = sb_lane(); const int tid = wave * 64 + lane; const int a = lane & 31, b = wave * 2 + (lane >> 5); (void)tid; (void)a; (void)b

    // ---- Gram matrix into registers
    double M[NE];
    {
    SB_THREAD_IDS();
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = b + 16 * j;
        const double uc = S.u[c], vc = S.v[c], sc = S.sw[c];
#pragma unroll
        for (int i = j >> 1; i < NS; ++i) {
            const int r = a + 32 * i;
            const double dx = S.u[r] - uc, dy = S.v[r] - vc;
            const double d2 = fma(dy, dy, dx * dx);
            const double k = S.sw[r] * (0.5 / (8.0 * M_PI)) * sc * r2logr2(d2, S.tab);
            SB_MST(i, j, (r < n && c < n) ? k : 0.0);
        }
    }
    if (tid == 0) S.stamp[1] = wall_clock64();
    // the three given reflectors: into LDS (S.at is free until the loop is over) and into the reflector store
    if (tid < NR) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double hv = tid < n ? in[(size_t)(3 + k) * n + tid] : 0.0;
            S.at[k][tid] = hv;
            Vg[(size_t)k * ldv + tid] = hv;
        }
    }
    }
    __syncthreads();

    // ---- three projection steps (given reflectors), then the tridiagonalisation of the trailing m x m block
#ifdef SB_PHASE_TRACE
    unsigned long long tr_acc[7] = {0, 0, 0, 0, 0, 0, 0}, tr_last = wall_clock64();
#ifndef SB_TRACE_STEP
#define SB_TRACE_STEP kk
#endif
#define SB_TR(k) do { const unsigned long long now_ = wall_clock64(); if (kk == (SB_TRACE_STEP)) tr_acc[k] += now_ - tr_last; tr_last = now_; } while (0)
#else
#define SB_TR(k) do {} while (0)
#endif
    for (int kk = 0; kk <= n - 3; ++kk) {
        const int buf = kk & 1;
        const bool proj = kk < 3;
        const int dl = proj ? 0 : kk + 1;          // rows / columns below dl are dead (final) for this step
        const int glo = dl >> 6;                   // groups of four 16-column blocks below glo are dead as a whole
        const int ilo = dl >> 5;                   // 32-row blocks below ilo
        if (proj) {
            SB_THREAD_IDS();
            if (tid < NR) S.vs[buf][tid] = S.at[kk][tid];
            if (tid == 0) { const double ht = Jp->htau[kk]; S.sc[buf] = ht; S.ttau[kk] = ht; }
        } else if (wave == ((kk & 15) >> 1)) {
            // the wave that holds column kk builds its Householder vector (dlarfg)
            SB_THREAD_IDS();
            if (kk == 3 && lane == 0) S.stamp[2] = wall_clock64();
            const int bk = kk & 15, jk = kk >> 4, half = bk & 1, r1 = kk + 1;
            const bool mine = (lane >> 5) == half;
            // column kk of this thread's blocks: a jump on the (block-uniform) column block instead of a select per element
            double x[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) x[i] = 0.0;
#define SB_XCASE(J) case J: if constexpr ((J) < NC) { _Pragma("unroll") for (int i = (J) >> 1; i < NS; ++i) x[i] = SB_MLD(i, J); } break;
            switch (jk) {
                SB_XCASE(0) SB_XCASE(1) SB_XCASE(2) SB_XCASE(3) SB_XCASE(4) SB_XCASE(5) SB_XCASE(6) SB_XCASE(7)
                SB_XCASE(8) SB_XCASE(9) SB_XCASE(10) SB_XCASE(11) SB_XCASE(12) SB_XCASE(13) SB_XCASE(14) SB_XCASE(15)
                default: break;
            }
#undef SB_XCASE
            // alpha = A[kk + 1][kk] and the diagonal entry A[kk][kk] sit in ONE known lane each: a readlane, not a reduction
            double ssl = 0.0, xa = 0.0, xd = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int r = a + 32 * i;
                if (mine && r > r1) ssl += x[i] * x[i];
                if (i == (r1 >> 5)) xa = x[i];
                if (i == (kk >> 5)) xd = x[i];
            }
            const double ss = half_total(ssl, half);
            const double alpha = lane_value(xa, 32 * half + (r1 & 31)), dkk = lane_value(xd, 32 * half + (kk & 31));
            double beta = alpha, tk = 0.0, scal = 0.0;
            if (ss != 0.0) {
                beta = -copysign(sqrt(alpha * alpha + ss), alpha);
                tk = (beta - alpha) / beta;
                scal = 1.0 / (alpha - beta);
            }
            if (mine) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int r = a + 32 * i;
                    const double vi = r == r1 ? 1.0 : (r > r1 ? x[i] * scal : 0.0);
                    S.vs[buf][r] = vi;
                    Vg[(size_t)kk * ldv + r] = vi;
                }
                if (a == 0) { S.ta[kk] = dkk; S.tb[kk] = beta; S.ttau[kk] = tk; S.sc[buf] = tk; }
            }
        }
        SB_TR(0);
        sb_barrier();                                                       // A: v is there
        SB_TR(1);
        const double tau = S.sc[buf];
        {   // p = A v: row sums (over b) and column sums (over a)
            SB_THREAD_IDS();
            double vr[NS], prow[NS], pc[16];
#pragma unroll
            for (int i = 0; i < NS; ++i) { vr[i] = S.vs[buf][a + 32 * i]; prow[i] = 0.0; }
#pragma unroll
            for (int j = 0; j < 16; ++j) pc[j] = 0.0;
#pragma unroll
            for (int g = 0; g < (NC + 3) / 4; ++g) {
                if (g >= glo) {         // a group of four column blocks (64 columns): dead ones inside it multiply zeros
                    double vc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) vc[q] = 4 * g + q < NC ? S.vs[buf][b + 16 * (4 * g + q)] : 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j = 4 * g + q;
                        if (j < NC) {
#pragma unroll
                            for (int i = j >> 1; i < NS; ++i) {
                                const double e = SB_MLD(i, j);
                                prow[i] = fma(e, vc[q], prow[i]);
                                if (j <= 2 * i - 1) pc[j] = fma(e, vr[i], pc[j]);      // strictly lower block: stands for its mirror image too
                            }
                        }
                    }
                }
            }
            const double ptot = sb_colsum16(pc, lane);
            if ((lane & 16) == 0) S.pcs[b + 16 * (lane & 15)] = ptot;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if (i >= ilo) {
                    const double t = xor32_sum(prow[i]);
                    if (lane < 32) S.part[wave][a + 32 * i] = t;
                }
            }
        }
        SB_TR(2);
        sb_barrier();                                                       // B: partial sums are there
        SB_TR(3);
        double fin_v = 0.0, fin_g = 0.0;
        if (wave < SB_NMAX / 64) {
            SB_THREAD_IDS();
            const int r = tid;
            double p = 0.0;
            if (r >= dl && r < NR) {
                double s = S.pcs[r];
#pragma unroll
                for (int w = 0; w < SB_THREADS / 64; ++w) s += S.part[w][r];
                p = tau * s;
            }
            fin_v = r < NR ? S.vs[buf][r] : 0.0;
            fin_g = (r >= 3) ? S.tg[r] : 0.0;
            S.ps[r] = p;
            const double pv = wave_sum(p * fin_v), vg = wave_sum(fin_v * fin_g);
            if (lane == 0) { S.red[wave][0] = pv; S.red[wave][1] = vg; }
        }
        SB_TR(4);
        sb_barrier();                                                       // C: p and the two dot products are there
        SB_TR(5);
        const double pv = (S.red[0][0] + S.red[1][0]) + (S.red[2][0] + S.red[3][0]);
        const double alpha = 0.5 * tau * pv;
        if (!proj && wave < SB_NMAX / 64) {
            SB_THREAD_IDS();
            const double vg = (S.red[0][1] + S.red[1][1]) + (S.red[2][1] + S.red[3][1]);
            if (tid >= 3) S.tg[tid] = fin_g - tau * vg * fin_v;
        }
        {   // A -= v w' + w v'
            SB_THREAD_IDS();
            double vr[NS], wr[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                vr[i] = S.vs[buf][a + 32 * i];
                wr[i] = S.ps[a + 32 * i] - alpha * vr[i];
            }
#pragma unroll
            for (int g = 0; g < (NC + 3) / 4; ++g) {
                if (g >= glo) {
                    double vc[4], wc[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j = 4 * g + q;
                        vc[q] = j < NC ? S.vs[buf][b + 16 * j] : 0.0;
                        wc[q] = j < NC ? S.ps[b + 16 * j] - alpha * vc[q] : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j = 4 * g + q;
                        if (j < NC) {
#pragma unroll
                            for (int i = j >> 1; i < NS; ++i) {
                                double e = SB_MLD(i, j);
                                e = fma(-vr[i], wc[q], e);
                                e = fma(-wr[i], vc[q], e);
                                SB_MST(i, j, e);
                            }
                        }
                    }
                }
            }
        }
        SB_TR(6);
        // no barrier: the next step's owner writes the OTHER v buffer; ps / part / pcs / red are rewritten only behind
        // the next step's barriers A and B, which every thread reaches after its update
    }
#ifdef SB_PHASE_TRACE
    if (threadIdx.x == 64 * SB_TRACE_WAVE) for (int k = 0; k < 7; ++k) S.misc[20 + k] = 0.01 * (double)tr_acc[k];
#endif
    {   // the last 2 x 2 block of the tridiagonal, and the three projected rows
        SB_THREAD_IDS();
        const int r = n - 1, c = n - 2;
        if (a == (c & 31) && b == (c & 15)) S.ta[c] = sb_get<NS>(M, S, tid, c >> 5, c >> 4);
        if (a == (r & 31) && b == (c & 15)) S.tb[c] = sb_get<NS>(M, S, tid, r >> 5, c >> 4);
        if (a == (r & 31) && b == (r & 15)) S.ta[r] = sb_get<NS>(M, S, tid, r >> 5, r >> 4);
        if (b < 3) {
#pragma unroll
            for (int i = 0; i < NS; ++i) S.at[b][a + 32 * i] = SB_MLD(i, 0);
        }
    }
    __syncthreads();
#undef SB_THREAD_IDS
}

// Part 2 (nothing in registers across phases): lambda, the solve, the back-transform, the outputs.
__device__ __noinline__ void sb_finish(const SmallJob *__restrict__ Jp, const int *__restrict__ perm,
                                       const double *__restrict__ Vg, int ldv, double *__restrict__ scr, SbShared &S,
                                       double *__restrict__ c_out, Knot *__restrict__ knots_out, SmallResult *__restrict__ res) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = Jp->n, m = n - 3, N = Jp->N, gcv_mode = Jp->gcv_mode;
    const double pure_ss = Jp->pure_ss, lambda_in = Jp->lambda;
    const double *ta = S.ta + 3, *tb = S.tb + 3, *tg = S.tg + 3;
    double *fs = &S.part[0][0];            // 1024 doubles: criterion values
    double *xs = &S.part[4][0];            // 1024 doubles: abscissae
    double *gl = &S.part[8][0];            // 256: compacted grid lambdas
    double *gv = &S.part[9][0];            // 256: compacted grid values
    double lam = lambda_in;
    int status = 0;
    if (tid == 0) { S.stamp[3] = wall_clock64(); S.stamp[4] = S.stamp[5] = S.stamp[6] = S.stamp[3]; }
    if (isnan(lambda_in)) {
        // extreme eigenvalues: 256-way section of the Gershgorin interval on Sturm counts, both ends at once
        if (tid == 0) {
            double lo = ta[0], hi = ta[0];
            for (int i = 0; i < m; ++i) {
                const double rr = (i > 0 ? fabs(tb[i - 1]) : 0.0) + (i + 1 < m ? fabs(tb[i]) : 0.0);
                lo = fmin(lo, ta[i] - rr);
                hi = fmax(hi, ta[i] + rr);
            }
            S.misc[0] = lo; S.misc[1] = hi; S.misc[2] = lo; S.misc[3] = hi;
            S.imisc[0] = 256; S.imisc[1] = 256;
        }
        __syncthreads();
        for (int round = 0; round < 20; ++round) {
            const int e = tid >> 8, jx = tid & 255;
            const double lo0 = S.misc[0], hi0 = S.misc[1], lo1 = S.misc[2], hi1 = S.misc[3];
            const bool done0 = !(hi0 - lo0 > 4e-16 * fmax(fabs(lo0), fabs(hi0)));
            const bool done1 = !(hi1 - lo1 > 4e-16 * fmax(fabs(lo1), fabs(hi1)));
            if (done0 && done1) break;
            {
                const double lo = e ? lo1 : lo0, hi = e ? hi1 : hi0;
                const double x = lo + (hi - lo) * ((double)(jx + 1) / 257.0);
                const int k = e ? 0 : m - 1;
                xs[tid] = x;
                if (sb_sturm(ta, tb, m, x) > k) atomicMin(&S.imisc[e], jx);
            }
            __syncthreads();
            if (tid < 2) {
                const int jmin = S.imisc[tid];
                const double lo = S.misc[2 * tid], hi = S.misc[2 * tid + 1];
                const double nhi = jmin < 256 ? xs[tid * 256 + jmin] : hi;
                const double nlo = jmin > 0 ? xs[tid * 256 + jmin - 1] : lo;
                S.misc[2 * tid] = fmax(lo, fmin(nlo, nhi)); S.misc[2 * tid + 1] = fmin(hi, fmax(nlo, nhi));
                S.imisc[tid] = 256;
            }
            __syncthreads();
        }
        if (tid == 0) S.stamp[4] = S.stamp[5] = S.stamp[6] = wall_clock64();
        const double emax = 0.5 * (S.misc[0] + S.misc[1]);
        const double emin = fmax(0.5 * (S.misc[2] + S.misc[3]), 1e-300);
        __syncthreads();
        // the bracket: l1 = emax 4^k until trA < 3.05, l2 = emin / 4^k until trA > 0.95 n (k < 20), all candidates at once
        if (tid < 64) {
            const int k = tid & 31;
            if (k < 20) {
                const double l = tid < 32 ? ldexp(emax, 2 * k) : ldexp(emin, -2 * k);
                double tr;
                sb_eval(ta, tb, tg, m, n, N, pure_ss, l, scr, tid, nullptr, &tr, nullptr);
                fs[tid] = tr;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int k1 = 20, k2 = 20;
            for (int k = 19; k >= 0; --k) { if (fs[k] < 3.0 + 0.05) k1 = k; if (fs[32 + k] > 0.95 * (double)n) k2 = k; }
            S.misc[4] = log(ldexp(emin, -2 * k2));      // la
            S.misc[5] = log(ldexp(emax, 2 * k1));       // lb
        }
        __syncthreads();
        const double la = S.misc[4], lb = S.misc[5];
        if (tid < 200) {
            const double l = exp(la + (lb - la) * (double)tid / 199.0);
            double gcv;
            sb_eval(ta, tb, tg, m, n, N, pure_ss, l, scr, tid, &gcv, nullptr, nullptr);
            fs[tid] = gcv; xs[tid] = l;
        }
        __syncthreads();
        if (tid == 0) {
            int cnt = 0;
            for (int i = 0; i < 200; ++i)
                if (!isnan(fs[i])) { gl[cnt] = xs[i]; gv[cnt] = fs[i]; ++cnt; }
            int il = 0;
            for (int i = 1; i < cnt; ++i) if (gv[i] < gv[il]) il = i;
            S.imisc[2] = cnt; S.imisc[3] = il;
            // mode of the refinement: 0 none (edge of the grid or empty), 1 fields' golden section, 2 converged
            int mode = 0;
            if (cnt == 0) { S.misc[6] = NAN; }
            else if (il == 0 || il + 1 == cnt) { S.misc[6] = gl[il]; }
            else mode = gcv_mode == MHS_GCV_FIELDS ? 1 : 2;
            S.imisc[4] = mode;
        }
        __syncthreads();
        const int mode = S.imisc[4], il = S.imisc[3];
        if (tid == 0) S.stamp[5] = S.stamp[6] = wall_clock64();
        if (mode != 0) {
            // golden section, 8 levels of its decision tree per round; every thread keeps (and replays) the walker's state
            GoldState st;
            double f1 = 0.0, f2 = 0.0, tol = 0.0;
            int iters = 0, maxit = 0;
            if (mode == 1) {
                const double ax = gl[il - 1], bx = gl[il], cx = gl[il + 1];
                const double r = 0.61803399, con = 1.0 - r;
                st.x0 = ax; st.x3 = cx;
                if (fabs(cx - bx) > fabs(bx - ax)) { st.x1 = bx; st.x2 = bx + con * (cx - bx); }
                else { st.x2 = bx; st.x1 = bx - con * (bx - ax); }
                tol = 0.01 * gv[il];
                maxit = 25;
            } else {
                const double lo = log(gl[il - 1]), hi = log(gl[il + 1]);
                const double r = 0.5 * (sqrt(5.0) - 1.0);
                st.x0 = lo; st.x3 = hi;
                st.x1 = hi - r * (hi - lo); st.x2 = lo + r * (hi - lo);
                maxit = 200;
            }
            __syncthreads();                  // gl / gv are read: fs / xs may be overwritten now
            if (tid < 2) {
                const double x = tid == 0 ? st.x1 : st.x2;
                double f;
                sb_eval(ta, tb, tg, m, n, N, pure_ss, mode == 1 ? x : exp(x), scr, tid, &f, nullptr, nullptr);
                fs[tid] = f;
            }
            __syncthreads();
            f1 = fs[0]; f2 = fs[1];
            __syncthreads();
            bool finished = false;
            while (!finished) {
                // every thread h = tid + 1 < 2^D is a node of the decision tree below the current state: the first
                // decision is known (f1, f2 are), the deeper ones are the bits of h under its leading one
                const int h = tid + 1;
                if (h < (1 << SB_TREE_DEPTH)) {
                    const int depth = 32 - __clz(h);               // 1 .. D
                    GoldState s = st;
                    double x = 0.0;
                    bool d = mode == 1 ? (f2 < f1) : (f1 < f2);
                    for (int l = 1; l <= depth; ++l) {
                        x = mode == 1 ? gold_step_fields(s, d) : gold_step_conv(s, d);
                        if (l < depth) d = (h >> (depth - 1 - l)) & 1;
                    }
                    double f;
                    sb_eval(ta, tb, tg, m, n, N, pure_ss, mode == 1 ? x : exp(x), scr, tid, &f, nullptr, nullptr);
                    fs[h] = f;
                }
                __syncthreads();
                // the walk (every thread replays it: no broadcast needed)
                int h2 = 1;
                for (int l = 1; l <= SB_TREE_DEPTH && !finished; ++l) {
                    if (mode == 1) {
                        const bool d = f2 < f1;
                        gold_step_fields(st, d);
                        if (d) { f1 = f2; f2 = fs[h2]; } else { f2 = f1; f1 = fs[h2]; }
                        ++iters;
                        if (fabs(f2 - f1) < tol || iters >= maxit) finished = true;
                        h2 = 2 * h2 + ((f2 < f1) ? 1 : 0);
                    } else {
                        const bool d = f1 < f2;
                        gold_step_conv(st, d);
                        if (d) { f2 = f1; f1 = fs[h2]; } else { f1 = f2; f2 = fs[h2]; }
                        ++iters;
                        if (fabs(st.x3 - st.x0) < 1e-13 || iters >= maxit) finished = true;
                        h2 = 2 * h2 + ((f1 < f2) ? 1 : 0);
                    }
                }
                __syncthreads();
            }
            lam = mode == 1 ? (f1 < f2 ? st.x1 : st.x2) : exp(0.5 * (st.x0 + st.x3));
        } else {
            lam = S.misc[6];
        }
        if (isnan(lam) || lam < 0) status = 1;
    }
    __syncthreads();

    // ---- solve in the tridiagonal basis
    if (tid == 0) {
        S.stamp[6] = wall_clock64();
        double gcv = NAN, tr = NAN;
        if (!status) sb_eval(ta, tb, tg, m, n, N, pure_ss, lam, scr, 0, &gcv, &tr, S.tq + 3);
        S.tq[0] = 0.0; S.tq[1] = 0.0; S.tq[2] = 0.0;
        S.misc[10] = gcv; S.misc[11] = tr;
    }
    if (tid >= n && tid < SB_NMAX) S.tq[tid] = 0.0;
    __syncthreads();

    // ---- back-transform, d, c: one wave, the vector in its registers (rows lane + 64 s)
    if (wave == 0) {
        constexpr int RS = SB_NMAX / 64;
        constexpr int PF = 4;               // reflectors in flight: a row of Vg is an L2 round trip away
        double ct[RS], vk[PF][RS];
#pragma unroll
        for (int s = 0; s < RS; ++s) ct[s] = lane + 64 * s < n ? S.tq[lane + 64 * s] : 0.0;
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int s = 0; s < RS; ++s) { const int k2 = n - 3 - p; vk[p][s] = (k2 >= 0 && lane + 64 * s < n) ? Vg[(size_t)k2 * ldv + lane + 64 * s] : 0.0; }
        for (int k0 = n - 3; k0 >= 0; k0 -= PF) {
            double vcur[PF][RS];
#pragma unroll
            for (int p = 0; p < PF; ++p)
#pragma unroll
                for (int s = 0; s < RS; ++s) vcur[p][s] = vk[p][s];
#pragma unroll
            for (int p = 0; p < PF; ++p)
#pragma unroll
                for (int s = 0; s < RS; ++s) { const int k2 = k0 - PF - p; vk[p][s] = (k2 >= 0 && lane + 64 * s < n) ? Vg[(size_t)k2 * ldv + lane + 64 * s] : 0.0; }
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int kk = k0 - p;
                if (kk < 0) break;
                if (kk == 2) {
                    // c2 is complete: d = R^-1 (w1 - A[0:3, 3:] c2)   (A's first three rows, by symmetry its first three columns)
                    double r3[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double sdot = 0.0;
#pragma unroll
                        for (int s = 0; s < RS; ++s) { const int r = lane + 64 * s; sdot += (r >= 3 && r < n) ? S.at[k][r] * ct[s] : 0.0; }
                        r3[k] = Jp->w1[k] - wave_sum(sdot);
                    }
                    const double d2 = r3[2] / Jp->R[8];
                    const double d1 = (r3[1] - Jp->R[1 + 3 * 2] * d2) / Jp->R[4];
                    const double d0 = (r3[0] - Jp->R[0 + 3 * 1] * d1 - Jp->R[0 + 3 * 2] * d2) / Jp->R[0];
                    if (lane == 0) { S.misc[12] = d0; S.misc[13] = d1; S.misc[14] = d2; }
                }
                double dot = 0.0;
#pragma unroll
                for (int s = 0; s < RS; ++s) dot += vcur[p][s] * ct[s];
                dot = wave_sum(dot);
                const double tk = S.ttau[kk];
#pragma unroll
                for (int s = 0; s < RS; ++s) ct[s] -= tk * dot * vcur[p][s];
            }
        }
        const double kcw = 0.5 / (8.0 * M_PI);
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            const int r = lane + 64 * s;
            if (r < n) {
                const double c = S.sw[r] * ct[s];
                c_out[r] = c;
                const int p = perm ? perm[r] : r;
                Knot kn; kn.u = S.u[r]; kn.v = S.v[r]; kn.cw = c * kcw; kn.pad = 0.0;
                knots_out[p] = kn;
            }
        }
        if (lane == 0) {
            res->lambda = lam; res->gcv = S.misc[10]; res->eff_df = S.misc[11];
            res->d[0] = S.misc[12]; res->d[1] = S.misc[13]; res->d[2] = S.misc[14];
            res->status = (double)status; res->pad = 0.0;
            const unsigned long long t7 = wall_clock64();
            for (int k = 0; k < 6; ++k) res->t_us[k] = 0.01 * (double)((k == 5 ? t7 : S.stamp[k + 2]) - S.stamp[k + 1]);
            res->t_us[6] = 0.01 * (double)(S.stamp[1] - S.stamp[0]); res->t_us[7] = 0.0;
#ifdef SB_PHASE_TRACE
            for (int k = 0; k < 7; ++k) res->t_us[k] = S.misc[20 + k];     // the step loop's phases as wave SB_TRACE_WAVE saw them
#endif
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(SB_THREADS) void tps_small_batch_kernel(const SmallJob *__restrict__ jobs, int njobs,
                                                                    const double *__restrict__ in, const int *__restrict__ perm,
                                                                    const double2 *__restrict__ gtab, double *__restrict__ scratch,
                                                                    int ldv, size_t scratch_per_block, double *__restrict__ c_out,
                                                                    Knot *__restrict__ knots_out, SmallResult *__restrict__ res) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    SbShared &S = *reinterpret_cast<SbShared *>(smem_raw);
    stage_log_table(S.tab, gtab);
    double *Vg = scratch + (size_t)blockIdx.x * scratch_per_block;
    double *scr = Vg + (size_t)ldv * ldv;
    const int tid = threadIdx.x;
    for (int job = blockIdx.x; job < njobs; job += gridDim.x) {
        const SmallJob *Jp = jobs + job;
        const int n = Jp->n;
        const double *jin = in + Jp->in_off;
        if (tid < SB_NMAX) {       // inputs to LDS
            const bool ok = tid < n;
            S.u[tid] = ok ? jin[tid] : 0.0;
            S.v[tid] = ok ? jin[n + tid] : 0.0;
            S.sw[tid] = ok ? jin[2 * n + tid] : 0.0;
            S.tg[tid] = ok ? jin[6 * n + tid] : 0.0;     // wv = Q'(W^1/2 ym); rows 3.. are g
            S.ta[tid] = 0.0; S.tb[tid] = 0.0; S.ttau[tid] = 0.0;
        }
        if (tid == 0) S.stamp[0] = wall_clock64();
        __syncthreads();
        switch ((n + 31) >> 5) {
            case 1: case 2: case 3: case 4: sb_reduce<4>(Jp, jin, Vg, ldv, S); break;
            case 5: sb_reduce<5>(Jp, jin, Vg, ldv, S); break;
            case 6: sb_reduce<6>(Jp, jin, Vg, ldv, S); break;
            case 7: sb_reduce<7>(Jp, jin, Vg, ldv, S); break;
            default: sb_reduce<8>(Jp, jin, Vg, ldv, S); break;
        }
        sb_finish(Jp, Jp->perm_off >= 0 ? perm + Jp->perm_off : nullptr, Vg, ldv, scr, S, c_out + Jp->c_off,
                  knots_out + Jp->knot_off, res + job);
    }
}

// ---------------------------------------------------------------------------------------------- host side
int small_batch_add(SmallBatch &B, const TpsPrep &P, double lambda, int gcv_mode, const int *perm) {
    const int n = (int)P.n;
    SmallJob J;
    memset(&J, 0, sizeof(J));
    J.n = n; J.N = (int)P.N; J.gcv_mode = gcv_mode;
    J.lambda = lambda; J.pure_ss = P.pure_ss;
    for (int k = 0; k < 3; ++k) { J.htau[k] = P.htau[k]; J.w1[k] = P.wv[k]; }
    for (int k = 0; k < 9; ++k) J.R[k] = P.R[k];
    J.in_off = (int64_t)B.in.size();
    B.in.insert(B.in.end(), P.uv.begin(), P.uv.end());          // u[n], v[n]
    B.in.insert(B.in.end(), P.sw.begin(), P.sw.end());
    for (int k = 0; k < 3; ++k) B.in.insert(B.in.end(), P.hv[k].begin(), P.hv[k].end());
    B.in.insert(B.in.end(), P.wv.begin(), P.wv.end());
    if (perm) { J.perm_off = (int64_t)B.perm.size(); B.perm.insert(B.perm.end(), perm, perm + n); }
    else J.perm_off = -1;
    J.c_off = B.c_total; B.c_total += n;
    J.knot_off = B.knot_total; B.knot_total += n;
    B.nmax = std::max(B.nmax, n);
    B.jobs.push_back(J);
    return B.count++;
}

int small_batch_launch(SmallBatch &B, FitLane &L, hipStream_t s, size_t extra_bytes, char **extra_dev) {
    if (B.count == 0 && extra_bytes == 0) return MHS_OK;
    const int nblk = std::max(1, std::min(B.count, std::max(1, ctx().n_cu)));
    const int ldv = (B.nmax + 31) & ~31;
    const size_t per_block = (size_t)ldv * ldv + (size_t)2 * ldv * SB_EV;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    const size_t o_jobs = off; off = up(off + sizeof(SmallJob) * B.jobs.size());
    const size_t o_in = off; off = up(off + sizeof(double) * B.in.size());
    const size_t o_perm = off; off = up(off + sizeof(int) * std::max<size_t>(B.perm.size(), 1));
    const size_t o_c = off; off = up(off + sizeof(double) * (size_t)B.c_total);
    const size_t o_k = off; off = up(off + sizeof(Knot) * (size_t)B.knot_total);
    const size_t o_res = off; off = up(off + sizeof(SmallResult) * (size_t)B.count);
    const size_t o_scr = off; off = up(off + sizeof(double) * per_block * (size_t)nblk);
    const size_t o_extra = off; off = up(off + extra_bytes);
    if (off > L.arena_cap) {
        if (L.arena) { (void)hipDeviceSynchronize(); (void)hipFree(L.arena); L.arena = nullptr; L.arena_cap = 0; }
        const size_t cap = off + off / 4;
        MHS_HIP(hipMalloc((void **)&L.arena, cap));
        L.arena_cap = cap;
    }
    char *base = L.arena;
    B.jobs_dev = (SmallJob *)(base + o_jobs);
    B.in_dev = (double *)(base + o_in);
    B.perm_dev = (int *)(base + o_perm);
    B.c_dev = (double *)(base + o_c);
    B.knots_dev = (Knot *)(base + o_k);
    B.res_dev = (SmallResult *)(base + o_res);
    double *scratch = (double *)(base + o_scr);
    if (extra_dev) *extra_dev = base + o_extra;
    if (B.count == 0) return MHS_OK;
    MHS_HIP(hipMemcpyAsync(B.jobs_dev, B.jobs.data(), sizeof(SmallJob) * B.jobs.size(), hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemcpyAsync(B.in_dev, B.in.data(), sizeof(double) * B.in.size(), hipMemcpyHostToDevice, s));
    if (!B.perm.empty()) MHS_HIP(hipMemcpyAsync(B.perm_dev, B.perm.data(), sizeof(int) * B.perm.size(), hipMemcpyHostToDevice, s));
    static_assert(sizeof(SbShared) <= 160 * 1024, "LDS");
    MHS_HIP(hipFuncSetAttribute((const void *)tps_small_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SbShared)));
    hipLaunchKernelGGL(tps_small_batch_kernel, dim3((unsigned)nblk), dim3(SB_THREADS), sizeof(SbShared), s, B.jobs_dev, B.count,
                       B.in_dev, B.perm_dev, ctx().log_tab, scratch, ldv, per_block, B.c_dev, B.knots_dev, B.res_dev);
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

int small_batch_results(const SmallBatch &B, hipStream_t s, std::vector<SmallResult> &res, std::vector<double> *c) {
    res.resize((size_t)B.count);
    if (B.count == 0) return MHS_OK;
    MHS_HIP(hipMemcpyAsync(res.data(), B.res_dev, sizeof(SmallResult) * (size_t)B.count, hipMemcpyDeviceToHost, s));
    if (c) {
        c->resize((size_t)B.c_total);
        MHS_HIP(hipMemcpyAsync(c->data(), B.c_dev, sizeof(double) * (size_t)B.c_total, hipMemcpyDeviceToHost, s));
    }
    MHS_HIP(hipStreamSynchronize(s));
    return MHS_OK;
}

}  // namespace mhs

using namespace mhs;


extern "C" int mhs_tps_fit_many(const double *const *xy, const double *const *y, const int64_t *N, int64_t count,
                                double lambda, int gcv_mode, mhs_tps **out, int *status) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(count >= 0 && (count == 0 || (xy && y && N && out)), "NULL argument");
    MHS_REQUIRE(std::isnan(lambda) || lambda >= 0, "lambda must be >= 0 or NaN");
    MHS_REQUIRE(gcv_mode == MHS_GCV_FIELDS || gcv_mode == MHS_GCV_CONVERGED, "bad gcv_mode");
    for (int64_t k = 0; k < count; ++k) {
        MHS_REQUIRE(xy[k] && y[k], "NULL argument");
        out[k] = nullptr;
        if (status) status[k] = MHS_OK;
    }
    if (count == 0) return MHS_OK;
    std::lock_guard<std::mutex> batch_lock(batch_mutex());
    FitLane *Lb = nullptr, *L0 = nullptr;
    if (int rc = batch_lane(&Lb)) return rc;
    if (int rc = fit_lane(0, &L0)) return rc;
    SmallBatch B;
    std::vector<TpsPrep> preps((size_t)count);
    std::vector<int> job_of((size_t)count, -1);
    std::vector<int64_t> big;
    for (int64_t k = 0; k < count; ++k) {
        int rc = (N[k] > 3 && N[k] < (1LL << 30)) ? tps_prepare(xy[k], y[k], N[k], preps[(size_t)k]) : MHS_ERR_INVALID;
        if (rc) { if (status) status[k] = rc; continue; }
        const TpsPrep &P = preps[(size_t)k];
        if (P.n >= SB_NMIN && P.n <= SB_NMAX) job_of[(size_t)k] = small_batch_add(B, P, lambda, gcv_mode, nullptr);
        else big.push_back(k);
    }
    if (int rc = small_batch_launch(B, *Lb, Lb->s)) return rc;
    // the fits the batch cannot hold run on lane 0 meanwhile
    for (int64_t k : big) {
        mhs_tps *t = nullptr;
        const int rc = tps_fit_lane(*L0, xy[k], y[k], N[k], lambda, gcv_mode, 0, &t);
        if (rc) { if (status) status[k] = rc; } else out[k] = t;
    }
    std::vector<SmallResult> res;
    std::vector<double> c;
    if (int rc = small_batch_results(B, Lb->s, res, &c)) return rc;
    if (getenv("MHS_TIMING") && !res.empty()) {
        double t[8] = {0};
        for (const SmallResult &r : res) for (int q = 0; q < 8; ++q) t[q] += r.t_us[q] / (double)res.size();
        fprintf(stderr, "[mhs_tps_fit_many] %d batched fits, mean us per fit: gram %.1f projection %.1f tridiagonalisation %.1f eigenvalues %.1f "
                "bracket+grid %.1f golden section %.1f solve+back-transform %.1f\n", (int)res.size(), t[6], t[0], t[1], t[2], t[3], t[4], t[5]);
    }
    for (int64_t k = 0; k < count; ++k) {
        const int j = job_of[(size_t)k];
        if (j < 0) continue;
        const SmallResult &r = res[(size_t)j];
        if (r.status != 0.0) { if (status) status[k] = MHS_ERR_NUMERIC; continue; }
        const TpsPrep &P = preps[(size_t)k];
        mhs_tps *t = new mhs_tps();
        t->n = P.n;
        t->lambda = r.lambda; t->eff_df = r.eff_df; t->gcv = r.gcv;
        memcpy(t->center, P.center, sizeof(t->center));
        memcpy(t->scale, P.scale, sizeof(t->scale));
        memcpy(t->d, r.d, sizeof(t->d));
        t->c.assign(c.begin() + B.jobs[(size_t)j].c_off, c.begin() + B.jobs[(size_t)j].c_off + P.n);
        t->knots_uv = P.uv;
        // the knot records are already on the device (natural order): device-to-device into the handle's own block
        t->knots_dev = (Knot *)pool_alloc(sizeof(Knot) * (size_t)P.n);
        if (!t->knots_dev) { tps_free_quiet(t); return MHS_ERR_ALLOC; }
        MHS_HIP(hipMemcpyAsync(t->knots_dev, B.knots_dev + B.jobs[(size_t)j].knot_off, sizeof(Knot) * (size_t)P.n,
                               hipMemcpyDeviceToDevice, Lb->s));
        out[k] = t;
    }
    MHS_HIP(hipStreamSynchronize(Lb->s));
    return MHS_OK;
}
