// GCV route of the spline fit, round 4: fields::Tps(x, Y) (V73:722, V73:751) needs lambda = arg min GCV over B = Q2'KQ2.
//
// Round 1-3 reduced B to a band of width 8 with 8-column Householder panels -- n/8 panels, each a chain of three globally
// dependent launches: 624 links at n = 5 000, 52 of the fit's 62 ms, 3 % of the FP64 peak (tps_fit.hip, kept as the
// fallback).  This file cuts the DEPTH of that chain:
//
//   stage 1   B = Q Bb Q', Bb of bandwidth 32, with 32-COLUMN panels: n/32 links.  A panel P (t x 32) is factorised by
//             CholeskyQR2 -- two passes of {Gram matrix on MFMA, 32 x 32 Cholesky, row-wise triangular solve}, every block
//             of rows working alone between two launches -- and the compact-WY form H = I - V T V' the two-sided update
//             needs is RECONSTRUCTED from the orthonormal factor: H [I; 0] = Q D (D = diag(+-1)) <=> L U = [I; 0] - Q D with
//             the signs chosen so that every pivot is >= 1 (Ballard et al., "Reconstructing Householder vectors from TSQR",
//             2015); V = L, T = U L1^-T.  Only the 32 x 32 top block is sequential, every block of rows redoes it.
//             Then Y = A22 V (MFMA, split-K), W = Y T - 1/2 V (T'(V'Y)T), A22 -= V W' + W V' (MFMA rank-64, the first block
//             column ahead of the rest so that the next panel starts behind it).
//   GCV       on the band itself, on the GPU: for one lambda ONE forward sweep of an LDL' recurrence over T + lambda I that
//             carries its own derivative with respect to lambda gives the inertia, tr (T + lambda I)^-1 = sum d'_j / d_j and
//             g'(T + lambda I)^-2 g = -d/dlambda sum y_j^2 / d_j: no back substitution, no stored factor.  A sweep is a
//             chain of m column steps; it runs from BOTH ends to the middle (twisted factorisation: two blocks per lambda,
//             a third kernel joins them), and a search round evaluates hundreds of lambdas side by side -- the search is
//             organised in few, wide rounds (1 023-point multi-section for the two extreme eigenvalues, the 40-point
//             bracket, the 200-point grid, a speculatively expanded golden section).
//   solve     (Bb + lambda I) q = g by the same sweep with the factor stored, c2 = Q q one launch per panel.
//
// tests/test_band32_gpu.py checks every piece against a numpy restatement kept with the test infrastructure.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include "common.h"
#include "devmath.h"
#include "tps_band32.h"

namespace mhs {

constexpr int NB = B32_NB;                 // 32
constexpr int MAXPART_H = B32_MAXPART;
constexpr int CHR = 256;                   // rows per chunk: one row per thread
constexpr int PT_S = CHR + 2;              // LDS stride of a transposed chunk Pt[a][i]: == 2 (mod 32), so the 16 x 4 MFMA
                                           // operand reads Pt[a0 + l15][k0 + l4] touch 32 distinct bank pairs per half-wave
typedef double d4v __attribute__((ext_vector_type(4)));

// -DB32_PROF: block 0 of every stage-1 kernel adds the shader-clock cycles between its phase marks to a table the host
// prints after the reduction (a development build; the shipped library carries none of this)
#ifdef B32_PROF
__device__ unsigned long long b32_prof_tab[8][16];
#define B32_PROF_BEGIN unsigned long long prof_t = __builtin_readcyclecounter();
#define B32_MARK(K, P)                                                                              \
    do {                                                                                            \
        const unsigned long long prof_n = __builtin_readcyclecounter();                             \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) b32_prof_tab[K][P] += prof_n - prof_t; \
        prof_t = prof_n;                                                                            \
    } while (0)
#else
#define B32_PROF_BEGIN
#define B32_MARK(K, P)
#endif

// ------------------------------------------------------------------------------------------------ small helpers --
// What a previous launch wrote sits in another XCD's L2 or in HBM: a load of it costs ~2 us, and a loop of "load, add" over
// the partial results of 10-20 blocks is a chain of them (the first version of this file spent 40 us of every small kernel
// that way).  Every kernel below therefore ISSUES all of its global loads before it waits for any of them.
__device__ __forceinline__ double frcp(double d) {      // 1 / d to ~1 ulp: v_rcp_f64 and two Newton steps (no div_scale / div_fixup)
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
// acc += Pt[ta-tile]' Pt[tb-tile] over the chunk's 256 rows: D[a][b] = sum_i Pa[i][a] Pb[i][b]
__device__ __forceinline__ void gram_chunk(const double *__restrict__ Pa, const double *__restrict__ Pb, d4v &acc, int ta, int tb) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const double *pa = Pa + (ta * 16 + l15) * PT_S + l4, *pb = Pb + (tb * 16 + l15) * PT_S + l4;
#pragma unroll 8
    for (int k0 = 0; k0 < CHR; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], acc, 0, 0, 0);
}
__device__ __forceinline__ void store_tile32(double *__restrict__ out, const d4v &acc, int ta, int tb) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(ta * 16 + l4 + 4 * r) * NB + tb * 16 + l15] = acc[r];
}
// one 16 x 16 tile (wave w: rows 16 (w >> 1) .., columns 16 (w & 1) ..) of the 32 x 32 product X Y (TA: X'Y) of two LDS
// matrices, on MFMA; entry r of the result is row 16 (w >> 1) + (lane >> 4) + 4 r, column 16 (w & 1) + (lane & 15)
template <bool TA>
__device__ __forceinline__ d4v mm32_tile(const double (*X)[NB + 1], const double (*Y)[NB + 1]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4, ta = wave >> 1, tb = wave & 1;
    d4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k0 = 0; k0 < NB; k0 += 4) {
        const double a = TA ? X[k0 + l4][ta * 16 + l15] : X[ta * 16 + l15][k0 + l4];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Y[k0 + l4][tb * 16 + l15], acc, 0, 0, 0);
    }
    return acc;
}
// the sum of up to MAXPART 32 x 32 partial matrices (added in block order), entry e = tid + 256 q: loads first, adds after
struct Parts4 { double v[4][MAXPART_H]; };
__device__ __forceinline__ void parts_issue(Parts4 &P, const double *__restrict__ part, int nparts) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < MAXPART_H; ++p) {
            const double x = part[(size_t)(p < nparts ? p : 0) * NB * NB + threadIdx.x + 256 * q];
            P.v[q][p] = p < nparts ? x : 0.0;
        }
}
__device__ __forceinline__ void parts_add(const Parts4 &P, double (*G)[NB + 1]) {      // a second batch onto the first one's sums
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < MAXPART_H; ++p) s += P.v[q][p];
        const int e = threadIdx.x + 256 * q;
        G[e >> 5][e & 31] += s;
    }
}
__device__ __forceinline__ void parts_sum(const Parts4 &P, double (*G)[NB + 1]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < MAXPART_H; ++p) s += P.v[q][p];
        const int e = threadIdx.x + 256 * q;
        G[e >> 5][e & 31] = s;
    }
}

// G (lower triangle, LDS [32][33]) -> its Cholesky factor L in place (G = L L'), 256 threads.  One barrier per PAIR of
// columns: a step reads columns k and k + 1 as they stand, redoes column k's effect on column k + 1 in registers and
// updates the columns right of them with the unscaled entries, G[i][j] -= G[i][k] G[j][k] / G[k][k] and then the same with
// k + 1 -- entry by entry the very operations of the one-column loop (a barrier-separated LDS round trip costs ~470 ns, and
// 31 of them were 14.6 us of the first pass's 32; the elimination in one wave's registers, broadcasting through v_readlane
// or through the LDS, was tried and is slower still: the compiler spills what it hoists).  The columns are scaled by
// 1 / sqrt(pivot) at the end.  Returns false (in every thread) if a pivot was not positive -- the panel was numerically
// rank deficient.
__device__ __forceinline__ bool chol32_lds(double (*G)[NB + 1], double *sd /* [32] */) {
    __shared__ double Gf[NB][NB + 1];      // the odd columns 1 .. 29 as their pair's step leaves them
    const int i = threadIdx.x >> 3, jg = threadIdx.x & 7;
    bool ok = true;
    for (int k = 0; k < NB - 2; k += 2) {
        __syncthreads();
        // straight-line: every thread rewrites its four entries (unchanged where the step does not reach), so that the
        // step is one batch of LDS reads and one of writes instead of four predicated read-modify-write round trips
        double d = G[k][k];
        const double e = G[k + 1][k], d1raw = G[k + 1][k + 1], gi0 = G[i][k], gi1 = G[i][k + 1];
        double gj0[4], gj1[4], old[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { gj0[q] = G[jg + 8 * q][k]; gj1[q] = G[jg + 8 * q][k + 1]; old[q] = G[i][jg + 8 * q]; }
        if (!(d > 0.0) || !(d < 1e300)) { ok = false; d = 1.0; }
        const double r0 = frcp(d);
        const double ek = e * r0;                         // what row k + 1 subtracts from the rows below it in column k's step
        double d1 = d1raw - ek * e;
        if (!(d1 > 0.0) || !(d1 < 1e300)) { ok = false; d1 = 1.0; }
        const double r1 = frcp(d1);
        const double gik = gi0 * r0;
        const double gik1 = (i > k ? gi1 - gik * e : gi1) * r1;      // column k + 1 of row i after column k's step
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = jg + 8 * q;
            const double v0 = (i > k && j > k && j <= i) ? old[q] - gik * gj0[q] : old[q];
            const double gj1u = j > k ? gj1[q] - (gj0[q] * r0) * e : gj1[q];      // G[j][k + 1] after column k's step
            // column k + 1 is still being READ by other threads in this step: its updated entries (final: no later step
            // touches them) go to the second copy
            double *dst = j == k + 1 ? &Gf[i][j] : &G[i][j];
            *dst = (i > k + 1 && j > k + 1 && j <= i) ? v0 - gik1 * gj1u : v0;
        }
    }
    {   // column 30 (the last one with a row below it)
        const int k = NB - 2;
        __syncthreads();
        double d = G[k][k];
        if (!(d > 0.0) || !(d < 1e300)) { ok = false; d = 1.0; }
        const double gik = G[i][k] * frcp(d);
        double gjk[4], old[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { gjk[q] = G[jg + 8 * q][k]; old[q] = G[i][jg + 8 * q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = jg + 8 * q;
            G[i][j] = (i > k && j > k && j <= i) ? old[q] - gik * gjk[q] : old[q];
        }
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        const int t = threadIdx.x;
        double d = ((t & 1) && t < NB - 2) ? Gf[t][t] : G[t][t];
        if (!(d > 0.0) || !(d < 1e300)) { ok = false; d = 1.0; }
        sd[t] = sqrt(d);
    }
    ok = __syncthreads_and(ok);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = jg + 8 * q;
        if (j < i) G[i][j] = (((j & 1) && j < NB - 2) ? Gf[i][j] : G[i][j]) / sd[j];
        else if (j == i) G[i][j] = sd[j];
    }
    __syncthreads();
    return ok;
}

// The solve with the row in REGISTERS and the factor read from LDS as broadcast 16-byte reads -- 264 ds_read_b128 and 496
// v_fma_f64 per row.  Two things make this the fast form (measured, round 4): (1) the compiler, left alone, hoists all
// 528 reads out of the row loop and spills ~700 registers -- so the LDS offset is laundered through an empty asm that
// also consumes the row just finished, every second row, which pins each pair of rows' reads behind the previous pair's
// arithmetic; (2) the factor in the SCALAR cache instead (s_load from a per-block slot of global memory, SGPR operands)
// was tried and costs ~1 us per batch of loads, 30 batches a row: 35 us per 256 rows against ~2 here.
// Lp: [32][32] row-major (Lp[j][k] = L[j][k], k < j), then 32 reciprocals of the diagonal; 16-byte aligned.
constexpr int TRS_SIZE = NB * NB + NB;
typedef double d2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) d2v *lds_c2ptr;
typedef const __attribute__((address_space(3))) double *lds_cptr;
__device__ __forceinline__ void trs_pack(double *__restrict__ Lp, const double (*L)[NB + 1], const double *rinv) {
    for (int e = threadIdx.x; e < NB * NB; e += 256) Lp[e] = L[e >> 5][e & 31];
    if (threadIdx.x < NB) Lp[NB * NB + threadIdx.x] = rinv[threadIdx.x];
    __syncthreads();
}
__device__ __forceinline__ void row_trsm_reg(double (&x)[NB], const double *Lp) {
    unsigned off = (unsigned)(uintptr_t)(lds_cptr)Lp;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if ((j & 1) == 0) asm volatile("" : "+v"(off) : "v"(x[j > 0 ? j - 1 : 0]) : "memory");
        lds_c2ptr row = (lds_c2ptr)(uintptr_t)(off + (unsigned)(j * NB * sizeof(double)));
        double s0 = x[j], s1 = 0.0;
#pragma unroll
        for (int k = 0; k < j; k += 2) {
            const d2v l = row[k >> 1];
            s0 = fma(-x[k], l.x, s0);
            if (k + 1 < j) s1 = fma(-x[k + 1], l.y, s1);
        }
        x[j] = (s0 + s1) * ((lds_cptr)(uintptr_t)off)[NB * NB + j];
    }
}

// sum of x[a] over the block's 256 threads for a = 0 .. 31, in a fixed order: thread (a, seg) adds 32 consecutive rows,
// thread a adds the 8 segments.  Result in threads 0 .. 31 (value a).  Pt: >= 32 * PT_S doubles, red: [8][32].
__device__ __forceinline__ double block_colsum32(const double (&x)[NB], double *Pt, double (*red)[NB]) {
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NB; ++a) Pt[a * PT_S + threadIdx.x] = x[a];
    __syncthreads();
    {
        const int a = threadIdx.x & 31, seg = threadIdx.x >> 5;
        const double *src = Pt + a * PT_S + seg * 32;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) s += src[q];
        red[seg][a] = s;
    }
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < NB) {
#pragma unroll
        for (int seg = 0; seg < 8; ++seg) tot += red[seg][threadIdx.x];
    }
    return tot;
}

// the V row of B-row il (0-based inside the panel): rows 32.. sit in place below the band, the top block V1 (a full 32 x 32
// matrix in the basis-kernel form, where the panel keeps the band entries) in the panel's record
__device__ __forceinline__ void load_v_row(double (&v)[NB], const double *__restrict__ Ppan, int64_t ld, int il, bool ok,
                                           const double *__restrict__ V1) {
    const bool top = il < NB;
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        const double x = top ? V1[(ok ? il : 0) * NB + a] : Ppan[(int64_t)a * ld + (ok ? il : 0)];
        v[a] = ok ? x : 0.0;
    }
}
// up to 16 x 32 partial sums of 32 values (added in block order) by threads 0 .. 31: loads first
__device__ __forceinline__ double sum_parts_32(const double *__restrict__ part, int nparts) {
    double v[MAXPART_H];
    const int a = threadIdx.x & 31;
#pragma unroll
    for (int p = 0; p < MAXPART_H; ++p) { const double x = part[(size_t)(p < nparts ? p : 0) * NB + a]; v[p] = p < nparts ? x : 0.0; }
    double s = 0.0;
#pragma unroll
    for (int p = 0; p < MAXPART_H; ++p) s += v[p];
    if (nparts > MAXPART_H) {      // the second batch (panels of more than 4 096 rows)
#pragma unroll
        for (int p = 0; p < MAXPART_H; ++p) {
            const double x = part[(size_t)(MAXPART_H + p < nparts ? MAXPART_H + p : 0) * NB + a];
            v[p] = MAXPART_H + p < nparts ? x : 0.0;
        }
        double s2 = 0.0;
#pragma unroll
        for (int p = 0; p < MAXPART_H; ++p) s2 += v[p];
        s += s2;
    }
    return s;
}

// ---------------------------------------------------------------------------------------------- stage 1 kernels --
// K0: partial Gram matrices of the panel P = A[r0 :, c0 : c0 + 32] (t rows), one per block of cpb chunks
__global__ __launch_bounds__(256) void b32_gram_kernel(const double *__restrict__ A, int64_t ld, int c0, int r0, int t, int cpb,
                                                       double *__restrict__ Gpart) {
    __shared__ double Pt[NB * PT_S];
    const int wave = threadIdx.x >> 6, ta = wave >> 1, tb = wave & 1;
    d4v acc = {0.0, 0.0, 0.0, 0.0};
    B32_PROF_BEGIN
    for (int ch = 0; ch < cpb; ++ch) {
        const int i = (blockIdx.x * cpb + ch) * CHR + threadIdx.x;
        const bool ok = i < t;
        const double *src = A + (int64_t)c0 * ld + r0 + (ok ? i : 0);
#pragma unroll
        for (int a = 0; a < NB; ++a) { const double x = src[(int64_t)a * ld]; Pt[a * PT_S + threadIdx.x] = ok ? x : 0.0; }
        __syncthreads();
        B32_MARK(0, 0);
        gram_chunk(Pt, Pt, acc, ta, tb);
        __syncthreads();
        B32_MARK(0, 1);
    }
    store_tile32(Gpart + (size_t)blockIdx.x * NB * NB, acc, ta, tb);
    B32_MARK(0, 2);
}

// K1: R1 = chol(P'P), Q1 = P R1^-1 (in place), partial Gram matrices of Q1
__global__ __launch_bounds__(256) void b32_cholqr1_kernel(double *__restrict__ A, int64_t ld, int c0, int r0, int t, int cpb,
                                                          const double *__restrict__ Gin, int nparts, double *__restrict__ Gout,
                                                          double *__restrict__ R1out, double *__restrict__ Qtop,
                                                          int *__restrict__ flags) {
    __shared__ double Pt[NB * PT_S];
    __shared__ double G[NB][NB + 1];
    __shared__ __attribute__((aligned(16))) double Lp[TRS_SIZE];
    __shared__ double sd[NB], rinv[NB];
    double x[NB];      // the first chunk's row travels while the Gram matrix is summed and factorised
    B32_PROF_BEGIN
    {
        Parts4 P;
        parts_issue(P, Gin, nparts);
        const int i = blockIdx.x * cpb * CHR + threadIdx.x;
        const double *row = A + (int64_t)c0 * ld + r0 + (i < t ? i : 0);
#pragma unroll
        for (int a = 0; a < NB; ++a) x[a] = row[(int64_t)a * ld];
        parts_sum(P, G);
        if (nparts > MAXPART_H) { parts_issue(P, Gin + (size_t)MAXPART_H * NB * NB, nparts - MAXPART_H); parts_add(P, G); }
    }
    B32_MARK(1, 0);
    const bool ok_chol = chol32_lds(G, sd);
    B32_MARK(1, 1);
    if (threadIdx.x < NB) rinv[threadIdx.x] = 1.0 / sd[threadIdx.x];
    if (!ok_chol && threadIdx.x == 0) atomicOr(flags, 1);
    __syncthreads();
    trs_pack(Lp, G, rinv);
    if (blockIdx.x == 0) {      // R1[k][j] = L[j][k], dense row-major
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = threadIdx.x + 256 * q, k = e >> 5, j = e & 31;
            R1out[e] = j >= k ? G[j][k] : 0.0;
        }
    }
    const int wave = threadIdx.x >> 6, ta = wave >> 1, tb = wave & 1;
    d4v acc = {0.0, 0.0, 0.0, 0.0};
    B32_MARK(1, 2);
    for (int ch = 0; ch < cpb; ++ch) {
        const int i = (blockIdx.x * cpb + ch) * CHR + threadIdx.x;
        const bool ok = i < t;
        double *row = A + (int64_t)c0 * ld + r0 + (ok ? i : 0);
        if (ch > 0) {
#pragma unroll
            for (int a = 0; a < NB; ++a) x[a] = row[(int64_t)a * ld];
        }
        if (!ok) {
#pragma unroll
            for (int a = 0; a < NB; ++a) x[a] = 0.0;
        }
        row_trsm_reg(x, Lp);
        B32_MARK(1, 3);
#pragma unroll
        for (int a = 0; a < NB; ++a) { if (ok) row[(int64_t)a * ld] = x[a]; Pt[a * PT_S + threadIdx.x] = x[a]; }
        if (i < NB) {      // the top block once more, aside: K2 overwrites it in place while other blocks still need it
#pragma unroll
            for (int a = 0; a < NB; ++a) Qtop[i * NB + a] = x[a];
        }
        __syncthreads();
        B32_MARK(1, 4);
        gram_chunk(Pt, Pt, acc, ta, tb);
        __syncthreads();
        B32_MARK(1, 5);
    }
    store_tile32(Gout + (size_t)blockIdx.x * NB * NB, acc, ta, tb);
    B32_MARK(1, 6);
}

// K2: the second Cholesky-QR pass and the Householder reconstruction WITHOUT a serial step (round 4, second form; the first
// form -- chol(Q1'Q1), a modified LU of [I; 0] - Q D, two triangular solves per row -- spent 45 us of a 55 us kernel in its
// 95 barrier-separated column steps):
//   * Q1 is orthonormal up to E = Q1'Q1 - I = O(eps cond(P)^2) <= ~1e-8, so chol(I + E) is its own expansion: with Phi(X) =
//     strict upper + half the diagonal, U = Phi(E) - Phi(Phi(E)'Phi(E)), R2 = I + U and R2^-1 = I - U + U^2, both to O(E^3);
//     beyond max |E| = 1e-5 the panel is flagged and the fit falls back.  Q = Q1 (I - U + U^2) is one MFMA product.
//   * the reconstruction in Yamamoto's basis-kernel form: with D_k = -sign(Q_kk) and Q^ = Q D, H = I - V T V', V = [I; 0] -
//     Q^, T = (I - Q^_1)^-T is orthogonal and H [I; 0] = Q^ -- no triangular structure, so V is Q with its signs flipped
//     (plus the identity on the top block V1, which goes to the panel's record because the panel keeps the band entries
//     there) and T is a 32 x 32 inverse that nothing before b32_w_kernel needs: b32_tfin forms it beside the symmetric
//     product.  cond(V1) stays below ~10 even for 64-row panels (the signs put Q^_1's diagonal at -|.|).
constexpr int AUX_U = 0, AUX_D = NB * NB, AUX_SIZE = NB * NB + NB;
constexpr int PREC = B32_PANEL_REC;      // doubles per panel record: T (32 x 32, row-major), then V1
__global__ __launch_bounds__(256) void b32_cholqr2_kernel(double *__restrict__ A, int64_t ld, int c0, int r0, int t, int cpb,
                                                          const double *__restrict__ Gin, int nparts, const double *__restrict__ Qtop,
                                                          double *__restrict__ Zc, int64_t vs, double *__restrict__ Vr,
                                                          double *__restrict__ aux, double *__restrict__ rec,
                                                          const double *__restrict__ g, double *__restrict__ sgpart,
                                                          int *__restrict__ flags) {
    __shared__ double Pt[NB * PT_S];
    __shared__ double Em[NB][NB + 1];           // E, then I - U + U^2 (the MFMA's B operand)
    __shared__ double U1[NB][NB + 1], Um[NB][NB + 1], Qt[NB][NB + 1];
    __shared__ double Dv[NB], red[8][NB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    double xrow[NB];                            // the first chunk's row of Q1
    B32_PROF_BEGIN
    {
        Parts4 P;
        parts_issue(P, Gin, nparts);
        double qt4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) qt4[q] = Qtop[threadIdx.x + 256 * q];
        const int i = blockIdx.x * cpb * CHR + threadIdx.x;
        const double *row = A + (int64_t)c0 * ld + r0 + (i < t ? i : 0);
#pragma unroll
        for (int a = 0; a < NB; ++a) xrow[a] = row[(int64_t)a * ld];
        parts_sum(P, Em);
        if (nparts > MAXPART_H) { parts_issue(P, Gin + (size_t)MAXPART_H * NB * NB, nparts - MAXPART_H); parts_add(P, Em); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = threadIdx.x + 256 * q; Qt[e >> 5][e & 31] = qt4[q]; }
    }
    __syncthreads();
    B32_MARK(2, 0);
    bool bad = false;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {      // E and U1 = Phi(E)
        const int a = e >> 5, b = e & 31;
        const double ev = 0.5 * (Em[a][b] + Em[b][a]) - (a == b ? 1.0 : 0.0);
        bad |= !(fabs(ev) <= 1e-5);
        U1[a][b] = a < b ? ev : (a == b ? 0.5 * ev : 0.0);
    }
    bad = __syncthreads_or(bad);
    if (bad && threadIdx.x == 0) atomicOr(flags, 2);
    {   // U = U1 - Phi(U1'U1), then I - U + U^2: two 32 x 32 products on MFMA (a tile per wave)
        const int ta = wave >> 1, tb = wave & 1;
        const d4v x = mm32_tile<true>(U1, U1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = ta * 16 + l4 + 4 * r, b = tb * 16 + l15;
            Um[a][b] = U1[a][b] - (a < b ? x[r] : (a == b ? 0.5 * x[r] : 0.0));
        }
        __syncthreads();
        const d4v y = mm32_tile<false>(Um, Um);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = ta * 16 + l4 + 4 * r, b = tb * 16 + l15;
            Em[a][b] = ((a == b ? 1.0 : 0.0) - Um[a][b]) + y[r];
        }
    }
    __syncthreads();
    if (threadIdx.x < NB) {      // the signs: D_k = -sign((Q1_top (I - U + U^2))_kk)
        const int k = threadIdx.x;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int j = 0; j < NB; j += 2) { s0 = fma(Qt[k][j], Em[j][k], s0); s1 = fma(Qt[k][j + 1], Em[j + 1][k], s1); }
        Dv[k] = (s0 + s1) >= 0.0 ? -1.0 : 1.0;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int e = threadIdx.x; e < NB * NB; e += 256) aux[AUX_U + e] = Um[e >> 5][e & 31];
        if (threadIdx.x < NB) aux[AUX_D + threadIdx.x] = Dv[threadIdx.x];
    }
    double sgacc = 0.0;
    B32_MARK(2, 1);
    for (int ch = 0; ch < cpb; ++ch) {
        const int chunk = blockIdx.x * cpb + ch;
        const int i = chunk * CHR + threadIdx.x;
        const bool ok = i < t;
        double *row = A + (int64_t)c0 * ld + r0 + (ok ? i : 0);
        __syncthreads();
        if (ch > 0) {
#pragma unroll
            for (int a = 0; a < NB; ++a) xrow[a] = row[(int64_t)a * ld];
        }
        const double gi = ok ? g[i] : 0.0;
#pragma unroll
        for (int a = 0; a < NB; ++a) Pt[a * PT_S + threadIdx.x] = ok ? xrow[a] : 0.0;
        __syncthreads();
        {      // Q = Q1 (I - U + U^2): wave w's 64 rows, in place (a wave reads only its own rows)
            d4v acc[4][2];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[rt][nt] = (d4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k0 = 0; k0 < NB; k0 += 4) {
                const double b0 = Em[k0 + l4][l15], b1 = Em[k0 + l4][16 + l15];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const double av = Pt[(k0 + l4) * PT_S + wave * 64 + rt * 16 + l15];
                    acc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, acc[rt][0], 0, 0, 0);
                    acc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, acc[rt][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = wave * 64 + rt * 16 + l4 + 4 * r;
                    Pt[l15 * PT_S + rw] = acc[rt][0][r];
                    Pt[(16 + l15) * PT_S + rw] = acc[rt][1][r];
                }
        }
        __syncthreads();
        B32_MARK(2, 3);
        double x[NB];      // the row of V = [I; 0] - Q D
#pragma unroll
        for (int a = 0; a < NB; ++a) x[a] = (a == i ? 1.0 : 0.0) - Dv[a] * Pt[a * PT_S + threadIdx.x];
        if (!ok) {
#pragma unroll
            for (int a = 0; a < NB; ++a) x[a] = 0.0;
        }
        if (ok && i >= NB) {
#pragma unroll
            for (int a = 0; a < NB; ++a) row[(int64_t)a * ld] = x[a];
        }
        if (i < NB) {      // the top block V1 goes to the panel's record
#pragma unroll
            for (int a = 0; a < NB; ++a) rec[NB * NB + i * NB + a] = x[a];
        }
        double prod[NB];
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            if (ok) { Zc[(int64_t)a * vs + i] = x[a]; }
            prod[a] = x[a] * gi;
        }
        if (i < t + 64) {      // row-major copy, and zero rows past the end for the symmetric product's last k-steps
            double2 *dst = (double2 *)(Vr + (int64_t)i * NB);
#pragma unroll
            for (int a = 0; a < NB; a += 2) dst[a >> 1] = ok ? make_double2(x[a], x[a + 1]) : make_double2(0.0, 0.0);
        }
        B32_MARK(2, 4);
        const double tot = block_colsum32(prod, Pt, red);
        sgacc += tot;
        B32_MARK(2, 5);
    }
    // rows t .. t + 63 of Vr that no chunk of this grid reaches
    {
        const int covered = (int)gridDim.x * cpb * CHR;
        if (blockIdx.x == gridDim.x - 1 && covered < t + 64) {
            for (int i = covered + threadIdx.x; i < t + 64; i += 256)
                for (int a = 0; a < NB; ++a) Vr[(int64_t)i * NB + a] = 0.0;
        }
    }
    if (threadIdx.x < NB) sgpart[(size_t)blockIdx.x * NB + threadIdx.x] = sgacc;
    B32_MARK(2, 6);
}

// T = V1^-T (Gauss-Jordan between two LDS copies: one barrier per column) and the band entries R~ = D (I + U) R1 of a
// Cholesky-QR panel.  One block, launched as an extra block of the symmetric product (nothing before b32_w_kernel needs T).
__device__ __forceinline__ void b32_tfin(double *__restrict__ A, int64_t ld, int c0, int r0, const double *__restrict__ aux,
                                         const double *__restrict__ R1, double *__restrict__ rec, int *__restrict__ flags,
                                         double *smem) {
    double(*Us)[NB + 1] = (double(*)[NB + 1])smem;
    double(*R1s)[NB + 1] = Us + NB;
    double(*Ma)[NB + 1] = R1s + NB;
    double(*Mb)[NB + 1] = Ma + NB;
    double *Dv = (double *)(Mb + NB);
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e >> 5, c = e & 31;
        Us[r][c] = aux[AUX_U + e]; R1s[r][c] = R1[e]; Ma[r][c] = rec[NB * NB + e];
    }
    if (threadIdx.x < NB) Dv[threadIdx.x] = aux[AUX_D + threadIdx.x];
    __syncthreads();
    const int i8 = threadIdx.x >> 3, cg = threadIdx.x & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // R~[r][a] = D_r sum_{k = r .. a} (I + U)[r][k] R1[k][a]
        const int r = i8, a = cg + 8 * q;
        if (a >= r) {
            double s = R1s[r][a];
            for (int k = r; k <= a; ++k) s = fma(Us[r][k], R1s[k][a], s);
            A[(int64_t)(c0 + a) * ld + r0 + r] = Dv[r] * s;
        }
    }
    bool bad = false;
    for (int k = 0; k < NB; ++k) {      // in-place inverse, reading one copy and writing the other
        double(*Mi)[NB + 1] = (k & 1) ? Mb : Ma;
        double(*Mo)[NB + 1] = (k & 1) ? Ma : Mb;
        const double p = Mi[k][k], f = Mi[i8][k];
        bad |= !(fabs(p) > 1e-6);
        const double rp = frcp(p);
        double mk[4], mi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { mk[q] = Mi[k][cg + 8 * q]; mi[q] = Mi[i8][cg + 8 * q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = cg + 8 * q;
            const double rowk = c == k ? rp : mk[q] * rp;
            Mo[i8][c] = i8 == k ? rowk : (c == k ? -f * rp : mi[q] - f * rowk);
        }
        __syncthreads();
    }
    if (bad && threadIdx.x == 0) atomicOr(flags, 4);
    for (int e = threadIdx.x; e < NB * NB; e += 256) rec[e] = Ma[e & 31][e >> 5];      // 32 steps: the inverse is back in Ma; T = its transpose
}
constexpr int TFIN_LDS = 4 * NB * (NB + 1) + NB;      // doubles

// The SHORT last panel (t < 64 rows, possibly fewer rows than columns): classical Householder QR in one block -- the
// Cholesky route needs t >= 32 and full column rank.  Same outputs as K1 + K2 + b32_tfin (one partial of V'g).
__global__ __launch_bounds__(256) void b32_panel_small_kernel(double *__restrict__ A, int64_t ld, int c0, int r0, int t,
                                                              double *__restrict__ Zc, int64_t vs, double *__restrict__ Vr,
                                                              double *__restrict__ rec, const double *__restrict__ g,
                                                              double *__restrict__ sgpart) {
    __shared__ double P[64][NB + 1], V[64][NB + 1], Gs[NB][NB + 1], Ts[NB][NB + 1];
    __shared__ double wpart[8][NB], taus[NB], ssh[2], red[8][NB];
    __shared__ double Pt[NB * PT_S];
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * NB; e += 256) {
        const int i = e & 63, a = e >> 6;
        P[i][a] = i < t ? A[(int64_t)(c0 + a) * ld + r0 + i] : 0.0;
        V[i][a] = 0.0;
    }
    __syncthreads();
    const int nref = min(NB, t - 1);
    for (int j = 0; j < NB; ++j) {
        if (j < nref) {
            if (tid < 64) {
                double s = tid > j && tid < t ? P[tid][j] * P[tid][j] : 0.0;
                s = wave_sum(s);
                if (tid == 0) ssh[0] = s;
            }
            __syncthreads();
            const double ss = ssh[0], alpha = P[j][j];
            double beta = alpha, tau = 0.0, scal = 0.0;
            if (ss != 0.0) {
                beta = -copysign(sqrt(alpha * alpha + ss), alpha);
                tau = (beta - alpha) / beta;
                scal = 1.0 / (alpha - beta);
            }
            __syncthreads();
            if (tid < 64) V[tid][j] = tid < j || tid >= t ? 0.0 : (tid == j ? 1.0 : P[tid][j] * scal);
            if (tid == 0) taus[j] = tau;
            __syncthreads();
            {      // w_c = v' P[:, c] for c > j: thread (c, seg) over 8 rows each
                const int c = tid & 31, seg = tid >> 5;
                double s = 0.0;
                if (c > j)
                    for (int q = 0; q < 8; ++q) { const int i = seg * 8 + q; s = fma(V[i][j], P[i][c], s); }
                wpart[seg][c] = s;
            }
            __syncthreads();
            {
                const int i = tid >> 2, cq = tid & 3;
                const double vi = V[i][j];
                for (int q = 0; q < 8; ++q) {
                    const int c = cq + 4 * q;
                    if (c > j && i >= j) {
                        double w = 0.0;
                        for (int s8 = 0; s8 < 8; ++s8) w += wpart[s8][c];
                        P[i][c] -= tau * w * vi;
                    }
                }
                if (tid == 0) P[j][j] = beta;
            }
            __syncthreads();
        } else {
            if (tid == 0) taus[j] = 0.0;
            __syncthreads();
        }
    }
    // G = V'V
    for (int e = tid; e < NB * NB; e += 256) {
        const int a = e >> 5, b = e & 31;
        double s = 0.0;
        for (int i = 0; i < 64; ++i) s = fma(V[i][a], V[i][b], s);
        Gs[a][b] = s;
        Ts[a][b] = 0.0;
    }
    __syncthreads();
    for (int j = 0; j < NB; ++j) {      // larft: T[0:j, j] = -tau_j T[0:j, 0:j] G[0:j, j]
        if (tid < j) {
            double s = 0.0;
            for (int l = tid; l < j; ++l) s = fma(Ts[tid][l], Gs[l][j], s);
            Ts[tid][j] = -taus[j] * s;
        }
        if (tid == j) Ts[j][j] = taus[j];
        __syncthreads();
    }
    for (int e = tid; e < NB * NB; e += 256) { rec[e] = Ts[e >> 5][e & 31]; rec[NB * NB + e] = V[e >> 5][e & 31]; }
    // outputs: in place (R on / above the diagonal, V below), dense V, the partial of V'g
    double x[NB], prod[NB];
    const int i = tid;
    const bool ok = i < t;
    const double gi = ok ? g[i] : 0.0;
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        x[a] = ok && i < 64 ? V[i][a] : 0.0;
        prod[a] = x[a] * gi;
        if (ok) {
            A[(int64_t)(c0 + a) * ld + r0 + i] = i > a ? x[a] : P[i][a];
            Zc[(int64_t)a * vs + i] = x[a];
        }
    }
    if (i < t + 64) {
#pragma unroll
        for (int a = 0; a < NB; ++a) Vr[(int64_t)i * NB + a] = x[a];
    }
    const double tot = block_colsum32(prod, Pt, red);
    if (tid < NB) sgpart[tid] = tot;
}

// K3: Y = A22 V as split-K partial sums.  Block = 64 rows x one column range; its four waves take a quarter of the range
// each and keep a 64 x 32 accumulator (8 MFMA tiles): per k-step of 4 columns a lane loads two 16-byte row pairs of A22
// (the 64 rows as {even, odd} x {low, high} tiles, so that the loads are 256 contiguous bytes per column) and two V
// entries (row-major V: 128 contiguous bytes per knot).  A22 is read exactly once: the kernel streams.  One EXTRA block
// (blockIdx.x == gridDim.x - 1, blockIdx.y == 0) finishes the panel's T and band entries beside it (b32_tfin).
__global__ __launch_bounds__(256) void b32_symm_kernel(double *__restrict__ A, int64_t ld, int r0, int t,
                                                       const double *__restrict__ Vr, double *__restrict__ Ypart, int64_t vs,
                                                       int ws, int tfin, const double *__restrict__ aux, const double *__restrict__ R1,
                                                       double *__restrict__ rec, int *__restrict__ flags) {
    __shared__ double red[4][NB][66];
    if (blockIdx.x == gridDim.x - 1) {
        if (blockIdx.y == 0 && tfin) b32_tfin(A, ld, r0 - NB, r0, aux, R1, rec, flags, &red[0][0][0]);
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int I0 = blockIdx.x * 64;
    B32_PROF_BEGIN
    const int jbeg = blockIdx.y * ws + wave * (ws >> 2);
    const int jend = min(jbeg + (ws >> 2), (t + 3) & ~3);
    const double *Ab = A + (int64_t)r0 * ld + r0;
    d4v acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (d4v){0.0, 0.0, 0.0, 0.0};
    const int rp0 = I0 + 2 * l15, rp1 = rp0 + 32;
    const bool ok00 = rp0 < t, ok01 = rp0 + 1 < t, ok10 = rp1 < t, ok11 = rp1 + 1 < t;
    const double *a0p = Ab + (ok00 ? rp0 : 0), *a1p = Ab + (ok10 ? rp1 : 0);
    // four k-steps per trip, their sixteen global loads issued before the first MFMA (the compiler declines to unroll this
    // loop by itself: with two loads in flight per wave the kernel streamed at 2.7 TB/s)
    // Round 6: the trailing matrix is kept in its LOWER triangle only (b32_rankk_kernel updates the tiles on and below the
    // diagonal: half the bytes and half the MFMA work of the update, the largest kernel of a large fit).  An entry above the
    // diagonal is read from its mirror image, A22[r][c] = A22[c][r] at Ab + c + r * ld: for a fixed row that is a
    // contiguous run along c, so over the four k-steps of a trip every 128-byte line a lane touches is used in full.
    const int r00 = ok00 ? rp0 : 0, r01 = ok01 ? rp0 + 1 : 0, r10 = ok10 ? rp1 : 0, r11 = ok11 ? rp1 + 1 : 0;
    for (int j = jbeg; j < jend; j += 16) {
        double2 a0[4], a1[4];
        double b0[4], b1[4];
        if (j + 16 <= I0) {      // every column of the trip lies left of the block's rows: the stored triangle
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jc = j + 4 * u + l4;
                const int col = min(jc, t - 1);
                a0[u] = *(const double2 *)(a0p + (int64_t)col * ld);
                a1[u] = *(const double2 *)(a1p + (int64_t)col * ld);
                const double *vrow = Vr + (int64_t)min(jc, t + 59) * NB + l15;      // rows t .. t + 63 of Vr are zero
                b0[u] = vrow[0]; b1[u] = vrow[16];
            }
        } else if (j >= I0 + 64) {      // every column of the trip lies right of the block's rows: the mirror image, whole.
            // Along a mirrored row the columns are contiguous, so a lane takes them in PAIRS (16-byte loads): k-steps 2 q and
            // 2 q + 1 of the trip stand for columns j + 8 q + 2 l4 and + 1 -- the sum over a trip's 16 columns does not care
            // in which order they come, as long as A22's and V's operands agree.
            const int tcl = (t - 2) & ~1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int jc = j + 8 * q + 2 * l4;
                const int col = jc < t ? jc : tcl;      // (jc = t - 1: its partner, column t, is finite memory of the same row and meets a zero row of V)
                const double2 m00 = *(const double2 *)(Ab + (int64_t)r00 * ld + col), m01 = *(const double2 *)(Ab + (int64_t)r01 * ld + col);
                const double2 m10 = *(const double2 *)(Ab + (int64_t)r10 * ld + col), m11 = *(const double2 *)(Ab + (int64_t)r11 * ld + col);
                const bool lv0 = jc < jend, lv1 = jc + 1 < jend;
                a0[2 * q].x = lv0 ? m00.x : 0.0; a0[2 * q].y = lv0 ? m01.x : 0.0; a1[2 * q].x = lv0 ? m10.x : 0.0; a1[2 * q].y = lv0 ? m11.x : 0.0;
                a0[2 * q + 1].x = lv1 ? m00.y : 0.0; a0[2 * q + 1].y = lv1 ? m01.y : 0.0; a1[2 * q + 1].x = lv1 ? m10.y : 0.0; a1[2 * q + 1].y = lv1 ? m11.y : 0.0;
                const double *vrow = Vr + (int64_t)min(jc, t + 58) * NB + l15;
                b0[2 * q] = vrow[0]; b1[2 * q] = vrow[16];
                b0[2 * q + 1] = vrow[NB]; b1[2 * q + 1] = vrow[NB + 16];
            }
        } else {                 // the trips across the diagonal: entry by entry, the mirror image where the column is the larger index
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jc = j + 4 * u + l4;
                const int col = min(jc, t - 1);
                a0[u].x = col <= r00 ? Ab[(int64_t)col * ld + r00] : Ab[(int64_t)r00 * ld + col];
                a0[u].y = col <= r01 ? Ab[(int64_t)col * ld + r01] : Ab[(int64_t)r01 * ld + col];
                a1[u].x = col <= r10 ? Ab[(int64_t)col * ld + r10] : Ab[(int64_t)r10 * ld + col];
                a1[u].y = col <= r11 ? Ab[(int64_t)col * ld + r11] : Ab[(int64_t)r11 * ld + col];
                const double *vrow = Vr + (int64_t)min(jc, t + 59) * NB + l15;
                b0[u] = vrow[0]; b1[u] = vrow[16];
            }
        }
        const bool mirrored = j >= I0 + 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool live = mirrored || j + 4 * u < jend;
            const double x0 = ok00 && live ? a0[u].x : 0.0, x1 = ok01 && live ? a0[u].y : 0.0;
            const double x2 = ok10 && live ? a1[u].x : 0.0, x3 = ok11 && live ? a1[u].y : 0.0;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b0[u], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b1[u], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b0[u], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b1[u], acc[1][1], 0, 0, 0);
            acc[2][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b0[u], acc[2][0], 0, 0, 0);
            acc[2][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b1[u], acc[2][1], 0, 0, 0);
            acc[3][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b0[u], acc[3][0], 0, 0, 0);
            acc[3][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b1[u], acc[3][1], 0, 0, 0);
        }
    }
    // tile rt = 2 h + parity holds rows I0 + 32 h + 2 m + parity, m = l4 + 4 r; column n = 16 nt + l15
    B32_MARK(3, 0);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][nt * 16 + l15][32 * (rt >> 1) + 2 * (l4 + 4 * r) + (rt & 1)] = acc[rt][nt][r];
    __syncthreads();
    for (int e = threadIdx.x; e < NB * 64; e += 256) {
        const int n = e >> 6, row = e & 63;
        const double s = (red[0][n][row] + red[1][n][row]) + (red[2][n][row] + red[3][n][row]);
        if (I0 + row < t) Ypart[((int64_t)blockIdx.y * NB + n) * vs + I0 + row] = s;
    }
    B32_MARK(3, 1);
}
static_assert(sizeof(double) * 4 * NB * 66 >= sizeof(double) * TFIN_LDS, "b32_tfin borrows the symmetric product's LDS");

// K4a: Y = sum of the split-K partials (at most 4), W^ = Y T on MFMA (row-major), the block's share of M = V'Y, and
// g <- H'g = g - V T'(V'g)
constexpr int SYMM_MAXSPLIT = 4;
__global__ __launch_bounds__(256) void b32_w_kernel(int t, int cpb, int nsplit, const double *__restrict__ Ypart, int64_t vs,
                                                    const double *__restrict__ Vr, const double *__restrict__ Tm,
                                                    const double *__restrict__ sgpart, int nsg, double *__restrict__ g,
                                                    double *__restrict__ Wh, double *__restrict__ Mpart) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *Vt = smem, *Yt = smem + NB * PT_S;
    double(*Ts)[NB + 1] = (double(*)[NB + 1])(smem + 2 * NB * PT_S);
    double *zs = smem + 2 * NB * PT_S + NB * (NB + 1), *sg = zs + NB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4, ta = wave >> 1, tb = wave & 1;
    d4v macc = {0.0, 0.0, 0.0, 0.0};
    B32_PROF_BEGIN
    for (int ch = 0; ch < cpb; ++ch) {
        const int i = (blockIdx.x * cpb + ch) * CHR + threadIdx.x;
        const bool ok = i < t;
        const int ii = ok ? i : 0;
        // everything this chunk reads from global memory, issued together
        double tq[4], sgv = 0.0, gi, v[NB];
        if (ch == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tq[q] = Tm[threadIdx.x + 256 * q];
            if (threadIdx.x < NB) sgv = sum_parts_32(sgpart, nsg);
        }
        gi = g[ii];
        {
            const double2 *src = (const double2 *)(Vr + (int64_t)ii * NB);
#pragma unroll
            for (int a = 0; a < NB; a += 2) { const double2 q = src[a >> 1]; v[a] = ok ? q.x : 0.0; v[a + 1] = ok ? q.y : 0.0; }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            double yp[16][SYMM_MAXSPLIT];
#pragma unroll
            for (int n = 0; n < 16; ++n)
#pragma unroll
                for (int sp = 0; sp < SYMM_MAXSPLIT; ++sp) {
                    const double x = Ypart[((int64_t)(sp < nsplit ? sp : 0) * NB + half * 16 + n) * vs + ii];
                    yp[n][sp] = sp < nsplit && ok ? x : 0.0;
                }
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                double s = 0.0;
#pragma unroll
                for (int sp = 0; sp < SYMM_MAXSPLIT; ++sp) s += yp[n][sp];
                Yt[(half * 16 + n) * PT_S + threadIdx.x] = s;
            }
        }
#pragma unroll
        for (int a = 0; a < NB; ++a) Vt[a * PT_S + threadIdx.x] = v[a];
        if (ch == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int e = threadIdx.x + 256 * q; Ts[e >> 5][e & 31] = tq[q]; }
            if (threadIdx.x < NB) sg[threadIdx.x] = sgv;
            __syncthreads();
            if (threadIdx.x < NB) {      // z = T' sg
                const int a = threadIdx.x;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
                for (int b = 0; b < NB; b += 2) { s0 = fma(Ts[b][a], sg[b], s0); s1 = fma(Ts[b + 1][a], sg[b + 1], s1); }
                zs[a] = s0 + s1;
            }
        }
        __syncthreads();
        B32_MARK(4, 0);
        if (ok) {
#pragma unroll
            for (int a = 0; a < NB; ++a) gi = fma(-v[a], zs[a], gi);
            g[i] = gi;
        }
        gram_chunk(Vt, Yt, macc, ta, tb);
        B32_MARK(4, 1);
        // W^ = Y T: wave w takes rows 64 w .. 64 w + 63 (4 row tiles) x 2 column tiles
        d4v wacc[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wacc[rt][nt] = (d4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < NB; k0 += 4) {
            const double b0 = Ts[k0 + l4][l15], b1 = Ts[k0 + l4][16 + l15];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const double av = Yt[(k0 + l4) * PT_S + wave * 64 + rt * 16 + l15];
                wacc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, wacc[rt][0], 0, 0, 0);
                wacc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, wacc[rt][1], 0, 0, 0);
            }
        }
        const int rbase = (blockIdx.x * cpb + ch) * CHR + wave * 64;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rbase + rt * 16 + l4 + 4 * r;
                if (row < t) {
                    Wh[(int64_t)row * NB + l15] = wacc[rt][0][r];
                    Wh[(int64_t)row * NB + 16 + l15] = wacc[rt][1][r];
                }
            }
        __syncthreads();
        B32_MARK(4, 2);
    }
    store_tile32(Mpart + (size_t)blockIdx.x * NB * NB, macc, ta, tb);
    B32_MARK(4, 3);
}
constexpr size_t B32_W_LDS = sizeof(double) * (2 * NB * PT_S + NB * (NB + 1) + 2 * NB);

// K4b: S = sym(T' sym(M) T), W = W^ - 1/2 V S on MFMA -> the W half of Z (column-major)
__global__ __launch_bounds__(256) void b32_wfin_kernel(int t, int cpb, const double *__restrict__ Mpart, int nM,
                                                       const double *__restrict__ Tm, const double *__restrict__ Vr,
                                                       const double *__restrict__ Wh, double *__restrict__ Zw, int64_t vs) {
    __shared__ double Vt[NB * PT_S];
    __shared__ double M[NB][NB + 1], X[NB][NB + 1], S[NB][NB + 1], Ts[NB][NB + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    B32_PROF_BEGIN
    {
        Parts4 P;
        parts_issue(P, Mpart, nM);
        double tq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) tq[q] = Tm[threadIdx.x + 256 * q];
        parts_sum(P, M);
        if (nM > MAXPART_H) { parts_issue(P, Mpart + (size_t)MAXPART_H * NB * NB, nM - MAXPART_H); parts_add(P, M); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = threadIdx.x + 256 * q; Ts[e >> 5][e & 31] = tq[q]; }
    }
    __syncthreads();
    B32_MARK(5, 0);
    for (int e = threadIdx.x; e < NB * NB; e += 256) { const int a = e >> 5, c = e & 31; S[a][c] = 0.5 * (M[a][c] + M[c][a]); }
    __syncthreads();
    {   // X = sym(M) T, then T'X: two 32 x 32 products on MFMA (a tile per wave)
        const int ta = wave >> 1, tb = wave & 1;
        const d4v x = mm32_tile<false>(S, Ts);
#pragma unroll
        for (int r = 0; r < 4; ++r) X[ta * 16 + l4 + 4 * r][tb * 16 + l15] = x[r];
        __syncthreads();
        const d4v y = mm32_tile<true>(Ts, X);
#pragma unroll
        for (int r = 0; r < 4; ++r) M[ta * 16 + l4 + 4 * r][tb * 16 + l15] = y[r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NB * NB; e += 256) { const int a = e >> 5, c = e & 31; S[a][c] = -0.25 * (M[a][c] + M[c][a]); }      // -1/2 sym
    B32_MARK(5, 1);
    for (int ch = 0; ch < cpb; ++ch) {
        const int cbase = (blockIdx.x * cpb + ch) * CHR;
        const int i = cbase + threadIdx.x;
        const bool ok = i < t;
        {
            const double2 *src = (const double2 *)(Vr + (int64_t)(ok ? i : 0) * NB);
            double v[NB];
#pragma unroll
            for (int a = 0; a < NB; a += 2) { const double2 q = src[a >> 1]; v[a] = ok ? q.x : 0.0; v[a + 1] = ok ? q.y : 0.0; }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < NB; ++a) Vt[a * PT_S + threadIdx.x] = v[a];
        }
        // accumulators start at W^ (rows l4 + 4 r of the tile, column l15)
        d4v wacc[4][2];
        const int rbase = cbase + wave * 64;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rbase + rt * 16 + l4 + 4 * r;
                const int rr = row < t ? row : 0;
                const double w0 = Wh[(int64_t)rr * NB + l15], w1 = Wh[(int64_t)rr * NB + 16 + l15];
                wacc[rt][0][r] = row < t ? w0 : 0.0;
                wacc[rt][1][r] = row < t ? w1 : 0.0;
            }
        __syncthreads();
        B32_MARK(5, 2);
#pragma unroll
        for (int k0 = 0; k0 < NB; k0 += 4) {
            const double b0 = S[k0 + l4][l15], b1 = S[k0 + l4][16 + l15];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                const double av = Vt[(k0 + l4) * PT_S + wave * 64 + rt * 16 + l15];
                wacc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, wacc[rt][0], 0, 0, 0);
                wacc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, wacc[rt][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rbase + rt * 16 + l4 + 4 * r;
                if (row < t) {
                    Zw[(int64_t)l15 * vs + row] = wacc[rt][0][r];
                    Zw[(int64_t)(16 + l15) * vs + row] = wacc[rt][1][r];
                }
            }
        B32_MARK(5, 3);
    }
}

// K5: A22 -= V W' + W V' = Z Zs' with Z = [V | W] (t x 64, column-major, stride vs) and Zs = [W | V], on the 128 x 128 tiles
// on and below the diagonal of the t x t block (round 6: the upper triangle is never read again -- the symmetric product
// takes it from its mirror image -- so it is not updated either).  The tile loop is tps_fit.hip's band_rankk_kernel with K = 64: 4 waves
// x 64 x 64, K streamed through two LDS buffers in chunks of 16, the C tile preloaded into the accumulators with a negated
// operand.  col0_only: the first block column only (look-ahead: the next panel lives in its first 32 columns).
constexpr int RK_T = 128, RK_KC = 16, RK_S = RK_T + 16, RK_K = 2 * NB;
__global__ __launch_bounds__(256, 2) void b32_rankk_kernel(double *__restrict__ A, int64_t ld, int r0, int t,
                                                           const double *__restrict__ Z, int64_t vs, int nt) {
    __shared__ __attribute__((aligned(16))) double sI[2][RK_KC * RK_S];
    __shared__ __attribute__((aligned(16))) double sJ[2][RK_KC * RK_S];
    // tiles on and below the diagonal only, column by column (round 6: b32_symm_kernel reads the upper triangle from its
    // mirror image): block index -> (bi >= bj)
    int bj = 0, rem = blockIdx.x;
    while (rem >= nt - bj) { rem -= nt - bj; ++bj; }
    const int bi = bj + rem;
    const int lmin = NB;      // columns 0 .. 31 belong to b32_strip_kernel (and, by now, to the next panel's kernels)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
    double *C = A + (int64_t)r0 * ld + r0;
    d4v acc[4][4];
    B32_PROF_BEGIN
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = bj * RK_T + wj + a * 16 + l4 + 4 * r;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = bi * RK_T + wi + b * 16 + l15;
                acc[a][b][r] = (i < t && l < t) ? C[(int64_t)l * ld + i] : 0.0;
            }
        }
    const int gk = tid >> 6, gr = tid & 63;
    const int rI0 = bi * RK_T + gr, rI1 = rI0 + 64, rJ0 = bj * RK_T + gr, rJ1 = rJ0 + 64;
    double gI[4][2], gJ[4][2];
#define RK_GLOAD(K0)                                                                                   \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                    \
        const int kp = (K0) + gk + 4 * q;                                                              \
        const double *cp = Z + (int64_t)kp * vs, *cq = Z + (int64_t)((kp + NB) & (RK_K - 1)) * vs;    \
        gI[q][0] = rI0 < t ? cp[rI0] : 0.0; gI[q][1] = rI1 < t ? cp[rI1] : 0.0;                        \
        gJ[q][0] = rJ0 < t ? cq[rJ0] : 0.0; gJ[q][1] = rJ1 < t ? cq[rJ1] : 0.0;                        \
    }
#define RK_SSTORE(BUF)                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                    \
        sI[BUF][(gk + 4 * q) * RK_S + gr] = gI[q][0]; sI[BUF][(gk + 4 * q) * RK_S + gr + 64] = gI[q][1]; \
        sJ[BUF][(gk + 4 * q) * RK_S + gr] = gJ[q][0]; sJ[BUF][(gk + 4 * q) * RK_S + gr + 64] = gJ[q][1]; \
    }
    RK_GLOAD(0)
    RK_SSTORE(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no prologue load (the C tile) pending into the loop
    __syncthreads();
    B32_MARK(6, 4);
    for (int c = 0; c < RK_K / RK_KC; ++c) {
        const int buf = c & 1;
        if (c + 1 < RK_K / RK_KC) { RK_GLOAD((c + 1) * RK_KC) }
#pragma unroll
        for (int kk = 0; kk < RK_KC; kk += 4) {
            double fi[4], fj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                fj[a] = -sJ[buf][(kk + l4) * RK_S + wj + a * 16 + l15];
                fi[a] = sI[buf][(kk + l4) * RK_S + wi + a * 16 + l15];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[a], fi[b], acc[a][b], 0, 0, 0);
        }
        if (c + 1 < RK_K / RK_KC) {
            RK_SSTORE(buf ^ 1)
            __syncthreads();
        }
    }
#undef RK_GLOAD
#undef RK_SSTORE
    B32_MARK(6, 5);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = bj * RK_T + wj + a * 16 + l4 + 4 * r;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = bi * RK_T + wi + b * 16 + l15;
                if (i < t && l < t && l >= lmin) C[(int64_t)l * ld + i] = acc[a][b][r];
            }
        }
    B32_MARK(6, 6);
}

// K5a: the same update for columns 0 .. 31 of A22 alone -- the next panel and the next diagonal block of the band, ALL the
// next panel's kernels wait for.  (The first version ran K5 on the whole first block column, 128 columns: one CU needs
// 6.8 us of FP64 MFMA time for a 128 x 128 x 64 tile, and 40 CUs were busy.)  Block = 64 rows, wave = 16 rows x 32 columns;
// no LDS: a lane loads its 16 + 2 x 16 MFMA operands straight from Z (16 contiguous doubles per quarter-wave), all before
// the first MFMA.
__global__ __launch_bounds__(256) void b32_strip_kernel(double *__restrict__ A, int64_t ld, int r0, int t, const double *__restrict__ Z,
                                                        int64_t vs) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int i = blockIdx.x * 64 + wave * 16 + l15;      // row of A22 this lane's B operands and results belong to
    double *C = A + (int64_t)r0 * ld + r0;
    B32_PROF_BEGIN
    if (blockIdx.x * 64 + wave * 16 >= t) return;
    const int ii = i < t ? i : 0;
    d4v acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = a * 16 + l4 + 4 * r;
            const double x = C[(int64_t)(l < t ? l : 0) * ld + ii];
            acc[a][r] = (i < t && l < t) ? x : 0.0;
        }
    double fi[RK_K / 4], fj[2][RK_K / 4];
#pragma unroll
    for (int q = 0; q < RK_K / 4; ++q) {
        const int kp = 4 * q + l4;
        fi[q] = Z[(int64_t)kp * vs + ii];
        const double *cq = Z + (int64_t)((kp + NB) & (RK_K - 1)) * vs;
        fj[0][q] = cq[l15 < t ? l15 : 0];
        fj[1][q] = cq[16 + l15 < t ? 16 + l15 : 0];
    }
    B32_MARK(6, 0);
#pragma unroll
    for (int q = 0; q < RK_K / 4; ++q) {
        const double b = i < t ? fi[q] : 0.0;
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 < t ? -fj[0][q] : 0.0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(16 + l15 < t ? -fj[1][q] : 0.0, b, acc[1], 0, 0, 0);
    }
    B32_MARK(6, 1);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int l = a * 16 + l4 + 4 * r;
            if (i < t && l < t) C[(int64_t)l * ld + i] = acc[a][r];
        }
    B32_MARK(6, 2);
}

// K5a and the NEXT panel's K0 in one launch: the strip update with 256-row blocks dealt as row_blocks(t - 32) deals the next
// panel (rows 32 .. t - 1 of this A22), each block leaving the partial Gram matrix of its rows beside the updated entries --
// what b32_gram_kernel would compute from them, bit for bit, without the launch and the round trip through memory between
// the two (strip 8 us + gap 3 us + Gram 12-15 us under the trailing update's traffic -> ~11 us).  Wave = 64 rows x 32
// columns (128 MFMAs); block 0's first two waves also update the 32 rows above the panel (the next diagonal block).
__global__ __launch_bounds__(256) void b32_stripgram_kernel(double *__restrict__ A, int64_t ld, int r0, int t,
                                                            const double *__restrict__ Z, int64_t vs, int cpb,
                                                            double *__restrict__ Gpart) {
    __shared__ double Pt[NB * PT_S];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4, ta = wave >> 1, tb = wave & 1;
    double *C = A + (int64_t)r0 * ld + r0;
    B32_PROF_BEGIN
    double fj[2][RK_K / 4];      // rows 0 .. 31 of [W | V], negated: the A operands of every tile
#pragma unroll
    for (int q = 0; q < RK_K / 4; ++q) {
        const double *cq = Z + (int64_t)((4 * q + l4 + NB) & (RK_K - 1)) * vs;
        fj[0][q] = -cq[l15];
        fj[1][q] = -cq[16 + l15];
    }
    d4v gacc = {0.0, 0.0, 0.0, 0.0};
    for (int ch = 0; ch < cpb; ++ch) {
        const int base = NB + (blockIdx.x * cpb + ch) * CHR + wave * 64;      // first of the wave's 64 rows of A22
        d4v acc[4][2];
        double fi[4][RK_K / 4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int row = base + rt * 16 + l15;
            const bool ok = row < t;
            const int rr = ok ? row : 0;
#pragma unroll
            for (int q = 0; q < RK_K / 4; ++q) { const double x = Z[(int64_t)(4 * q + l4) * vs + rr]; fi[rt][q] = ok ? x : 0.0; }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const double x = C[(int64_t)(a * 16 + l4 + 4 * r) * ld + rr]; acc[rt][a][r] = ok ? x : 0.0; }
        }
        B32_MARK(6, 0);
#pragma unroll
        for (int q = 0; q < RK_K / 4; ++q)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                acc[rt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[0][q], fi[rt][q], acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[1][q], fi[rt][q], acc[rt][1], 0, 0, 0);
            }
        B32_MARK(6, 1);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int row = base + rt * 16 + l15;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int l = a * 16 + l4 + 4 * r;
                    if (row < t) C[(int64_t)l * ld + row] = acc[rt][a][r];
                    Pt[l * PT_S + wave * 64 + rt * 16 + l15] = acc[rt][a][r];      // zero past the last row
                }
        }
        __syncthreads();
        B32_MARK(6, 2);
        gram_chunk(Pt, Pt, gacc, ta, tb);
        __syncthreads();
        B32_MARK(0, 1);
    }
    store_tile32(Gpart + (size_t)blockIdx.x * NB * NB, gacc, ta, tb);
    if (blockIdx.x == 0 && wave < 2) {      // rows 0 .. 31 of A22
        const int row = wave * 16 + l15;
        d4v acc[2];
        double fi[RK_K / 4];
#pragma unroll
        for (int q = 0; q < RK_K / 4; ++q) fi[q] = Z[(int64_t)(4 * q + l4) * vs + row];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][r] = C[(int64_t)(a * 16 + l4 + 4 * r) * ld + row];
#pragma unroll
        for (int q = 0; q < RK_K / 4; ++q) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[0][q], fi[q], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[1][q], fi[q], acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(int64_t)(a * 16 + l4 + 4 * r) * ld + row] = acc[a][r];
    }
    B32_MARK(0, 2);
}

// lower band of B -> ab[j * 33 + d] = B[j + d][j]
__global__ void b32_extract_kernel(const double *__restrict__ A, int64_t ld, int off0, int m, double *__restrict__ ab) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * (NB + 1)) return;
    const int j = e / (NB + 1), d = e - j * (NB + 1);
    ab[e] = (j + d < m) ? A[(int64_t)(off0 + j) * ld + off0 + j + d] : 0.0;
}

// ------------------------------------------------------------------------------------------ Q'g replay, Q q --
// g <- H_p' g for ANOTHER right-hand side with a stored reduction (the reduction cache's refits): two launches per panel
// that repeat what K2 (partials of V'g, same row blocks, same order) and K4a (z = T' sum, g -= V z) did to g during the
// reduction, so that the rotated right-hand side is the full fit's bit for bit.
__global__ __launch_bounds__(256) void b32_qt_part_kernel(const double *__restrict__ A, int64_t ld, int c0, int r0, int t, int cpb,
                                                          const double *__restrict__ rec, const double *__restrict__ g,
                                                          double *__restrict__ sgpart) {
    __shared__ double Pt[NB * PT_S];
    __shared__ double red[8][NB];
    double sgacc = 0.0;
    for (int ch = 0; ch < cpb; ++ch) {
        const int i = (blockIdx.x * cpb + ch) * CHR + threadIdx.x;
        const bool ok = i < t;
        double v[NB], prod[NB];
        load_v_row(v, A + (int64_t)c0 * ld + r0, ld, i, ok, rec + NB * NB);
        const double gi = ok ? g[i] : 0.0;
#pragma unroll
        for (int a = 0; a < NB; ++a) prod[a] = v[a] * gi;
        sgacc += block_colsum32(prod, Pt, red);
    }
    if (threadIdx.x < NB) sgpart[(size_t)blockIdx.x * NB + threadIdx.x] = sgacc;
}
__global__ __launch_bounds__(256) void b32_qt_apply_kernel(const double *__restrict__ A, int64_t ld, int c0, int r0, int t,
                                                           const double *__restrict__ Tm, const double *__restrict__ sgpart, int nsg,
                                                           double *__restrict__ g) {
    __shared__ double Ts[NB][NB + 1], zs[NB], sg[NB];
    for (int e = threadIdx.x; e < NB * NB; e += 256) Ts[e >> 5][e & 31] = Tm[e];
    if (threadIdx.x < NB) sg[threadIdx.x] = sum_parts_32(sgpart, nsg);
    __syncthreads();
    if (threadIdx.x < NB) {      // the very sums of b32_w_kernel
        const int a = threadIdx.x;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int b = 0; b < NB; b += 2) { s0 = fma(Ts[b][a], sg[b], s0); s1 = fma(Ts[b + 1][a], sg[b + 1], s1); }
        zs[a] = s0 + s1;
    }
    __syncthreads();
    const int i = blockIdx.x * CHR + threadIdx.x;
    if (i >= t) return;
    double v[NB];
    load_v_row(v, A + (int64_t)c0 * ld + r0, ld, i, true, Tm + NB * NB);
    double gi = g[i];
#pragma unroll
    for (int a = 0; a < NB; ++a) gi = fma(-v[a], zs[a], gi);
    g[i] = gi;
}

// r <- Q r = H_0 H_1 ... H_{P-1} r, one many-block launch per panel: launch p applies panel p (r -= V_p (T_p s_p), s_p =
// V_p'r summed from the previous launch's partials) and, in the same pass over its rows, forms the partials of s_{p-1}.
__global__ __launch_bounds__(256) void b32_bt_step_kernel(const double *__restrict__ A, int64_t ld, int off0, int m, int p,
                                                          int npanels, const double *__restrict__ Tall, double *__restrict__ r,
                                                          double *__restrict__ part /* [2][BT_MAXBLK][32] */) {
    __shared__ double Pt[NB * PT_S];
    __shared__ double red[8][NB], ssum[NB], zs[NB];
    const int i = blockIdx.x * CHR + threadIdx.x;
    double ri = i < m ? r[i] : 0.0;
    if (p < npanels) {
        const int base = p * NB + NB;
        {      // 8 groups of 32 threads take every 8th block's partial (loads in flight together), then the groups in order
            const int a = threadIdx.x & 31, grp = threadIdx.x >> 5;
            double v[B32_BT_MAXBLK / 8];
#pragma unroll
            for (int q = 0; q < B32_BT_MAXBLK / 8; ++q) {
                const unsigned b = grp + 8 * q;
                const double x = part[((size_t)(p & 1) * B32_BT_MAXBLK + (b < gridDim.x ? b : 0)) * NB + a];
                v[q] = b < gridDim.x ? x : 0.0;
            }
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < B32_BT_MAXBLK / 8; ++q) sacc += v[q];
            red[grp][a] = sacc;
        }
        __syncthreads();
        if (threadIdx.x < NB) {
            double s = 0.0;
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) s += red[grp][threadIdx.x];
            ssum[threadIdx.x] = s;
        }
        __syncthreads();
        if (threadIdx.x < NB) {
            const int a = threadIdx.x;
            const double *T = Tall + (size_t)p * PREC;
            double s = 0.0;
            for (int b = 0; b < NB; ++b) s = fma(T[a * NB + b], ssum[b], s);
            zs[a] = s;
        }
        __syncthreads();
        const int il = i - base;
        if (il >= 0 && i < m) {
            double v[NB];
            load_v_row(v, A + (int64_t)(off0 + p * NB) * ld + off0 + base, ld, il, true, Tall + (size_t)p * PREC + NB * NB);
#pragma unroll
            for (int a = 0; a < NB; ++a) ri = fma(-v[a], zs[a], ri);
            r[i] = ri;
        }
    }
    if (p > 0) {
        const int q = p - 1, base = q * NB + NB, il = i - base;
        const bool ok = il >= 0 && i < m;
        double v[NB], prod[NB];
        load_v_row(v, A + (int64_t)(off0 + q * NB) * ld + off0 + base, ld, il, ok, Tall + (size_t)q * PREC + NB * NB);
#pragma unroll
        for (int a = 0; a < NB; ++a) prod[a] = v[a] * ri;
        const double tot = block_colsum32(prod, Pt, red);
        if (threadIdx.x < NB) part[((size_t)(q & 1) * B32_BT_MAXBLK + blockIdx.x) * NB + threadIdx.x] = tot;
    }
}


// ------------------------------------------------------------------------------------- GCV terms on the band --
// One lambda, one direction: LDL' of M = Bb + lambda I column by column on a window of 34 live rows in LDS, 256 threads, one
// barrier per column, carrying d/dlambda of every quantity (DERIV).  Logical index jj runs from the top (dir 0: matrix
// index jj) or from the bottom (dir 1: matrix index m-1-jj); the sweep eliminates `ncols` columns and leaves the window
// holding the 32 x 32 block behind them (the middle block of the twisted factorisation, with the eliminated part's Schur
// complement folded in), which b32_mid_kernel joins with the other direction's.  STORE (the final solve): the factor's
// columns and the forward-substituted right-hand side go to global memory.
//   window: entry (r, c), r >= c, at Wb[r mod 40][r - c] -- band-relative, so only ROW indices wrap and every column offset
//   a thread needs is a constant of the thread; the 528 entries (jj + i, jj + k), 1 <= k <= i <= 32, a step updates are
//   dealt to the threads as three slots each (slot e = tid + 256 q; rows of the triangle are contiguous in e).
constexpr int SW = 64;                       // rows jj .. jj + 33 are live; index mod 64 (one v_and)
constexpr int SWS = NB + 2;                  // row stride (doubles): offsets 0 .. 32 = the band, 33 = the right-hand side y
constexpr int SYO = NB + 1;                  // offset of y in a row
constexpr int SDUMMY = SW * SWS;             // where idle slots write
constexpr int SCH = 7;                       // rows per staging chunk (7 x 34 = 238 values: one register per thread)
constexpr int SWIN = 2 * NB * NB + 2 * NB;   // doubles of a stored window: W, dW (32 x 32, [a][b]), y, dy
struct SweepLds {
    double W[SW * SWS + 2], dW[SW * SWS + 2];
    double stage[2][SCH][NB + 2];            // [..][33] = g of the row
};
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // no vmcnt(0): the staging loads stay in flight

// A step subtracts p_i l_k from entry (jj + i, jj + k) for 1 <= k <= i <= 32 (528 entries) and p_i (y_jj / d) from y_(jj+i)
// (32 more: the right-hand side rides along as offset 33 of every row, "k = 0, offset 33").  560 slots, three per thread
// (slot e = tid + 256 q); the step is straight-line code -- all of a thread's LDS reads are issued before anything waits,
// idle slots write to a dummy word -- because a chain of predicated read-modify-write blocks costs an LDS round trip each.
struct SweepSlots { int i[3], k[3], offk[3], offw[3]; bool on[3]; };
__device__ __forceinline__ SweepSlots sweep_slots() {
    SweepSlots s;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = threadIdx.x + 256 * q, ntri = NB * (NB + 1) / 2;
        int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);      // row of the triangle: i (i - 1) / 2 <= e < i (i + 1) / 2
        while (i * (i - 1) / 2 > e) --i;
        while (i * (i + 1) / 2 <= e) ++i;
        if (e < ntri) { s.on[q] = true; s.i[q] = i; s.k[q] = e - i * (i - 1) / 2 + 1; s.offk[q] = s.k[q]; s.offw[q] = i - s.k[q]; }
        else if (e < ntri + NB) { s.on[q] = true; s.i[q] = e - ntri + 1; s.k[q] = 0; s.offk[q] = SYO; s.offw[q] = SYO; }
        else { s.on[q] = false; s.i[q] = 1; s.k[q] = 1; s.offk[q] = 1; s.offw[q] = 0; }
    }
    return s;
}

template <bool DERIV>
__device__ __forceinline__ void sweep_step(SweepLds &L, const SweepSlots &sl, int o, double &neg, double &tr, double &q2,
                                           double *__restrict__ Lcol, double *__restrict__ ycol) {
    int ai[3], ak[3], aw[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int ri = (o + sl.i[q]) & (SW - 1), rk = (o + sl.k[q]) & (SW - 1);
        ai[q] = ri * SWS + sl.i[q]; ak[q] = rk * SWS + sl.offk[q];
        aw[q] = sl.on[q] ? ri * SWS + sl.offw[q] : SDUMMY;
    }
    double d = L.W[o * SWS], dd = 0.0, pi[3], pk[3], w[3], dpi[3], dpk[3], dw[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { pi[q] = L.W[ai[q]]; pk[q] = L.W[ak[q]]; w[q] = L.W[aw[q]]; }
    if (DERIV) {
        dd = L.dW[o * SWS];
#pragma unroll
        for (int q = 0; q < 3; ++q) { dpi[q] = L.dW[ai[q]]; dpk[q] = L.dW[ak[q]]; dw[q] = L.dW[aw[q]]; }
    }
    if (d == 0.0) d = -1e-300;
    const double inv = frcp(d);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double lk = pk[q] * inv;
        L.W[aw[q]] = w[q] - pi[q] * lk;
        if (DERIV) {
            const double dlk = (dpk[q] - lk * dd) * inv;
            L.dW[aw[q]] = dw[q] - (dpi[q] * lk + pi[q] * dlk);
        }
        if (Lcol && sl.on[q] && sl.k[q] == 0) Lcol[sl.i[q]] = pi[q] * inv;
    }
    if ((threadIdx.x >> 6) == 2) {      // one wave keeps the books (lane 0's copy is the one written out)
        const double y0 = L.W[o * SWS + SYO];
        if (d < 0.0) neg += 1.0;
        if (DERIV) {
            const double dy0 = L.dW[o * SWS + SYO];
            tr += dd * inv;
            q2 -= (2.0 * y0 * dy0 * d - y0 * y0 * dd) * (inv * inv);
        }
        if (Lcol && threadIdx.x == 128) { Lcol[0] = d; *ycol = y0; }
    }
}

// joins the two directions of one lambda: S = Wtop + Wbot (mirrored) - (the block itself), eliminates its mb columns
template <bool DERIV>
__global__ __launch_bounds__(256) void b32_mid_kernel(const double *__restrict__ ab, const double *__restrict__ g, int m, int mid,
                                                      int mb, const double *__restrict__ lams, const double *__restrict__ win,
                                                      const double *__restrict__ res, double *__restrict__ out) {
    __shared__ SweepLds L;
    const int tid = threadIdx.x;
    const double lam = lams[blockIdx.x];
    const double *wt = win + ((int64_t)blockIdx.x * 2) * SWIN, *wb = wt + SWIN;
    const SweepSlots sl = sweep_slots();
    for (int e = tid; e < SW * SWS + 2; e += 256) { L.W[e] = 0.0; L.dW[e] = 0.0; }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
        const int a = e >> 5, b = e & 31;
        if (b <= a && a < mb) {
            const int eb = (mb - 1 - b) * NB + (mb - 1 - a);      // the other direction counts the block's rows backwards
            const double orig = ab[(int64_t)(mid + b) * (NB + 1) + (a - b)] + (a == b ? lam : 0.0);
            L.W[a * SWS + (a - b)] = (wt[e] + wb[eb]) - orig;
            if (DERIV) L.dW[a * SWS + (a - b)] = (wt[NB * NB + e] + wb[NB * NB + eb]) - (a == b ? 1.0 : 0.0);
        }
    }
    if (tid < mb) {
        L.W[tid * SWS + SYO] = (wt[2 * NB * NB + tid] + wb[2 * NB * NB + (mb - 1 - tid)]) - g[mid + tid];
        if (DERIV) L.dW[tid * SWS + SYO] = wt[2 * NB * NB + NB + tid] + wb[2 * NB * NB + NB + (mb - 1 - tid)];
    }
    __syncthreads();
    double neg = 0.0, tr = 0.0, q2 = 0.0;
    for (int jj = 0; jj < mb; ++jj) {
        sweep_step<DERIV>(L, sl, jj, neg, tr, q2, nullptr, nullptr);
        lds_barrier();
    }
    if (tid == 128) {
        const double *r0 = res + (int64_t)blockIdx.x * 8, *r1 = r0 + 4;
        out[blockIdx.x * 4 + 0] = neg + r0[0] + r1[0];
        out[blockIdx.x * 4 + 1] = tr + r0[1] + r1[1];
        out[blockIdx.x * 4 + 2] = q2 + r0[2] + r1[2];
    }
}

// ------------------------------------------------------------------------------------ the sweep, one WAVE per chain --
// The kernel above spends ~1 000 cycles per column: four waves share a chain, so every column ends in an s_barrier behind
// the slowest wave's LDS round trips.  Here a chain is ONE wave (LDS operations of a wave complete in order: no barrier at
// all, a column costs what its ~35 LDS and ~70 FP64 instructions cost) and the window is stored by COLUMNS, unrotated:
//       Wc[c * 36 + d]  = entry (row c + d, column c), d = 0 .. 32;    [33] = 0 (the band ends);    [34] = y_c
// so the pivot column is contiguous, every address a lane uses moves by one column (36 entries) per step -- a
// trip of eight steps is straight-line code whose LDS instructions differ in their immediate offsets only -- and a chunk
// of the band arrives as a linear copy (b32_pack34_kernel lays the band out as 34-entry columns, once per direction).
// The step subtracts p_i l_k from entry (jj + i, jj + k), 0 <= k <= i <= 32, i >= 1 (k = 0: the right-hand side, "l_0" =
// y_jj / d): 560 entries in 53 units of 4 rows x 3 columns (rows 4 A + 1 .. 4 A + 4, columns 3 b .. 3 b + 2, clipped by
// k <= i), one unit per lane: 4 + 3 pivot-column reads feed 12 entries.  With DERIV an entry is the pair (value, d/dlambda)
// and moves as one 16-byte LDS access.  96 columns are resident; after 64 steps the 32 live ones move to the front and the
// next 64 (in registers since the chunk began) are parked behind them.
constexpr int S1_ST = 36, S1_NC = 96, S1_CH = 64, S1_U = 8, S1_PF = (S1_CH * (NB + 2)) / 64;
constexpr int S1_ZO = NB + 1, S1_YO = NB + 2;      // per column: [33] stays zero, [34] = y
constexpr int S1_IDLE = 64 + S1_U * S1_ST;         // a word per lane for the slots outside the triangle, reached with the steps' immediate offsets
template <bool DERIV> struct S1Elem { typedef double T; };
template <> struct S1Elem<true> { typedef d2v T; };
template <class E> __device__ __forceinline__ E s1_get(unsigned a) { return *(const __attribute__((address_space(3))) E *)(uintptr_t)a; }
template <class E> __device__ __forceinline__ void s1_put(unsigned a, E v) { *(__attribute__((address_space(3))) E *)(uintptr_t)a = v; }
__device__ __forceinline__ double s1_val(double e) { return e; }
__device__ __forceinline__ double s1_val(d2v e) { return e.x; }
__device__ __forceinline__ double s1_der(double) { return 0.0; }
__device__ __forceinline__ double s1_der(d2v e) { return e.y; }
__device__ __forceinline__ void s1_make(double &e, double v, double) { e = v; }
__device__ __forceinline__ void s1_make(d2v &e, double v, double dv) { e.x = v; e.y = dv; }

// layout the sweeps read: column c of direction 0 is B[c .. c + 32][c], of direction 1 (the matrix mirrored) B[m-1-c][m-1-c-d]
__global__ void b32_pack34_kernel(const double *__restrict__ ab, const double *__restrict__ g, int m, double *__restrict__ abF,
                                  double *__restrict__ abR) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * (NB + 2)) return;
    const int c = e / (NB + 2), d = e - c * (NB + 2);
    if (d <= NB) {
        abF[e] = ab[(int64_t)c * (NB + 1) + d];
        const int cc = m - 1 - c - d;
        abR[e] = cc >= 0 ? ab[(int64_t)cc * (NB + 1) + d] : 0.0;
    } else {
        abF[e] = g[c];
        abR[e] = g[m - 1 - c];
    }
}

template <bool DERIV, bool STORE>
__global__ __launch_bounds__(64) void b32_sweep1_kernel(const double *__restrict__ abF, const double *__restrict__ abR, int m, int n0,
                                                        int n1, const double *__restrict__ lams, double *__restrict__ win,
                                                        double *__restrict__ res, double *__restrict__ Lbuf,
                                                        double *__restrict__ ybuf) {
    typedef typename S1Elem<DERIV>::T E;
    constexpr unsigned ES = sizeof(E), CS = S1_ST * ES;      // bytes per entry, per column
    __shared__ __attribute__((aligned(16))) E Wc[S1_NC * S1_ST + S1_IDLE];
    const int dir = blockIdx.y, ncols = dir == 0 ? n0 : n1, lane = threadIdx.x;
    const double *__restrict__ src = dir == 0 ? abF : abR;
    const double lam = lams[blockIdx.x];
    const int last_row = STORE ? m - 1 : min(m - 1, ncols + NB - 1);      // rows the sweep needs
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) E *)Wc;
    // this lane's unit: rows i0 .. i0 + 3, columns k0 .. k0 + 2 (column 0 = the right-hand side)
    int ua = 7;
#pragma unroll
    for (int a = 7; a >= 1; --a) {
        const int pre = a == 1 ? 2 : a == 2 ? 5 : a == 3 ? 10 : a == 4 ? 16 : a == 5 ? 23 : a == 6 ? 32 : 42;
        if (lane < pre) ua = a - 1;
    }
    const int upre = ua == 0 ? 0 : ua == 1 ? 2 : ua == 2 ? 5 : ua == 3 ? 10 : ua == 4 ? 16 : ua == 5 ? 23 : ua == 6 ? 32 : 42;
    const bool idle = lane >= 53;
    const int ub = idle ? 0 : lane - upre, i0 = 4 * ua + 1, k0 = 3 * ub;
    // byte addresses for the column at the head of the window; a_pk: entry k of the pivot column (k = 0: y), a_pk2: entry
    // k + 1 (k = 0: y) -- what "column k" of a TWO-column step reads from the first of its columns
    // Slots outside the triangle (and the eleven idle lanes) read and write a word of their OWN behind the window, which
    // does not move: forty lanes storing to one dummy address serialise (42 cycles per ds_write_b64 instead of 6).
    unsigned a_piv = base, a_pi = base + (unsigned)i0 * ES, a_pk[3], a_pk2[3], a_w[4][3];
    bool on[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int k = k0 + c;
        a_pk[c] = base + (unsigned)(k == 0 ? S1_YO : k) * ES;
        a_pk2[c] = base + (unsigned)(k == 0 ? S1_YO : k + 1) * ES;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + r;
            on[r][c] = !idle && k <= i;
            a_w[r][c] = base + (unsigned)(!on[r][c] ? S1_NC * S1_ST + lane : k == 0 ? i * S1_ST + S1_YO : k * S1_ST + (i - k)) * ES;
        }
    }
    auto advance = [&](unsigned bytes) {
        a_piv += bytes; a_pi += bytes;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a_pk[c] += bytes; a_pk2[c] += bytes;
#pragma unroll
            for (int r = 0; r < 4; ++r) a_w[r][c] += on[r][c] ? bytes : 0u;
        }
    };
    // value of element e of the linear stream of columns c_lo, c_lo + 1, .. (34 entries each: 33 of the band, then g)
    auto fetch = [&](int c_lo, int e) -> double {
        const int c = c_lo + e / (NB + 2), d = e % (NB + 2);
        const bool ok = d <= NB ? c + d <= last_row : c <= last_row;
        return ok ? src[(int64_t)c_lo * (NB + 2) + e] : 0.0;
    };
    auto park = [&](int slot_col, int c_lo, int e, double v) {      // into LDS column slot_col + e / 34
        const int cl = e / (NB + 2), d = e % (NB + 2);
        const bool diag = d == 0 && c_lo + cl <= last_row;
        E x;
        s1_make(x, v + (diag ? lam : 0.0), diag ? 1.0 : 0.0);
        Wc[(slot_col + cl) * S1_ST + (d <= NB ? d : S1_YO)] = x;
    };
    {
        E z;
        s1_make(z, 0.0, 0.0);
        for (int e = lane; e < S1_NC * S1_ST + S1_IDLE; e += 64) Wc[e] = z;
        for (int e = lane; e < S1_NC * (NB + 2); e += 64) park(0, 0, e, fetch(0, e));
    }
    double neg = 0.0, tr = 0.0, q2 = 0.0;
    int jj = 0;
    const uint64_t t_cyc = __builtin_readcyclecounter(), t_wall = wall_clock64();
    auto book = [&](double d, double dd, double inv, double y, double dy) {
        if (d < 0.0) neg += 1.0;
        if (DERIV) {
            tr += dd * inv;
            q2 -= (2.0 * y * dy * d - y * y * dd) * (inv * inv);
        }
    };
    // one column
    auto step = [&](auto tag) {
        constexpr unsigned off = (unsigned)decltype(tag)::value * CS;
        const E pv = s1_get<E>(a_piv + off), yv = s1_get<E>(a_piv + off + S1_YO * ES);
        E pi[4], pk[3], w[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) pi[r] = s1_get<E>(a_pi + off + r * ES);
#pragma unroll
        for (int c = 0; c < 3; ++c) pk[c] = s1_get<E>(a_pk[c] + off);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) w[r][c] = s1_get<E>(a_w[r][c] + off);
        double d = s1_val(pv);
        const double dd = s1_der(pv);
        if (d == 0.0) d = -1e-300;
        const double inv = frcp(d);
        double lk[3], dlk[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lk[c] = s1_val(pk[c]) * inv;
            dlk[c] = DERIV ? fma(-lk[c], dd, s1_der(pk[c])) * inv : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                E x;
                s1_make(x, fma(-s1_val(pi[r]), lk[c], s1_val(w[r][c])),
                        DERIV ? fma(-s1_der(pi[r]), lk[c], fma(-s1_val(pi[r]), dlk[c], s1_der(w[r][c]))) : 0.0);
                s1_put<E>(a_w[r][c] + off, x);
            }
        book(d, dd, inv, s1_val(yv), s1_der(yv));
        if (STORE) {
            const int64_t col = (int64_t)(dir == 0 ? jj : n0 + jj) + (int64_t)decltype(tag)::value;      // dir 1's columns follow dir 0's in the buffers
            double *Lcol = Lbuf + (col + (int64_t)blockIdx.x * (n0 + n1)) * (NB + 1);
            if (!idle && ub == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Lcol[i0 + r] = s1_val(pi[r]) * inv;
            }
            if (lane == 0) { Lcol[0] = d; ybuf[col + (int64_t)blockIdx.x * (n0 + n1)] = s1_val(yv); }
        }
    };
    // two columns (s, s + 1) in one pass over the window: with p = column s and q = column s + 1 as they stand,
    //       l_1 = p_1 / d,   d' = q_1 - p_1 l_1,   p'_i = q_i - p_i l_1   (column s + 1 after the first elimination),
    //       entry (s + i, s + k) -= p_i (p_k / d) + p'_i (p'_k / d'),   2 <= k <= i <= 33   (p_33 = 0: the band ends)
    // -- every lane redoes the 2 x 2 pivot block and the 4 + 3 entries of p' it needs; the window entries make ONE round trip
    // through the LDS for two columns.  Unit (rows, columns) (i', k') of the one-column step stand for (i' + 1, k' + 1) here.
    auto step2 = [&](auto tag) {
        constexpr unsigned off = (unsigned)decltype(tag)::value * CS;
        const E pv = s1_get<E>(a_piv + off), p1 = s1_get<E>(a_piv + off + ES), qv = s1_get<E>(a_piv + off + CS);
        const E y0 = s1_get<E>(a_piv + off + S1_YO * ES), y1 = s1_get<E>(a_piv + off + CS + S1_YO * ES);
        E pi[4], qi[4], pk[3], qk[3], w[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) { pi[r] = s1_get<E>(a_pi + off + (r + 1) * ES); qi[r] = s1_get<E>(a_pi + off + CS + r * ES); }
#pragma unroll
        for (int c = 0; c < 3; ++c) { pk[c] = s1_get<E>(a_pk2[c] + off); qk[c] = s1_get<E>(a_pk[c] + off + CS); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) w[r][c] = s1_get<E>(a_w[r][c] + off + CS);
        double d = s1_val(pv);
        const double dd = s1_der(pv);
        if (d == 0.0) d = -1e-300;
        const double inv = frcp(d);
        const double l1 = s1_val(p1) * inv, dl1 = DERIV ? fma(-l1, dd, s1_der(p1)) * inv : 0.0;
        double d2 = fma(-s1_val(p1), l1, s1_val(qv));
        const double dd2 = DERIV ? fma(-s1_der(p1), l1, fma(-s1_val(p1), dl1, s1_der(qv))) : 0.0;
        if (d2 == 0.0) d2 = -1e-300;
        const double inv2 = frcp(d2);
        double pp[4], dpp[4], lk[3], dlk[3], lk2[3], dlk2[3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pp[r] = fma(-s1_val(pi[r]), l1, s1_val(qi[r]));
            dpp[r] = DERIV ? fma(-s1_der(pi[r]), l1, fma(-s1_val(pi[r]), dl1, s1_der(qi[r]))) : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lk[c] = s1_val(pk[c]) * inv;
            dlk[c] = DERIV ? fma(-lk[c], dd, s1_der(pk[c])) * inv : 0.0;
            const double pkp = fma(-s1_val(pk[c]), l1, s1_val(qk[c]));
            const double dpkp = DERIV ? fma(-s1_der(pk[c]), l1, fma(-s1_val(pk[c]), dl1, s1_der(qk[c]))) : 0.0;
            lk2[c] = pkp * inv2;
            dlk2[c] = DERIV ? fma(-lk2[c], dd2, dpkp) * inv2 : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                E x;
                s1_make(x, fma(-pp[r], lk2[c], fma(-s1_val(pi[r]), lk[c], s1_val(w[r][c]))),
                        DERIV ? fma(-dpp[r], lk2[c], fma(-pp[r], dlk2[c], fma(-s1_der(pi[r]), lk[c], fma(-s1_val(pi[r]), dlk[c], s1_der(w[r][c])))))
                              : 0.0);
                s1_put<E>(a_w[r][c] + off + CS, x);
            }
        // the right-hand side of row s + 1 as the first elimination leaves it (the formulas of the entries above, k = 0)
        const double ly = s1_val(y0) * inv, dly = DERIV ? fma(-ly, dd, s1_der(y0)) * inv : 0.0;
        const double y1p = fma(-s1_val(p1), ly, s1_val(y1));
        const double dy1p = DERIV ? fma(-s1_der(p1), ly, fma(-s1_val(p1), dly, s1_der(y1))) : 0.0;
        book(d, dd, inv, s1_val(y0), s1_der(y0));
        book(d2, dd2, inv2, y1p, dy1p);
        if (STORE) {
            const int64_t col = (int64_t)(dir == 0 ? jj : n0 + jj) + (int64_t)decltype(tag)::value;
            double *La = Lbuf + (col + (int64_t)blockIdx.x * (n0 + n1)) * (NB + 1), *Lb = La + (NB + 1);
            if (!idle && ub == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (i0 + r + 1 <= NB) La[i0 + r + 1] = s1_val(pi[r]) * inv;
                    Lb[i0 + r] = pp[r] * inv2;
                }
            }
            if (lane == 0) {
                double *yc = ybuf + col + (int64_t)blockIdx.x * (n0 + n1);
                La[0] = d; La[1] = l1; Lb[0] = d2; yc[0] = s1_val(y0); yc[1] = y1p;
            }
        }
    };
    for (int j0 = 0; j0 < ncols; j0 += S1_CH) {
        const bool more = j0 + S1_CH < ncols;
        double pf[S1_PF];
        if (more) {
#pragma unroll
            for (int q = 0; q < S1_PF; ++q) pf[q] = fetch(j0 + S1_NC, lane + 64 * q);
        }
        const int nst = min(S1_CH, ncols - j0);
        int s = 0;
#pragma unroll 1
        for (; s + S1_U <= nst; s += S1_U) {
            step2(std::integral_constant<int, 0>{}); step2(std::integral_constant<int, 2>{});
            step2(std::integral_constant<int, 4>{}); step2(std::integral_constant<int, 6>{});
            advance(S1_U * CS);
            jj += S1_U;
        }
#pragma unroll 1
        for (; s + 2 <= nst; s += 2) {
            step2(std::integral_constant<int, 0>{});
            advance(2 * CS);
            jj += 2;
        }
        if (s < nst) {
            step(std::integral_constant<int, 0>{});
            advance(CS);
            ++jj;
        }
        if (more) {
            for (int e = lane; e < NB * S1_ST; e += 64) Wc[e] = Wc[S1_CH * S1_ST + e];
#pragma unroll
            for (int q = 0; q < S1_PF; ++q) park(NB, j0 + S1_NC, lane + 64 * q, pf[q]);
            advance(0u - (unsigned)S1_CH * CS);
        }
    }
    // the block behind the eliminated columns: logical rows / columns ncols .. ncols + 31
    const int o = ncols == 0 ? 0 : ncols - ((ncols - 1) / S1_CH) * S1_CH;
    double *wout = win + ((int64_t)blockIdx.x * 2 + dir) * SWIN;
    for (int e = lane; e < NB * NB; e += 64) {
        const int a = e >> 5, b = e & 31;
        if (b <= a) {
            const E x = Wc[(o + b) * S1_ST + (a - b)];
            wout[e] = s1_val(x);
            wout[NB * NB + e] = s1_der(x);
        }
    }
    if (lane < NB) {
        const E x = Wc[(o + lane) * S1_ST + S1_YO];
        wout[2 * NB * NB + lane] = s1_val(x);
        wout[2 * NB * NB + NB + lane] = s1_der(x);
    }
    if (lane == 0) {
        double *r3 = res + ((int64_t)blockIdx.x * 2 + dir) * 4;
        r3[0] = neg; r3[1] = tr; r3[2] = q2;
        // diagnostic (MHS_FIT_TIMING prints it): shader-clock cycles of direction 0, 100 MHz ticks of direction 1
        r3[3] = dir == 0 ? (double)(__builtin_readcyclecounter() - t_cyc) : (double)(wall_clock64() - t_wall);
    }
}
static_assert(S1_ZO == NB + 1 && S1_YO < S1_ST, "a two-column step reads entry 33 of its first column as zero");
static_assert(S1_PF * 64 == S1_CH * (NB + 2), "a chunk of new columns is a whole number of loads per lane");

// ================================================================================================ host side ==
int band32_npanels(int m) {
    int np = 0;
    for (int c = 0; m - c - NB >= 2; c += NB) ++np;
    return np;
}
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t band32_layout(Band32Ws *w, char *base, int m, int64_t n) {
    size_t off = 0;
    auto take = [&](size_t doubles) -> double * {
        off = al256(off);
        double *p = base ? (double *)(base + off) : nullptr;
        off += doubles * sizeof(double);
        return p;
    };
    const size_t vs = (size_t)n, np = (size_t)std::max(1, band32_npanels(m));
    Band32Ws d;
    d.Gp1 = take((size_t)B32_MAXBLK * NB * NB);
    d.Gp2 = take((size_t)B32_MAXBLK * NB * NB);
    d.R1 = take(NB * NB);
    d.Qtop = take(NB * NB);
    d.aux = take(AUX_SIZE + 32);
    for (int k = 0; k < 2; ++k) { d.Zc[k] = take(2 * NB * vs + 16); d.Vr[k] = take(((size_t)m + 256) * NB); }
    d.Yp = take((size_t)SYMM_MAXSPLIT * NB * vs);
    d.Wh = take(((size_t)m + 256) * NB);
    d.Mp = take((size_t)B32_MAXBLK * NB * NB);
    d.sgp = take((size_t)B32_MAXBLK * NB);
    d.Tall = take(np * PREC);
    d.ab = take((size_t)m * (NB + 1) + 64);
    d.abF = take((size_t)m * (NB + 2) + 64);
    d.abR = take((size_t)m * (NB + 2) + 64);
    d.win = take((size_t)B32_MAXLAM * 2 * SWIN);
    d.res = take((size_t)B32_MAXLAM * 8);
    d.lamd = take((size_t)B32_MAXLAM);
    d.out = take((size_t)B32_MAXLAM * 4);
    d.Lbuf = take((size_t)m * (NB + 1) + 64);
    d.ybuf = take((size_t)m + 64);
    d.btpart = take((size_t)2 * B32_BT_MAXBLK * NB);
    off = al256(off);
    d.flags = base ? (int *)(base + off) : nullptr;
    off += 256;
    if (w) *w = d;
    return off;
}
size_t band32_workspace_bytes(int m, int64_t n) { return band32_layout(nullptr, nullptr, m, n); }
void band32_carve(Band32Ws &w, char *base, int m, int64_t n) { (void)band32_layout(&w, base, m, n); }

int band32_pinned(FitLane &L, double **out) {
    if (!L.pinned) MHS_HIP(hipHostMalloc((void **)&L.pinned, sizeof(double) * (size_t)B32_MAXLAM * 8, hipHostMallocDefault));
    *out = L.pinned;
    return MHS_OK;
}

// chunks of 256 rows per block so that a panel has at most MAXPART row blocks
static inline void row_blocks(int t, int *cpb, int *nblk) {
    const int nch = (t + CHR - 1) / CHR;
    *cpb = (nch + B32_MAXBLK - 1) / B32_MAXBLK;
    *nblk = (nch + *cpb - 1) / *cpb;
}

int band32_reduce(FitLane &L, hipStream_t s, hipStream_t s2, double *A, int64_t ld, int m, int64_t vs, double *g_dev, Band32Ws &ws,
                  int *breakdown) {
    const int npanels = band32_npanels(m), off0 = 3;
    std::vector<hipEvent_t> &pool = L.pool;
    while ((int)pool.size() < 2 * npanels + 2) {
        hipEvent_t e;
        MHS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        pool.push_back(e);
    }
    // per call: the attribute is per device, and fits run on several lanes (host threads) and device slots
    MHS_HIP(hipFuncSetAttribute((const void *)b32_w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B32_W_LDS));
    MHS_HIP(hipMemsetAsync(ws.flags, 0, sizeof(int), s));
    hipEvent_t pending_rest = nullptr;
    bool have_gram = false;      // the previous panel's strip kernel left this panel's partial Gram matrices in ws.Gp1
    for (int p = 0; p < npanels; ++p) {
        const int c = p * NB, t = m - c - NB, c0 = off0 + c, r0 = off0 + c + NB;
        double *Zc = ws.Zc[p & 1], *Vr = ws.Vr[p & 1], *Tp = ws.Tall + (size_t)p * PREC, *gp = g_dev + c + NB;
        int cpb, nblk;
        row_blocks(t, &cpb, &nblk);
        int nsg = nblk;
        if (t >= 64) {
            if (!have_gram) hipLaunchKernelGGL(b32_gram_kernel, dim3(nblk), dim3(256), 0, s, A, ld, c0, r0, t, cpb, ws.Gp1);
            hipLaunchKernelGGL(b32_cholqr1_kernel, dim3(nblk), dim3(256), 0, s, A, ld, c0, r0, t, cpb, ws.Gp1, nblk, ws.Gp2, ws.R1, ws.Qtop, ws.flags);
            hipLaunchKernelGGL(b32_cholqr2_kernel, dim3(nblk), dim3(256), 0, s, A, ld, c0, r0, t, cpb, ws.Gp2, nblk, ws.Qtop, Zc, vs, Vr, ws.aux, Tp,
                               gp, ws.sgp, ws.flags);
        } else {
            hipLaunchKernelGGL(b32_panel_small_kernel, dim3(1), dim3(256), 0, s, A, ld, c0, r0, t, Zc, vs, Vr, Tp, gp, ws.sgp);
            nsg = 1; cpb = 1; nblk = 1;
        }
        // Y = A22 V: the whole trailing matrix as the previous panel's update left it
        if (pending_rest) { MHS_HIP(hipStreamWaitEvent(s, pending_rest, 0)); pending_rest = nullptr; }
        const int nrb = (t + 63) / 64;
        int nsplit = std::max(1, std::min(SYMM_MAXSPLIT, (512 + nrb / 2) / nrb));
        int wsplit = (((t + nsplit - 1) / nsplit) + 15) & ~15;
        nsplit = (t + wsplit - 1) / wsplit;
        hipLaunchKernelGGL(b32_symm_kernel, dim3(nrb + 1, nsplit), dim3(256), 0, s, A, ld, r0, t, Vr, ws.Yp, vs, wsplit, t >= 64 ? 1 : 0, ws.aux, ws.R1, Tp, ws.flags);
        hipLaunchKernelGGL(b32_w_kernel, dim3(nblk), dim3(256), B32_W_LDS, s, t, cpb, nsplit, ws.Yp, vs, Vr, Tp, ws.sgp, nsg, gp, ws.Wh, ws.Mp);
        hipLaunchKernelGGL(b32_wfin_kernel, dim3(nblk), dim3(256), 0, s, t, cpb, ws.Mp, nblk, Tp, Vr, ws.Wh, Zc + (int64_t)NB * vs, vs);
        // the rest of A22 on the second stream (it needs W, nothing of the strip), the strip in front of the next panel
        const int nt = (t + RK_T - 1) / RK_T;
        if (t > NB) {
            hipEvent_t ev_w = pool[2 * p], ev_rest = pool[2 * p + 1];
            MHS_HIP(hipEventRecord(ev_w, s));
            MHS_HIP(hipStreamWaitEvent(s2, ev_w, 0));
            hipLaunchKernelGGL(b32_rankk_kernel, dim3(nt * (nt + 1) / 2), dim3(256), 0, s2, A, ld, r0, t, Zc, vs, nt);
            MHS_HIP(hipEventRecord(ev_rest, s2));
            pending_rest = ev_rest;
        }
        have_gram = p + 1 < npanels && t - NB >= 64;      // the next panel takes the Cholesky route: its Gram matrix comes with the strip
        if (have_gram) {
            int cpb2, nblk2;
            row_blocks(t - NB, &cpb2, &nblk2);
            hipLaunchKernelGGL(b32_stripgram_kernel, dim3(nblk2), dim3(256), 0, s, A, ld, r0, t, Zc, vs, cpb2, ws.Gp1);
        } else
            hipLaunchKernelGGL(b32_strip_kernel, dim3((t + 63) / 64), dim3(256), 0, s, A, ld, r0, t, Zc, vs);
    }
    if (pending_rest) MHS_HIP(hipStreamWaitEvent(s, pending_rest, 0));
    hipLaunchKernelGGL(b32_extract_kernel, dim3((unsigned)((m * (NB + 1) + 255) / 256)), dim3(256), 0, s, A, ld, off0, m, ws.ab);
    MHS_HIP(hipGetLastError());
    int h_flags = 0;
    MHS_HIP(hipMemcpyAsync(&h_flags, ws.flags, sizeof(int), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    *breakdown = h_flags != 0;
#ifdef B32_PROF
    {
        unsigned long long tab[8][16];
        MHS_HIP(hipMemcpyFromSymbol(tab, HIP_SYMBOL(b32_prof_tab), sizeof(tab)));
        static const char *names[7] = {"gram", "cholqr1", "cholqr2", "symm", "w", "wfin", "rankk"};
        for (int k = 0; k < 7; ++k) {
            fprintf(stderr, "[b32 prof m=%d] %-8s", m, names[k]);
            for (int q = 0; q < 8; ++q) fprintf(stderr, " %8.2f", (double)tab[k][q] / 2400.0 / npanels);
            fprintf(stderr, "   us per panel (block 0)\n");
        }
        memset(tab, 0, sizeof(tab));
        MHS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(b32_prof_tab), tab, sizeof(tab)));
    }
#endif
    return MHS_OK;
}

int band32_qt(hipStream_t s, const double *A, int64_t ld, int m, const double *Tall, double *g_dev, double *sgp) {
    const int npanels = band32_npanels(m), off0 = 3;
    for (int p = 0; p < npanels; ++p) {
        const int c = p * NB, t = m - c - NB, c0 = off0 + c, r0 = off0 + c + NB;
        int cpb, nblk;
        row_blocks(t, &cpb, &nblk);
        if (t < 64) { cpb = 1; nblk = 1; }
        hipLaunchKernelGGL(b32_qt_part_kernel, dim3(nblk), dim3(256), 0, s, A, ld, c0, r0, t, cpb, Tall + (size_t)p * PREC, g_dev + c + NB, sgp);
        hipLaunchKernelGGL(b32_qt_apply_kernel, dim3((t + CHR - 1) / CHR), dim3(256), 0, s, A, ld, c0, r0, t, Tall + (size_t)p * PREC, sgp, nblk,
                           g_dev + c + NB);
    }
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

int band32_backtransform(hipStream_t s, const double *A, int64_t ld, int m, const double *Tall, double *r_dev, double *btpart) {
    const int npanels = band32_npanels(m);
    const unsigned nblk = (unsigned)((m + CHR - 1) / CHR);
    if (nblk > (unsigned)B32_BT_MAXBLK) { set_error("band32_backtransform: order %d too large", m); return MHS_ERR_INVALID; }
    for (int p = npanels; p >= 0; --p)
        hipLaunchKernelGGL(b32_bt_step_kernel, dim3(nblk), dim3(256), 0, s, A, ld, 3, m, p, npanels, Tall, r_dev, btpart);
    MHS_HIP(hipGetLastError());
    return MHS_OK;
}

// -------------------------------------------------------------------------------------------------- the search --
static inline void split_mid(int m, int *mid, int *mb, int *n0, int *n1) {
    *mb = std::min(NB, m);
    *mid = std::max(0, (m - NB) / 2);
    *n0 = *mid;
    *n1 = m - *mid - *mb;
}

// the band as 34-entry columns, one copy per direction (b32_sweep1_kernel reads them as linear streams)
int Band32Search::pack() {
    if (packed) return MHS_OK;
    hipLaunchKernelGGL(b32_pack34_kernel, dim3((unsigned)((m * (NB + 2) + 255) / 256)), dim3(256), 0, s, ab_dev, g_dev, m, ws->abF, ws->abR);
    MHS_HIP(hipGetLastError());
    packed = true;
    return MHS_OK;
}

// one batch on one stream: lambdas up, the two-directional sweep, the joining kernel, the terms down; `off` = the batch's first
// slot in the per-lambda buffers (two batches may be in flight on two streams)
int Band32Search::enqueue(hipStream_t st, int off, const double *lam, int nl, bool deriv) {
    int mid, mb, n0, n1;
    split_mid(m, &mid, &mb, &n0, &n1);
    double *hl = pin + off, *hr = pin + B32_MAXLAM + (size_t)4 * off;
    memcpy(hl, lam, sizeof(double) * nl);
    double *dl = ws->lamd + off, *dwin = ws->win + (size_t)off * 2 * SWIN, *dres = ws->res + (size_t)off * 8, *dout = ws->out + (size_t)off * 4;
    MHS_HIP(hipMemcpyAsync(dl, hl, sizeof(double) * nl, hipMemcpyHostToDevice, st));
    if (deriv) {
        hipLaunchKernelGGL((b32_sweep1_kernel<true, false>), dim3(nl, 2), dim3(64), 0, st, ws->abF, ws->abR, m, n0, n1, dl, dwin, dres,
                           (double *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL((b32_mid_kernel<true>), dim3(nl), dim3(256), 0, st, ab_dev, g_dev, m, mid, mb, dl, dwin, dres, dout);
    } else {
        hipLaunchKernelGGL((b32_sweep1_kernel<false, false>), dim3(nl, 2), dim3(64), 0, st, ws->abF, ws->abR, m, n0, n1, dl, dwin, dres,
                           (double *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL((b32_mid_kernel<false>), dim3(nl), dim3(256), 0, st, ab_dev, g_dev, m, mid, mb, dl, dwin, dres, dout);
    }
    MHS_HIP(hipGetLastError());
    MHS_HIP(hipMemcpyAsync(hr, dout, sizeof(double) * 4 * nl, hipMemcpyDeviceToHost, st));
    return MHS_OK;
}
void Band32Search::collect(int off, int nl, double *neg, double *tr, double *q2) const {
    const double *hr = pin + B32_MAXLAM + (size_t)4 * off;
    for (int i = 0; i < nl; ++i) {
        if (neg) neg[i] = hr[4 * i];
        if (tr) tr[i] = hr[4 * i + 1];
        if (q2) q2[i] = hr[4 * i + 2];
    }
}
void Band32Search::report(int off, int nl, bool deriv) const {
    if (!getenv("MHS_TIMING")) return;
    int mid, mb, n0, n1;
    split_mid(m, &mid, &mb, &n0, &n1);
    double r8[8];
    if (hipMemcpy(r8, ws->res + (size_t)off * 8, sizeof(r8), hipMemcpyDeviceToHost) != hipSuccess) return;
    fprintf(stderr, "[gcv32 m=%d] sweep of %d + %d columns, %d lambdas%s: %.0f cycles, %.1f us (100 MHz counter)\n", m, n0, n1, nl,
            deriv ? " with derivative" : "", r8[3], r8[7] * 0.01);
}

int Band32Search::eval_batch(const double *lam, int count, bool deriv, double *neg, double *tr, double *q2) {
    if (int rc = pack()) return rc;
    for (int base = 0; base < count; base += B32_MAXLAM) {
        const int nl = std::min(B32_MAXLAM, count - base);
        const auto w0 = std::chrono::steady_clock::now();
        if (int rc = enqueue(s, 0, lam + base, nl, deriv)) return rc;
        MHS_HIP(hipStreamSynchronize(s));
        if (getenv("MHS_TIMING"))
            fprintf(stderr, "[gcv32 m=%d] round of %d lambdas: %.3f ms on the host clock\n", m, nl,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
        report(0, nl, deriv);
        collect(0, nl, neg ? neg + base : nullptr, tr ? tr + base : nullptr, q2 ? q2 + base : nullptr);
        ++rounds;
    }
    return MHS_OK;
}

// Two batches side by side: inertia counts for lamA (on s) and counts + tr M^-1 for lamB (on s_aux; on s when there is no
// second stream).  A batch with derivatives needs 61 KB of LDS a chain -- two chains a compute unit -- so folding the
// speculative bracket's 40 lambdas into a 380-lambda cut turned the whole round into a two-wave launch of the slower kernel.
int Band32Search::eval_pair(const double *lamA, int nA, double *negA, const double *lamB, int nB, double *negB, double *trB) {
    if (int rc = pack()) return rc;
    if (nA > B32_PAIR_SPLIT || nB > B32_MAXLAM - B32_PAIR_SPLIT) { set_error("Band32Search::eval_pair: batch too large"); return MHS_ERR_INVALID; }
    hipStream_t sb = s_aux ? s_aux : s;
    if (int rc = enqueue(s, 0, lamA, nA, false)) return rc;
    if (int rc = enqueue(sb, B32_PAIR_SPLIT, lamB, nB, true)) return rc;
    MHS_HIP(hipStreamSynchronize(s));
    if (sb != s) MHS_HIP(hipStreamSynchronize(sb));
    report(0, nA, false);
    report(B32_PAIR_SPLIT, nB, true);
    collect(0, nA, negA, nullptr, nullptr);
    collect(B32_PAIR_SPLIT, nB, negB, trB, nullptr);
    ++rounds;
    return MHS_OK;
}

void Band32Search::gcv_from_terms(double lam, double tr_inv, double qq, double *gcv, double *tra) const {
    const double rss = lam * lam * qq;
    const double trv = 3.0 + (double)m - lam * tr_inv;
    double mse = rss / (double)n;
    if (N - n > 0) mse += pure_ss / (double)(N - n);
    const double den = 1.0 - trv / (double)n;
    if (gcv) *gcv = den > 0 ? mse / (den * den) : NAN;
    if (tra) *tra = trv;
}

// The two extreme eigenvalues by multi-section on the inertia count, P points per bracket and round (constants: the
// sequence of brackets, hence the last bits of lambda, must not depend on anything but the band).  emax starts from
// [largest column norm, min(Gershgorin, Frobenius)], emin from (dmin 2^-200, dmin = smallest diagonal entry]; a bracket whose
// ends are positive and more than a factor 4 apart is cut geometrically, otherwise linearly.  emin's FIRST cut is uneven:
// 200 of its points cover the top 50 octaves (a matrix whose smallest eigenvalue is below 1e-15 of its smallest diagonal
// entry is singular for every later purpose), 55 the 150 octaves below.  Rounds are what the search costs (a sweep is a
// chain of m / 2 column steps, ~0.6 ms at n = 5 000, whatever the number of lambdas), so
//   * the check of the brackets' ends rides along with the first cut (a failed check -- a matrix that is not positive
//     definite -- repeats the round), and
//   * gcv.Krig's bracket (tr A at emax 4^k and emin / 4^k) is evaluated SPECULATIVELY in the round that starts with both
//     eigenvalues known to 1e-6: its 40 points only decide two integers, k1 and k2, and tr A moves by less than
//     (tr A - 3) or (n - tr A) times the relative error of lambda -- decisions closer than 1e-3 to their threshold (or any
//     other surprise) are redone with the final eigenvalues.
static const int EIG_P[2] = {127, 255};      // a round whose bracket is within 256 x the tolerance takes the smallest of 15, 31, 63, 127, 255 points that finishes it
static const int EIG_PMAX = 255;
static const double EIG_TOL = 3e-10;      // lambda moves by about half the relative error of either end (they only place the grid)
static const double EIG_SPEC = 1e-6;
int Band32Search::find_lambda(int mode, double *lam_out) {
    const bool timing = getenv("MHS_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[gcv32 m=%d] %-24s %8.3f ms (%d rounds so far)\n", m, what, std::chrono::duration<double, std::milli>(now - t_last).count(), rounds);
        t_last = now;
    };
    const int W = NB + 1;
    double dmax = ab_host[0], dmin = ab_host[0], ghi = ab_host[0], fro2 = 0.0, cmax2 = 0.0;
    {   // one pass over the band: entry (j + d, j) counts for row j + d and, mirrored, for row j
        std::vector<double> rs((size_t)m + NB + 1, 0.0), cs((size_t)m + NB + 1, 0.0);
        for (int j = 0; j < m; ++j) {
            const double *col = ab_host + (size_t)j * W;
            double r = 0.0, c2 = 0.0;
            for (int d = 1; d <= NB; ++d) {      // entries past the last row are stored as zeros
                const double v = col[d], a = fabs(v), q = v * v;
                r += a; c2 += q;
                rs[(size_t)j + d] += a; cs[(size_t)j + d] += q;
            }
            rs[j] += r; cs[j] += c2;
            fro2 += 2.0 * c2 + col[0] * col[0];
        }
        for (int j = 0; j < m; ++j) {
            const double a = ab_host[(size_t)j * W];
            dmax = std::max(dmax, a); dmin = std::min(dmin, a); ghi = std::max(ghi, a + rs[j]); cmax2 = std::max(cmax2, cs[j] + a * a);
        }
    }
    if (!(dmin > 0.0)) { set_error("mhs_tps_fit: the projected matrix has a non-positive diagonal entry"); return MHS_ERR_NUMERIC; }
    const double up = std::min(ghi, sqrt(fro2)) * (1.0 + 1e-9) + 1e-300, down = std::max(dmax, sqrt(cmax2) * (1.0 - 1e-9));
    double lo[2] = {down, dmin * ldexp(1.0, -200)}, hi[2] = {up, dmin};
    const int64_t kth[2] = {m - 1, 0};
    bool done[2] = {false, false};
    std::vector<double> xs(2 * EIG_PMAX), cnt(2 * EIG_PMAX);
    bool ends_checked = false, spec_have = false;
    int np[2] = {EIG_P[0], EIG_P[1]};      // points of this round
    double spec_e[2] = {0.0, 0.0}, spec_lam[40], spec_neg[40], spec_tr[40];
    auto relw = [&](int e) { return (hi[e] - lo[e]) / std::max(fabs(lo[e]), fabs(hi[e])); };
    for (int it = 0; it < 80; ++it) {
        int live = 0;
        for (int e = 0; e < 2; ++e) {
            if (done[e]) continue;
            if (ends_checked && hi[e] - lo[e] <= EIG_TOL * std::max(fabs(lo[e]), fabs(hi[e]))) { done[e] = true; continue; }
            int P = EIG_P[e];
            if (ends_checked)
                for (int q = 15; q <= EIG_PMAX; q = 2 * q + 1)
                    if ((hi[e] - lo[e]) / (double)(q + 1) <= 0.5 * EIG_TOL * std::max(fabs(lo[e]), fabs(hi[e]))) { P = q; break; }
            np[e] = P;
            const bool geo = lo[e] > 0.0 && hi[e] > 4.0 * lo[e];
            bool distinct = true;
            for (int i = 0; i < P; ++i) {
                double x;
                if (e == 1 && !ends_checked) {      // the uneven first cut of emin: hi 2^-o, o = 150 (54 - i) / 55 + 50 below point 55, 0.25 (P - i) above
                    const double o = i < 55 ? 50.0 + 150.0 * (double)(55 - i) / 56.0 : 0.25 * (double)(P - i);
                    x = hi[e] * exp2(-o);
                } else {
                    const double f = (double)(i + 1) / (double)(P + 1);
                    x = geo ? lo[e] * exp(f * log(hi[e] / lo[e])) : lo[e] + (hi[e] - lo[e]) * f;
                }
                xs[e * EIG_PMAX + i] = x;
                if (!(x > lo[e]) || !(x < hi[e]) || (i > 0 && !(x > xs[e * EIG_PMAX + i - 1]))) distinct = false;
            }
            if (!distinct) {
                const double midp = 0.5 * (lo[e] + hi[e]);
                if (midp <= lo[e] || midp >= hi[e]) { done[e] = true; continue; }
                for (int i = 0; i < P; ++i) xs[e * EIG_PMAX + i] = midp;
            }
            ++live;
        }
        if (!live) break;
        std::vector<double> lamv;
        std::vector<int> owner;      // >= 0: xs slot; -1 .. -4: the brackets' ends
        for (int e = 0; e < 2; ++e)
            if (!done[e]) for (int i = 0; i < np[e]; ++i) { lamv.push_back(-xs[e * EIG_PMAX + i]); owner.push_back(e * EIG_PMAX + i); }
        if (!ends_checked) {
            const double e4[4] = {-lo[0], -hi[0], -lo[1], -hi[1]};
            for (int q = 0; q < 4; ++q) { lamv.push_back(e4[q]); owner.push_back(-1 - q); }
        }
        const bool spec_now = ends_checked && !spec_have && (done[0] || relw(0) <= EIG_SPEC) && (done[1] || relw(1) <= EIG_SPEC);
        if (spec_now) {
            spec_e[0] = 0.5 * (lo[0] + hi[0]); spec_e[1] = std::max(0.5 * (lo[1] + hi[1]), 1e-300);
            for (int q = 0; q < 20; ++q) { spec_lam[q] = spec_e[0] * pow(4.0, q); spec_lam[20 + q] = spec_e[1] / pow(4.0, q); }
        }
        std::vector<double> cv(lamv.size());
        if (spec_now) {
            if (int rc = eval_pair(lamv.data(), (int)lamv.size(), cv.data(), spec_lam, 40, spec_neg, spec_tr)) return rc;
            spec_have = true;
        } else if (int rc = eval_batch(lamv.data(), (int)lamv.size(), false, cv.data(), nullptr, nullptr)) return rc;
        double c4[4] = {0, 0, 0, 0};
        for (size_t q = 0; q < owner.size(); ++q) {
            if (owner[q] >= 0) cnt[owner[q]] = cv[q];
            else c4[-1 - owner[q]] = cv[q];
        }
        bool redo[2] = {false, false};
        if (!ends_checked) {      // the ends themselves: count(lo) must not exceed k, count(hi) must
            ends_checked = true;
            if (c4[0] > (double)kth[0]) { lo[0] = 0.0; redo[0] = true; }                 // cannot happen for a symmetric matrix (emax >= |T e_j|)
            if (!(c4[1] > (double)kth[0])) { hi[0] = 2.0 * ghi + 1.0; redo[0] = true; }
            if (c4[2] > (double)kth[1]) { done[1] = true; lo[1] = hi[1] = 1e-300; }      // an eigenvalue below every floor: as the legacy route's max(ev, 1e-300)
            else if (!(c4[3] > (double)kth[1])) { done[1] = true; lo[1] = hi[1] = dmin; }   // emin <= every diagonal entry: equality only
        }
        for (int e = 0; e < 2; ++e) {
            if (done[e] || redo[e]) continue;
            double nlo = lo[e], nhi = hi[e];
            for (int i = 0; i < np[e]; ++i) {
                if (cnt[e * EIG_PMAX + i] > (double)kth[e]) { nhi = xs[e * EIG_PMAX + i]; break; }
                nlo = xs[e * EIG_PMAX + i];
            }
            lo[e] = nlo; hi[e] = nhi;
        }
        if (timing)
            fprintf(stderr, "[gcv32 m=%d] cut %d: %d + %d points, relative widths now %.2e (largest) %.2e (smallest)\n", m, it, done[0] ? 0 : np[0],
                    done[1] ? 0 : np[1], relw(0), relw(1));
    }
    const double emax = 0.5 * (lo[0] + hi[0]), emin = std::max(0.5 * (lo[1] + hi[1]), 1e-300);
    lap("extreme eigenvalues");
    // gcv.Krig's bracket: l1 = emax 4^k until trA < 3.05, l2 = emin / 4^k until trA > 0.95 n
    double l1 = emax, l2 = emin;
    bool bracket_done = false;
    if (spec_have && fabs(emax - spec_e[0]) <= 4.0 * EIG_SPEC * emax && fabs(emin - spec_e[1]) <= 4.0 * EIG_SPEC * emin) {
        bool sure = true;
        int k1 = 0, k2 = 0;
        for (int k = 0; k < 20 && sure; ++k) {
            if (spec_neg[k] > 0) { sure = false; break; }
            double tra;
            gcv_from_terms(spec_lam[k], spec_tr[k], 0.0, nullptr, &tra);
            if (!(fabs(tra - 3.05) > 1e-3 * 3.05)) { sure = false; break; }
            if (tra < 3.0 + 0.05) break;
            ++k1;
        }
        for (int k = 0; k < 20 && sure; ++k) {
            if (spec_neg[20 + k] > 0) { sure = false; break; }
            double tra;
            gcv_from_terms(spec_lam[20 + k], spec_tr[20 + k], 0.0, nullptr, &tra);
            if (!(fabs(tra - 0.95 * (double)n) > 1e-3 * 0.95 * (double)n)) { sure = false; break; }
            if (tra > 0.95 * (double)n) break;
            ++k2;
        }
        if (sure) {
            for (int k = 0; k < k1; ++k) l1 *= 4.0;
            for (int k = 0; k < k2; ++k) l2 /= 4.0;
            bracket_done = true;
        }
    }
    if (!bracket_done) {
        double lamb[40], negb[40], trb[40];
        for (int i = 0; i < 20; ++i) { lamb[i] = emax * pow(4.0, i); lamb[20 + i] = emin / pow(4.0, i); }
        if (int rc = eval_batch(lamb, 40, true, negb, trb, nullptr)) return rc;
        for (int k = 0; k < 20; ++k) {
            if (negb[k] > 0) { *lam_out = NAN; return MHS_OK; }
            double tra;
            gcv_from_terms(lamb[k], trb[k], 0.0, nullptr, &tra);
            if (tra < 3.0 + 0.05) break;
            l1 *= 4.0;
        }
        for (int k = 0; k < 20; ++k) {
            if (negb[20 + k] > 0) break;
            double tra;
            gcv_from_terms(lamb[20 + k], trb[20 + k], 0.0, nullptr, &tra);
            if (tra > 0.95 * (double)n) break;
            l2 /= 4.0;
        }
    }
    lap("bracket (40 evals)");
    const int nstep = 200;
    std::vector<double> lamv(nstep), negv(nstep), trv(nstep), q2v(nstep), gcvv(nstep);
    const double la = log(l2), lb = log(l1);
    for (int i = 0; i < nstep; ++i) lamv[i] = exp(la + (lb - la) * (double)i / (double)(nstep - 1));
    if (int rc = eval_batch(lamv.data(), nstep, true, negv.data(), trv.data(), q2v.data())) return rc;
    std::vector<double> grid, gv;
    for (int i = 0; i < nstep; ++i) {
        double gcv = NAN;
        if (!(negv[i] > 0)) gcv_from_terms(lamv[i], trv[i], q2v[i], &gcv, nullptr);
        if (!std::isnan(gcv)) { grid.push_back(lamv[i]); gv.push_back(gcv); }
    }
    lap("grid (200 evals)");
    if (grid.empty()) { *lam_out = NAN; return MHS_OK; }
    size_t il = 0;
    for (size_t i = 1; i < gv.size(); ++i) if (gv[i] < gv[il]) il = i;
    if (il == 0 || il + 1 == gv.size()) { *lam_out = grid[il]; return MHS_OK; }
    // Golden sections, expanded speculatively: the points a search visits depend on the path of comparisons only, so the
    // tree of the next DEPTH steps (2^(DEPTH+1) - 2 points) is evaluated in ONE round and the true path replayed on it --
    // the very iterates of the sequential search (same formulas, same operands), DEPTH steps per round.
    auto fbatch = [&](const std::vector<double> &pts, std::vector<double> &vals, bool is_log) -> int {
        std::vector<double> lv(pts.size()), ng(pts.size()), tv(pts.size()), qv(pts.size());
        for (size_t i = 0; i < pts.size(); ++i) lv[i] = is_log ? exp(pts[i]) : pts[i];
        if (int rc = eval_batch(lv.data(), (int)lv.size(), true, ng.data(), tv.data(), qv.data())) return rc;
        vals.resize(pts.size());
        for (size_t i = 0; i < pts.size(); ++i) {
            double gcv = NAN;
            if (!(ng[i] > 0)) gcv_from_terms(lv[i], tv[i], qv[i], &gcv, nullptr);
            vals[i] = gcv;
        }
        return MHS_OK;
    };
    const int DEPTH = 7;
    if (mode == MHS_GCV_FIELDS) {      // golden.section.search, tol = 0.01 GCVmin, at most 25 steps
        const double r = 0.61803399, con = 1.0 - r, tol = 0.01 * gv[il];
        const double ax = grid[il - 1], bx = grid[il], cx = grid[il + 1];
        struct St { double x0, x1, x2, x3; };
        St st;
        st.x0 = ax; st.x3 = cx;
        if (fabs(cx - bx) > fabs(bx - ax)) { st.x1 = bx; st.x2 = bx + con * (cx - bx); }
        else { st.x2 = bx; st.x1 = bx - con * (bx - ax); }
        double f1 = NAN, f2 = NAN;
        bool have = false;
        int steps = 0;
        bool stop = false;
        while (!stop && steps < 25) {
            const int depth = std::min(DEPTH, 25 - steps);
            // node id in a complete binary tree: root 1, child 2 id (f2 < f1) / 2 id + 1 (else); the point a child adds
            std::vector<St> node((size_t)1 << (depth + 1));
            std::vector<double> pts;
            std::vector<int> slot((size_t)1 << (depth + 1), -1);
            node[1] = st;
            if (!have) { pts.push_back(st.x1); pts.push_back(st.x2); }
            for (int id = 1; id < (1 << depth); ++id) {
                const St &q = node[id];
                St a = q, b = q;
                a.x0 = q.x1; a.x1 = q.x2; a.x2 = r * a.x1 + con * q.x3;
                b.x3 = q.x2; b.x2 = q.x1; b.x1 = r * b.x2 + con * q.x0;
                node[2 * id] = a; node[2 * id + 1] = b;
                slot[2 * id] = (int)pts.size(); pts.push_back(a.x2);
                slot[2 * id + 1] = (int)pts.size(); pts.push_back(b.x1);
            }
            std::vector<double> vals;
            if (int rc = fbatch(pts, vals, false)) return rc;
            if (!have) { f1 = vals[0]; f2 = vals[1]; have = true; }
            int id = 1;
            for (int d = 0; d < depth && !stop; ++d) {
                if (f2 < f1) { id = 2 * id; f1 = f2; f2 = vals[slot[id]]; }
                else { id = 2 * id + 1; f2 = f1; f1 = vals[slot[id]]; }
                ++steps;
                if (fabs(f2 - f1) < tol) stop = true;
            }
            st = node[id];
        }
        lap("golden section");
        *lam_out = f1 < f2 ? st.x1 : st.x2;
        return MHS_OK;
    }
    // converged: golden section on log(lambda) down to 1e-13
    {
        double lo2 = log(grid[il - 1]), hi2 = log(grid[il + 1]);
        const double r = 0.5 * (sqrt(5.0) - 1.0);
        struct St { double lo, hi, x1, x2; };
        St st;
        st.lo = lo2; st.hi = hi2; st.x1 = hi2 - r * (hi2 - lo2); st.x2 = lo2 + r * (hi2 - lo2);
        double f1 = NAN, f2 = NAN;
        bool have = false, stop = false;
        int steps = 0;
        while (!stop && steps < 200) {
            const int depth = std::min(DEPTH, 200 - steps);
            std::vector<St> node((size_t)1 << (depth + 1));
            std::vector<double> pts;
            std::vector<int> slot((size_t)1 << (depth + 1), -1);
            node[1] = st;
            if (!have) { pts.push_back(st.x1); pts.push_back(st.x2); }
            for (int id = 1; id < (1 << depth); ++id) {
                const St &q = node[id];
                St a = q, b = q;
                a.hi = q.x2; a.x2 = q.x1; a.x1 = a.hi - r * (a.hi - a.lo);          // f1 < f2
                b.lo = q.x1; b.x1 = q.x2; b.x2 = b.lo + r * (b.hi - b.lo);          // else
                node[2 * id] = a; node[2 * id + 1] = b;
                slot[2 * id] = (int)pts.size(); pts.push_back(a.x1);
                slot[2 * id + 1] = (int)pts.size(); pts.push_back(b.x2);
            }
            std::vector<double> vals;
            if (int rc = fbatch(pts, vals, true)) return rc;
            if (!have) { f1 = vals[0]; f2 = vals[1]; have = true; }
            int id = 1;
            for (int d = 0; d < depth && !stop; ++d) {
                if (f1 < f2) { id = 2 * id; f2 = f1; f1 = vals[slot[id]]; }
                else { id = 2 * id + 1; f1 = f2; f2 = vals[slot[id]]; }
                ++steps;
                if (fabs(node[id].hi - node[id].lo) < 1e-13) stop = true;
            }
            st = node[id];
        }
        lap("golden section (converged)");
        *lam_out = exp(0.5 * (st.lo + st.hi));
    }
    return MHS_OK;
}

// q = (Bb + lambda I)^-1 g: the twisted sweep once more with the factor stored, then the joint block and the two back
// substitutions on the host (O(33 m) work on 1.3 MB at n = 5 000).
int Band32Search::solve(double lam, double *gcv, double *eff_df, double *q_host) {
    int mid, mb, n0, n1;
    split_mid(m, &mid, &mb, &n0, &n1);
    const int W = NB + 1;
    double *dl = ws->lamd;
    pin[0] = lam;
    MHS_HIP(hipMemcpyAsync(dl, pin, sizeof(double), hipMemcpyHostToDevice, s));
    if (int rc = pack()) return rc;
    hipLaunchKernelGGL((b32_sweep1_kernel<true, true>), dim3(1, 2), dim3(64), 0, s, ws->abF, ws->abR, m, n0, n1, dl, ws->win, ws->res, ws->Lbuf, ws->ybuf);
    MHS_HIP(hipGetLastError());
    std::vector<double> Lh((size_t)(n0 + n1) * W + 1), yh((size_t)(n0 + n1) + 1), wh(2 * SWIN), rh(8);
    MHS_HIP(hipMemcpyAsync(Lh.data(), ws->Lbuf, sizeof(double) * (size_t)(n0 + n1) * W, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipMemcpyAsync(yh.data(), ws->ybuf, sizeof(double) * (size_t)(n0 + n1), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipMemcpyAsync(wh.data(), ws->win, sizeof(double) * 2 * SWIN, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipMemcpyAsync(rh.data(), ws->res, sizeof(double) * 8, hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    // joint block S (mb x mb) with its derivative, right-hand side y
    std::vector<double> S((size_t)mb * mb, 0.0), dS((size_t)mb * mb, 0.0), y(mb), dy(mb);
    const double *wt = wh.data(), *wb = wh.data() + SWIN;
    for (int a = 0; a < mb; ++a) {
        for (int b = 0; b <= a; ++b) {
            const int e = a * NB + b, eb = (mb - 1 - b) * NB + (mb - 1 - a);
            const double orig = ab_host[(size_t)(mid + b) * W + (a - b)] + (a == b ? lam : 0.0);
            S[(size_t)a * mb + b] = (wt[e] + wb[eb]) - orig;
            dS[(size_t)a * mb + b] = (wt[NB * NB + e] + wb[NB * NB + eb]) - (a == b ? 1.0 : 0.0);
        }
        y[a] = (wt[2 * NB * NB + a] + wb[2 * NB * NB + (mb - 1 - a)]) - g_host[mid + a];
        dy[a] = wt[2 * NB * NB + NB + a] + wb[2 * NB * NB + NB + (mb - 1 - a)];
    }
    // LDL' of S with the derivative carried along (the same recurrences as the device sweep), forward substitution
    double neg = rh[0] + rh[4], tr = rh[1] + rh[5], q2 = rh[2] + rh[6];
    std::vector<double> dm(mb), Lm((size_t)mb * mb, 0.0);
    for (int j = 0; j < mb; ++j) {
        double d = S[(size_t)j * mb + j];
        if (d == 0.0) d = -1e-300;
        if (d < 0) neg += 1.0;
        const double dd = dS[(size_t)j * mb + j], inv = 1.0 / d;
        tr += dd * inv;
        q2 -= (2.0 * y[j] * dy[j] * d - y[j] * y[j] * dd) * (inv * inv);
        dm[j] = d;
        for (int i = j + 1; i < mb; ++i) {
            const double pi = S[(size_t)i * mb + j], dpi = dS[(size_t)i * mb + j];
            const double li = pi * inv, dli = (dpi - li * dd) * inv;
            Lm[(size_t)i * mb + j] = li;
            dy[i] -= dli * y[j] + li * dy[j];
            y[i] -= li * y[j];
            for (int k = j + 1; k <= i; ++k) {
                const double pk = S[(size_t)k * mb + j], lk = pk * inv, dlk = (dS[(size_t)k * mb + j] - lk * dd) * inv;
                dS[(size_t)i * mb + k] -= dpi * lk + pi * dlk;
                S[(size_t)i * mb + k] -= pi * lk;
            }
        }
    }
    if (neg > 0) { set_error("mhs_tps_fit: band matrix + lambda I is not positive definite"); return MHS_ERR_NUMERIC; }
    gcv_from_terms(lam, tr, q2, gcv, eff_df);
    // back substitution: the joint block, then outwards on both sides
    std::vector<double> x((size_t)m, 0.0);
    for (int j = mb - 1; j >= 0; --j) {
        double sacc = y[j] / dm[j];
        for (int i = j + 1; i < mb; ++i) sacc -= Lm[(size_t)i * mb + j] * x[mid + i];
        x[mid + j] = sacc;
    }
    for (int j = n0 - 1; j >= 0; --j) {
        const double *c = &Lh[(size_t)j * W];
        double s0 = yh[j] / c[0], s1 = 0.0;
        for (int i = 1; i <= NB; i += 2) { s0 -= c[i] * x[j + i]; s1 -= c[i + 1] * x[j + i + 1 < m ? j + i + 1 : j + i]; }
        x[j] = s0 + s1;
    }
    for (int jj = n1 - 1; jj >= 0; --jj) {
        const double *c = &Lh[(size_t)(n0 + jj) * W];
        const int r = m - 1 - jj;
        double s0 = yh[n0 + jj] / c[0], s1 = 0.0;
        for (int i = 1; i <= NB; i += 2) { s0 -= c[i] * x[r - i]; s1 -= c[i + 1] * x[r - i - 1 >= 0 ? r - i - 1 : r - i]; }
        x[r] = s0 + s1;
    }
    memcpy(q_host, x.data(), sizeof(double) * (size_t)m);
    return MHS_OK;
}

}  // namespace mhs

using namespace mhs;

// ---- test hooks (include/machisplin_hip.h) ------------------------------------------------------------------------
extern "C" int mhs_band32_reduce(const double *B, const double *g, int64_t m64, double *ab, double *gq, const double *r, double *Qr,
                                 int *breakdown) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(B && g && ab && gq && breakdown, "NULL argument");
    MHS_REQUIRE(m64 >= 66 && m64 <= B32_MAX_M, "m must be between 66 and 32 768");
    const int m = (int)m64;
    const int64_t n = m + 3, ld = (n + 15) & ~(int64_t)15;
    FitLane *L = nullptr;
    if (int rc = fit_lane(0, &L)) return rc;
    hipStream_t s = L->s;
    DevBuf<double> dA, dg, dr;
    DevBuf<char> dws;
    MHS_HIP(dA.alloc((size_t)ld * n + 4)); MHS_HIP(dg.alloc((size_t)m + 64)); MHS_HIP(dr.alloc((size_t)m + 64));
    MHS_HIP(dws.alloc(band32_workspace_bytes(m, n)));
    // same placement as the fit: B starts at row / column 3, row 3 of every column on a 16-byte boundary
    double *A = dA.p + 1;
    MHS_HIP(hipMemsetAsync(dA.p, 0, sizeof(double) * ((size_t)ld * n + 4), s));
    MHS_HIP(hipMemcpy2DAsync(A + 3 * ld + 3, sizeof(double) * ld, B, sizeof(double) * m, sizeof(double) * m, (size_t)m, hipMemcpyHostToDevice, s));
    MHS_HIP(hipMemcpyAsync(dg.p, g, sizeof(double) * m, hipMemcpyHostToDevice, s));
    Band32Ws w;
    band32_carve(w, dws.p, m, n);
    if (int rc = band32_reduce(*L, s, L->s2r ? L->s2r : L->s2, A, ld, m, n, dg.p, w, breakdown)) return rc;
    MHS_HIP(hipMemcpyAsync(ab, w.ab, sizeof(double) * (size_t)m * (B32_NB + 1), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipMemcpyAsync(gq, dg.p, sizeof(double) * m, hipMemcpyDeviceToHost, s));
    if (r && Qr) {
        MHS_HIP(hipMemcpyAsync(dr.p, r, sizeof(double) * m, hipMemcpyHostToDevice, s));
        if (int rc = band32_backtransform(s, A, ld, m, w.Tall, dr.p, w.btpart)) return rc;
        MHS_HIP(hipMemcpyAsync(Qr, dr.p, sizeof(double) * m, hipMemcpyDeviceToHost, s));
    }
    MHS_HIP(hipStreamSynchronize(s));
    MHS_HIP(hipStreamSynchronize(L->s2r ? L->s2r : L->s2));
    return MHS_OK;
}

static int hook_search(const double *ab, const double *g, int m, Band32Search &bs, Band32Ws &w, DevBuf<char> &dws, DevBuf<double> &dg) {
    FitLane *L = nullptr;
    if (int rc = fit_lane(0, &L)) return rc;
    const int64_t n = m + 3;
    MHS_HIP(dws.alloc(band32_workspace_bytes(m, n)));
    MHS_HIP(dg.alloc((size_t)m + 64));
    band32_carve(w, dws.p, m, n);
    double *pin = nullptr;
    if (int rc = band32_pinned(*L, &pin)) return rc;
    MHS_HIP(hipMemcpyAsync(w.ab, ab, sizeof(double) * (size_t)m * (B32_NB + 1), hipMemcpyHostToDevice, L->s));
    MHS_HIP(hipMemcpyAsync(dg.p, g, sizeof(double) * m, hipMemcpyHostToDevice, L->s));
    bs.s = L->s; bs.ab_dev = w.ab; bs.g_dev = dg.p; bs.ab_host = ab; bs.g_host = g; bs.m = m; bs.n = n; bs.N = n; bs.pure_ss = 0.0; bs.ws = &w; bs.pin = pin;
    return MHS_OK;
}

extern "C" int mhs_band32_gcv_terms(const double *ab, const double *g, int64_t m, const double *lambda, int n_lambda, int deriv,
                                    double *neg, double *tr_inv, double *gM2g) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(ab && g && lambda && neg && n_lambda >= 1 && m >= 2 && m <= B32_MAX_M, "bad arguments");
    Band32Search bs;
    Band32Ws w;
    DevBuf<char> dws;
    DevBuf<double> dg;
    if (int rc = hook_search(ab, g, (int)m, bs, w, dws, dg)) return rc;
    return bs.eval_batch(lambda, n_lambda, deriv != 0, neg, deriv ? tr_inv : nullptr, deriv ? gM2g : nullptr);
}

extern "C" int mhs_band32_solve(const double *ab, const double *g, int64_t m, double lambda, double *q) {
    if (int rc = require_ready()) return rc;
    MHS_REQUIRE(ab && g && q && m >= 2 && m <= B32_MAX_M, "bad arguments");
    Band32Search bs;
    Band32Ws w;
    DevBuf<char> dws;
    DevBuf<double> dg;
    if (int rc = hook_search(ab, g, (int)m, bs, w, dws, dg)) return rc;
    double gcv, df;
    return bs.solve(lambda, &gcv, &df, q);
}
