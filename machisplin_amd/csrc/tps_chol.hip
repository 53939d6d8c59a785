// Fixed-lambda route of fields::Tps on gfx950: blocked Cholesky of B + lambda I (B = Q2'KQ2, SPD) and the two
// triangular solves, with every O(n^3) flop on v_mfma_f64_16x16x4_f64.  (V73:722, V73:751 when lambda is given;
// also the solve the GCV route finishes with once lambda is known.)
//
// Right-looking, panels of NB = 128 columns, lower triangle, in place, column-major:
//   chol_diag_kernel   one block: L11 = chol(A11) in LDS (16-column sub-panels: 16 x 16 factor in the registers of
//                      one wave, row-parallel triangular solve, MFMA rank-16 update), L11 stored; then L11^-1 in
//                      place (block columns from the right, MFMA products) and stored to the workspace; y_j = L11^-1
//                      b_j -- the forward substitution rides along with the factorisation.
//   chol_trsm_kernel   X = A21 L11^-T as an MFMA GEMM against the explicit inverse (64 rows per block, the result
//                      kept in accumulators and written in place); b[below] -= X y_j.
//   chol_syrk_kernel   A22 -= X X' on 128 x 128 tiles of the lower triangle (4 waves x 64 x 64, K = 128 streamed
//                      through LDS in double-buffered chunks of 16).  Launched twice per panel: first the tiles of
//                      the next panel's block column (the factorisation goes on behind them, look-ahead), then
//                      the rest on the lane's second stream.
//   chol_bsolve_*      x = L^-T y, panels from the last: a 128 x 128 product with L_jj^-T, then one pass over the
//                      panel's 128 rows of L to update every earlier entry of y.
// The matrix is padded to a multiple of NB with an identity block (rows / columns the caller provides beyond m), so
// no kernel carries edge cases.  Algorithmic work: m^3/3 flop (syrk) + m^2 NB (trsm, half of it skipped through
// the triangular structure of L11^-1); bytes: every syrk tile is read and written once per panel -- m^3/(3 NB) x 8 B
// each way -- which at NB = 128 asks 4.8 TB/s of HBM at the full MFMA rate (the tiles live in the 256 MB Infinity
// Cache up to m ~ 5 000).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "devmath.h"
#include "tps_host.h"
#include "tps_chol.h"

namespace mhs {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int CH_NB = 128;
constexpr int CH_LDP = CH_NB + 1;       // LDS leading dimension of the diagonal block
constexpr int CH_SB = 16;               // sub-panel width inside the diagonal block

// A[row][col] (lower) of the padded SPD matrix lives at base + row + col * ld
__global__ void chol_pad_kernel(double *__restrict__ A, int64_t ld, int off, int m, int m_pad, double *__restrict__ rhs) {
    // identity in the padding block, zeros beside it; rhs padding = 0
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int np = m_pad - m;
    const int64_t total = (int64_t)np * m_pad;
    if (e < np) rhs[m + e] = 0.0;
    if (e >= total) return;
    const int r = (int)(e % m_pad), c = m + (int)(e / m_pad);    // column c of the padding, every row r
    double *a = A + (int64_t)off * ld + off;
    a[(int64_t)c * ld + r] = r == c ? 1.0 : 0.0;
    if (r < m) a[(int64_t)r * ld + c] = 0.0;                      // and the mirrored row segment
}

__device__ __forceinline__ double &SS(double *S, int i, int j) { return S[i + j * CH_LDP]; }

constexpr int DG_THREADS = 512;   // 2 waves per SIMD: 256 registers per lane for the unrolled 16 x 16 steps
__global__ __launch_bounds__(DG_THREADS) void chol_diag_kernel(double *__restrict__ A, int64_t ld, int off,
                                                         double *__restrict__ Tinv /* NB x NB, column-major */,
                                                         double *__restrict__ rhs /* b_j in, y_j out */,
                                                         int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *S = (double *)smem;                          // NB x LDP
    double *tmp = S + CH_NB * CH_LDP;                    // (NB - 16) x 16
    double *TD = tmp + (CH_NB - CH_SB) * CH_SB;          // 16 x 17
    double *rdiag = TD + CH_SB * (CH_SB + 1);            // 16
    double *bs = rdiag + CH_SB;                          // NB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    double *a = A + (int64_t)off * ld + off;
    for (int e = tid; e < CH_NB * CH_NB; e += DG_THREADS) {
        const int i = e & (CH_NB - 1), j = e >> 7;
        SS(S, i, j) = i >= j ? a[(int64_t)j * ld + i] : 0.0;
    }
    if (tid < CH_NB) bs[tid] = rhs[tid];
    __syncthreads();
    // ---------------------------------------------------------------- factor --
    for (int jb = 0; jb < CH_NB / CH_SB; ++jb) {
        const int c0 = jb * CH_SB, R = CH_NB - c0 - CH_SB;     // rows below the sub-panel's diagonal block
        if (wave == 0) {
            double d[CH_SB];
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) d[c] = (lane < CH_SB && c <= lane) ? SS(S, c0 + lane, c0 + c) : 0.0;
            bool bad = false;
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) {
                const double piv = lane_value(d[c], c);
                bad |= !(piv > 0.0);
                const double l = sqrt(piv), rl = 1.0 / l;
                d[c] = lane == c ? l : d[c] * rl;
                if (lane == c) rdiag[c] = rl;
#pragma unroll
                for (int cc = c + 1; cc < CH_SB; ++cc) d[cc] -= d[c] * lane_value(d[c], cc);
            }
            if (bad && lane == 0) atomicCAS(info, 0, off + c0 + 1);
            if (lane < CH_SB) {
#pragma unroll
                for (int c = 0; c < CH_SB; ++c) if (c <= lane) SS(S, c0 + lane, c0 + c) = d[c];
            }
        }
        __syncthreads();
        if (tid < R) {      // X = S[below, sub-panel] L_D^-T, one row per thread
            const int i = c0 + CH_SB + tid;
            double x[CH_SB];
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) {
                double s = SS(S, i, c0 + c);
#pragma unroll
                for (int k = 0; k < c; ++k) s -= x[k] * SS(S, c0 + c, c0 + k);
                x[c] = s * rdiag[c];
            }
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) SS(S, i, c0 + c) = x[c];
        }
        __syncthreads();
        {   // S[below, below] -= X X' (lower tiles of 16 x 16, one MFMA chain of 4 per tile)
            const int nT = R / CH_SB, npairs = nT * (nT + 1) / 2;
            for (int p = wave; p < npairs; p += DG_THREADS / 64) {
                int ti = 0, rem = p;
                while (rem > ti) { rem -= ti + 1; ++ti; }
                const int tj = rem;
                const int i0 = c0 + CH_SB + ti * CH_SB, j0 = c0 + CH_SB + tj * CH_SB;
                d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < CH_SB; kk += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(SS(S, j0 + l15, c0 + kk + l4), SS(S, i0 + l15, c0 + kk + l4), acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + l15, j = j0 + l4 + 4 * r;
                    if (j <= i) SS(S, i, j) -= acc[r];
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < CH_NB * CH_NB; e += DG_THREADS) {      // L11 back to the matrix
        const int i = e & (CH_NB - 1), j = e >> 7;
        if (i >= j) a[(int64_t)j * ld + i] = SS(S, i, j);
    }
    // ------------------------------------------------------- L11^-1 in place --
    for (int jb = CH_NB / CH_SB - 1; jb >= 0; --jb) {
        const int c0 = jb * CH_SB, R = CH_NB - c0 - CH_SB;
        if (wave == 0) {      // TD = L_D^-1: lane c < 16 solves column c
            double x[CH_SB];
#pragma unroll
            for (int r = 0; r < CH_SB; ++r) {
                double s = (r == lane) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < r; ++k) s -= SS(S, c0 + r, c0 + k) * x[k];
                x[r] = s / SS(S, c0 + r, c0 + r);
            }
            if (lane < CH_SB) {
#pragma unroll
                for (int r = 0; r < CH_SB; ++r) TD[r * (CH_SB + 1) + lane] = r >= lane ? x[r] : 0.0;
            }
        }
        __syncthreads();
        if (tid < R) {        // W = L[below, block] TD, row by row, in place
            const int i = c0 + CH_SB + tid;
            double l[CH_SB], w[CH_SB];
#pragma unroll
            for (int k = 0; k < CH_SB; ++k) l[k] = SS(S, i, c0 + k);
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) {
                double s = 0.0;
#pragma unroll
                for (int k = c; k < CH_SB; ++k) s = fma(l[k], TD[k * (CH_SB + 1) + c], s);
                w[c] = s;
            }
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) SS(S, i, c0 + c) = w[c];
        }
        __syncthreads();
        {   // tmp = Tinv[below, below] W   (row tiles of 16, k tiles up to the diagonal one)
            const int nT = R / CH_SB;
            for (int ti = wave; ti < nT; ti += DG_THREADS / 64) {
                d4 acc = {0.0, 0.0, 0.0, 0.0};
                const int i0 = c0 + CH_SB + ti * CH_SB;
                for (int kt = 0; kt <= ti; ++kt) {
                    const int k0 = c0 + CH_SB + kt * CH_SB;
#pragma unroll
                    for (int kk = 0; kk < CH_SB; kk += 4)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(SS(S, k0 + kk + l4, c0 + l15), SS(S, i0 + l15, k0 + kk + l4), acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) tmp[(ti * CH_SB + l15) + (l4 + 4 * r) * (CH_NB - CH_SB)] = acc[r];
            }
        }
        __syncthreads();
        for (int e = tid; e < R * CH_SB; e += DG_THREADS) {
            const int i = e % R, c = e / R;
            SS(S, c0 + CH_SB + i, c0 + c) = -tmp[i + c * (CH_NB - CH_SB)];
        }
        if (tid < CH_SB * CH_SB) {
            const int r = tid & 15, c = tid >> 4;
            if (r >= c) SS(S, c0 + r, c0 + c) = TD[r * (CH_SB + 1) + c];
        }
        __syncthreads();
    }
    for (int e = tid; e < CH_NB * CH_NB; e += DG_THREADS) {
        const int i = e & (CH_NB - 1), j = e >> 7;
        Tinv[e] = SS(S, i, j);                          // L11^-1, zeros above the diagonal
        Tinv[CH_NB * CH_NB + e] = SS(S, j, i);          // and its transpose (read by the back substitution)
    }
    if (tid < CH_NB) {                  // y_j = L11^-1 b_j
        double s = 0.0;
        for (int k = 0; k <= tid; ++k) s = fma(SS(S, tid, k), bs[k], s);
        rhs[tid] = s;
    }
}

// X = A21 L11^-T for the t rows below the diagonal block of the panel at `off` (in place), and b[below] -= X y_j.
// Block = 32 rows (t / 32 blocks: the kernel sits on the factorisation's critical path, so it wants every CU even at
// small t); wave w owns rows 16 (w & 1) .. + 15 and the column half (w >> 1) (4 accumulator tiles).  K = 128 is
// streamed in chunks of 16 through two LDS buffers (the next chunk travels global -> registers while this one is
// multiplied).  L11^-1 is lower triangular: products with k > j are skipped.
constexpr int TR_ROWS = 32, TR_KC = 16, TR_SA = TR_ROWS + 16, TR_ST = CH_NB + 16;
__global__ __launch_bounds__(256) void chol_trsm_kernel(double *__restrict__ A, int64_t ld, int off,
                                                        const double *__restrict__ Tinv, double *__restrict__ rhs) {
    __shared__ __attribute__((aligned(16))) double sT[2][TR_KC * TR_ST];   // [k][j]
    __shared__ __attribute__((aligned(16))) double sA[2][TR_KC * TR_SA];   // [k][i]
    __shared__ double ys[CH_NB];
    __shared__ double part[2][TR_ROWS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int rh = wave & 1, ch = wave >> 1;
    double *a21 = A + (int64_t)off * ld + off + CH_NB + (int64_t)blockIdx.x * TR_ROWS;   // rows of this block, column 0 of the panel
    if (tid < CH_NB) ys[tid] = rhs[tid];   // the caller passes rhs + j: y_j at [0, NB), the entries below the panel after it
    d4 acc[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) acc[mb] = (d4){0.0, 0.0, 0.0, 0.0};
    // chunk loads: Tinv 16 columns k x 128 entries j = 1024 double2 (4 per thread: column k = wave + 4 q, rows 2 lane);
    // A21 16 columns k x 32 rows = 256 double2 (one per thread: column tid >> 4, rows 2 (tid & 15))
    const int tk = tid >> 6, tj = (tid & 63) * 2, ak = tid >> 4, ai = (tid & 15) * 2;
    double2 gT0, gT1, gT2, gT3, gA;
#define TR_GLOAD(K0)                                                                      \
    do {                                                                                  \
        const double *q = Tinv + (int64_t)((K0) + tk) * CH_NB + tj;                       \
        gT0 = *(const double2 *)q; gT1 = *(const double2 *)(q + 4 * CH_NB);               \
        gT2 = *(const double2 *)(q + 8 * CH_NB); gT3 = *(const double2 *)(q + 12 * CH_NB); \
        gA = *(const double2 *)&a21[(int64_t)((K0) + ak) * ld + ai];                      \
    } while (0)
#define TR_SSTORE(BUF)                                                                    \
    do {                                                                                  \
        double *d = &sT[BUF][tk * TR_ST + tj];                                            \
        *(double2 *)d = gT0; *(double2 *)(d + 4 * TR_ST) = gT1;                           \
        *(double2 *)(d + 8 * TR_ST) = gT2; *(double2 *)(d + 12 * TR_ST) = gT3;            \
        *(double2 *)&sA[BUF][ak * TR_SA + ai] = gA;                                       \
    } while (0)
    TR_GLOAD(0);
    TR_SSTORE(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see chol_syrk_kernel
    __syncthreads();
    for (int c = 0; c < CH_NB / TR_KC; ++c) {
        const int buf = c & 1, k0 = c * TR_KC;
        if (c + 1 < CH_NB / TR_KC) TR_GLOAD(k0 + TR_KC);
        if (k0 <= ch * 64 + 63) {                      // this wave's columns all have j < k0: nothing left to add
#pragma unroll
            for (int kk = 0; kk < TR_KC; kk += 4) {
                const double b = sA[buf][(kk + l4) * TR_SA + rh * 16 + l15];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    if (k0 + kk > (ch * 4 + mb) * 16 + 15) continue;
                    acc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(sT[buf][(kk + l4) * TR_ST + (ch * 4 + mb) * 16 + l15], b, acc[mb], 0, 0, 0);
                }
            }
        }
        if (c + 1 < CH_NB / TR_KC) {
            TR_SSTORE(buf ^ 1);
            __syncthreads();
        }
    }
#undef TR_GLOAD
#undef TR_SSTORE
    // acc[mb][r] = X[i = 16 rh + l15][j = 64 ch + 16 mb + l4 + 4 r]
    double s = 0.0;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = ch * 64 + mb * 16 + l4 + 4 * r;
            a21[(int64_t)j * ld + rh * 16 + l15] = acc[mb][r];
            s = fma(acc[mb][r], ys[j], s);
        }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (l4 == 0) part[ch][rh * 16 + l15] = s;
    __syncthreads();
    if (tid < TR_ROWS) rhs[CH_NB + (int64_t)blockIdx.x * TR_ROWS + tid] -= part[0][tid] + part[1][tid];
}

// C -= X X' on 128 x 128 tiles of the lower triangle of the trailing matrix.  X = the panel just solved (t x 128,
// rows from `off + NB`), C = A[off+NB.., off+NB..].  tile_first / n_tiles select a range of the tile list: the tiles
// are numbered block column after block column... (bi, bj), bj <= bi: id = bi (bi + 1) / 2 + bj in row-major order of
// the triangle; col0_only launches the first block column only (bj = 0, id -> bi).
constexpr int SY_T = 128, SY_KC = 16, SY_S = SY_T + 16;
#ifndef SY_BAND_V
#define SY_BAND_V 16
#endif
constexpr int SY_BAND = SY_BAND_V;   // tile rows per band of the update's tile order
// General form: the trailing block C starts at row/column `coff` (absolute), X = the K columns from column `xcol` on,
// rows from `coff` (K = 128: one panel; K = 256: a pair of panels applied in one pass over the tiles -- half the HBM
// traffic per flop, which is what holds the K = 128 update at 57 % MFMA-busy at n = 20 000).  Tiles (bi, bj),
// bj <= bi < nt: head = 1 selects block columns [0, ncol) (the look-ahead launch), head = 0 the columns from ncol on.
__global__ __launch_bounds__(256, 2) void chol_syrk_kernel(double *__restrict__ A, int64_t ld, int coff, int xcol, int K,
                                                           int nt, int ncol, int head) {
    __shared__ __attribute__((aligned(16))) double sI[2][SY_KC * SY_S];
    __shared__ __attribute__((aligned(16))) double sJ[2][SY_KC * SY_S];
    int bi, bj;
    if (head) {      // block columns 0 .. ncol-1, column after column
        int id = blockIdx.x;
        bj = 0;
        while (bj < ncol && id >= nt - bj) { id -= nt - bj; ++bj; }
        if (bj >= ncol) return;
        bi = bj + id;
    } else {
        // tiles with ncol <= bj <= bi < nt, numbered row-major in that triangle; blocks are dealt to the XCDs round-robin
        // (block b runs on XCD b % 8), so give every XCD a contiguous range of tile rows: its X_I stays in its L2
        // Order within the triangle: BANDS of SY_BAND tile rows, inside a band column after column, inside a column
        // the band's rows.  The blocks of one XCD (a contiguous range of this order) then work on ~SY_BAND tiles that
        // share one X_J chunk while the band's X_I chunks stay in the XCD's L2 -- in plain row-major order every tile
        // row streamed all of X_J again (one 128 KB chunk per tile: as many bytes as the C tiles themselves).
        const int nr = nt - ncol, total = nr * (nr + 1) / 2;
        const int per = (total + 7) / 8;
        int id = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
        if (id >= total) return;
        int a0 = 0, h = 0;
        for (;; a0 += SY_BAND) {
            h = min(SY_BAND, nr - a0);
            const int cnt = a0 * h + h * (h + 1) / 2;
            if (id < cnt) break;
            id -= cnt;
        }
        int r, c;
        if (id < a0 * h) { c = id / h; r = a0 + id % h; }
        else {
            id -= a0 * h;
            int q = 0;
            while (id >= h - q) { id -= h - q; ++q; }
            c = a0 + q; r = c + id;
        }
        bi = r + ncol; bj = c + ncol;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
    const double *X = A + (int64_t)xcol * ld + coff;             // X[row][k] at X[row + k * ld]
    const double *xI = X + (int64_t)bi * SY_T, *xJ = X + (int64_t)bj * SY_T;
    // The accumulators START as the C tile (its loads are in flight while the first chunk of X is staged) and the
    // X_J operand is negated, so the MFMA chain itself computes C - X_I X_J' and the epilogue is stores only -- no
    // read-modify-write latency at the end of a tile.
    // acc[a][b][r] <-> C[i = bi T + wi + 16 b + l15][j = bj T + wj + 16 a + l4 + 4 r]
    double *C = A + (int64_t)coff * ld + coff;
    d4 acc[4][4];   // [a: j sub-block][b: i sub-block]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double *cj = C + (int64_t)(bj * SY_T + wj + a * 16 + l4 + 4 * r) * ld + bi * SY_T + wi + l15;
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b][r] = cj[b * 16];
        }
    // chunk = 16 columns k of 128 rows for each of X_I, X_J: 1024 double2 each, 4 + 4 per thread
    // (thread -> column k = wave + 4 q, rows 2 lane, 2 lane + 1: one wave reads one whole 1 KB column)
    const int gk = tid >> 6, gr = (tid & 63) * 2;
    const double *pI = xI + (int64_t)gk * ld + gr, *pJ = xJ + (int64_t)gk * ld + gr;
    const int64_t kstep = 4 * ld;
    double2 gI0, gI1, gI2, gI3, gJ0, gJ1, gJ2, gJ3;
#define SY_GLOAD(K0)                                                                                    \
    do {                                                                                                \
        const double *qI = pI + (int64_t)(K0) * ld, *qJ = pJ + (int64_t)(K0) * ld;                      \
        gI0 = *(const double2 *)qI; gI1 = *(const double2 *)(qI + kstep);                               \
        gI2 = *(const double2 *)(qI + 2 * kstep); gI3 = *(const double2 *)(qI + 3 * kstep);             \
        gJ0 = *(const double2 *)qJ; gJ1 = *(const double2 *)(qJ + kstep);                               \
        gJ2 = *(const double2 *)(qJ + 2 * kstep); gJ3 = *(const double2 *)(qJ + 3 * kstep);             \
    } while (0)
#define SY_SSTORE(BUF)                                                                                  \
    do {                                                                                                \
        double *dI = &sI[BUF][gk * SY_S + gr], *dJ = &sJ[BUF][gk * SY_S + gr];                          \
        *(double2 *)dI = gI0; *(double2 *)(dI + 4 * SY_S) = gI1;                                        \
        *(double2 *)(dI + 8 * SY_S) = gI2; *(double2 *)(dI + 12 * SY_S) = gI3;                          \
        *(double2 *)dJ = gJ0; *(double2 *)(dJ + 4 * SY_S) = gJ1;                                        \
        *(double2 *)(dJ + 8 * SY_S) = gJ2; *(double2 *)(dJ + 12 * SY_S) = gJ3;                          \
    } while (0)
    SY_GLOAD(0);
    SY_SSTORE(0);
    // every prologue load (the C tile in the accumulators) retired BEFORE the loop: otherwise the compiler carries
    // "accumulator load pending" into the loop and plants s_waitcnt vmcnt(3..0) beside the first MFMAs of every
    // iteration -- where, in steady state, they wait for the chunk loads issued a moment earlier
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    const int nchunk = K / SY_KC;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) SY_GLOAD((c + 1) * SY_KC);
#pragma unroll
        for (int kk = 0; kk < SY_KC; kk += 4) {
            double fi[4], fj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                fj[a] = -sJ[buf][(kk + l4) * SY_S + wj + a * 16 + l15];
                fi[a] = sI[buf][(kk + l4) * SY_S + wi + a * 16 + l15];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[a], fi[b], acc[a][b], 0, 0, 0);
        }
        if (c + 1 < nchunk) {
            SY_SSTORE(buf ^ 1);
            __syncthreads();
        }
    }
#undef SY_GLOAD
#undef SY_SSTORE
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double *cj = C + (int64_t)(bj * SY_T + wj + a * 16 + l4 + 4 * r) * ld + bi * SY_T + wi + l15;
#pragma unroll
            for (int b = 0; b < 4; ++b) cj[b * 16] = acc[a][b][r];
        }
}

// ---- back substitution x = L^-T y, panel by panel from the last ------------------------------------------------------
// x_j = L_jj^-T y_j : one block, Tinv_j = L_jj^-1 (lower, column-major): x[i] = sum_{k >= i} Tinv[k][i] y[k]
__global__ __launch_bounds__(128) void chol_bsolve_diag_kernel(const double *__restrict__ TinvT, double *__restrict__ y) {
    __shared__ double ys[CH_NB];
    ys[threadIdx.x] = y[threadIdx.x];
    __syncthreads();
    // TinvT[i + NB k] = Tinv[k][i]: thread i walks row i of the transpose, consecutive threads read consecutive
    // addresses, every load independent of the running sum (entries with k < i are zero)
    const double *row = TinvT + threadIdx.x;
    double s = 0.0;
#pragma unroll 16
    for (int k = 0; k < CH_NB; ++k) s = fma(row[(int64_t)k * CH_NB], ys[k], s);
    y[threadIdx.x] = s;
}
// y[c] -= sum_r L[j0 + r][c] x_j[r] for every column c < j0: one wave per column (128 rows = 2 per lane)
__global__ __launch_bounds__(256) void chol_bsolve_update_kernel(const double *__restrict__ A, int64_t ld, int off, int j0,
                                                                 double *__restrict__ y) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= j0) return;
    const double *col = A + (int64_t)(off + c) * ld + off + j0;
    const double2 l2 = *(const double2 *)&col[2 * lane];
    const double2 x2 = *(const double2 *)&y[j0 + 2 * lane];
    const double s = wave_sum(fma(l2.x, x2.x, l2.y * x2.y));
    if (lane == 0) y[c] -= s;
}

// Factor B + lambda I (already shifted) at A[off.., off..], order m, and solve for the right-hand side in rhs_dev
// (m_pad entries; overwritten with the solution).  A must hold m_pad = chol_padded(m) rows / columns from `off`.
// work: chol_work_doubles(m) doubles.  Launches on L.s / L.s2; returns after the solve has been ENQUEUED and the
// pivot flag read back (one stream synchronisation).
int chol_padded(int m) { return (m + CH_NB - 1) / CH_NB * CH_NB; }
size_t chol_work_doubles(int m) { return (size_t)(chol_padded(m) / CH_NB) * 2 * CH_NB * CH_NB; }

int cholesky_solve_mfma(FitLane &L, double *A, int64_t ld, int off, int m, double *rhs_dev, double *work, int *info_dev) {
    hipStream_t s = L.s, s2 = L.s2;
    const int m_pad = chol_padded(m), np = m_pad / CH_NB;
    // the panels are read with 16-byte loads: row `off` of every column must sit on a 16-byte boundary
    if ((((uintptr_t)(A + off)) & 15) || (ld & 1) || (((uintptr_t)rhs_dev) & 15) || (((uintptr_t)work) & 15)) {
        set_error("cholesky_solve_mfma: matrix rows from `off` must be 16-byte aligned (ld even)");
        return MHS_ERR_INVALID;
    }
    MHS_HIP(hipMemsetAsync(info_dev, 0, sizeof(int), s));
    if (m_pad > m) {
        const int64_t total = (int64_t)(m_pad - m) * m_pad;
        hipLaunchKernelGGL(chol_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A, ld, off, m, m_pad, rhs_dev);
    }
    static const size_t diag_lds = (size_t)(CH_NB * CH_LDP + (CH_NB - CH_SB) * CH_SB + CH_SB * (CH_SB + 1) + CH_SB + CH_NB) * sizeof(double);
    // per call: the attribute is per device, and factorisations run on several lanes (host threads) and device slots
    MHS_HIP(hipFuncSetAttribute((const void *)chol_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
    std::vector<hipEvent_t> &pool = L.pool;
    while ((int)pool.size() < 2 * np + 4) {
        hipEvent_t e;
        MHS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        pool.push_back(e);
    }
    // Panels are processed in PAIRS (A, B): A is factorised and solved, its update goes to B's block column only
    // (K = 128, one column of tiles), B is factorised and solved, and the rest of the trailing matrix receives both
    // panels in ONE pass over its tiles (K = 256).  Look-ahead as before: the first two block columns of that pass
    // -- the next pair's own columns -- are launched first on the main stream, the rest on the second stream, and the
    // next pair's factorisation runs beside it.
    constexpr bool pairs = true;
    hipEvent_t pending = nullptr;     // the second-stream update the next trailing pass has to wait for
    auto syrk = [&](int coff, int xcol, int K, int nt, int ncol) -> int {      // trailing block at coff, nt tiles a side
        if (nt <= 0) return MHS_OK;
        if (pending) { MHS_HIP(hipStreamWaitEvent(s, pending, 0)); pending = nullptr; }
        const int nc = std::min(ncol, nt);
        int head_tiles = 0;
        for (int c = 0; c < nc; ++c) head_tiles += nt - c;
        hipLaunchKernelGGL(chol_syrk_kernel, dim3((unsigned)head_tiles), dim3(256), 0, s, A, ld, coff, xcol, K, nt, nc, 1);
        return MHS_OK;
    };
    int nev = 0;
    auto syrk_rest = [&](int coff, int xcol, int K, int nt, int ncol) -> int {
        const int nr = nt - ncol;
        MHS_HIP(hipEventRecord(pool[nev], s));
        MHS_HIP(hipStreamWaitEvent(s2, pool[nev], 0));
        ++nev;
        if (nr > 0) {
            const int total = nr * (nr + 1) / 2, per = (total + 7) / 8;
            hipLaunchKernelGGL(chol_syrk_kernel, dim3((unsigned)(per * 8)), dim3(256), 0, s2, A, ld, coff, xcol, K, nt, ncol, 0);
        }
        MHS_HIP(hipEventRecord(pool[nev], s2));
        pending = pool[nev];
        ++nev;
        return MHS_OK;
    };
    for (int p = 0; p < np;) {
        const int jA = p * CH_NB, tA = m_pad - jA - CH_NB;
        double *TA = work + (size_t)p * 2 * CH_NB * CH_NB;
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(DG_THREADS), diag_lds, s, A, ld, off + jA, TA, rhs_dev + jA, info_dev);
        if (tA <= 0) { ++p; break; }
        hipLaunchKernelGGL(chol_trsm_kernel, dim3((unsigned)(tA / TR_ROWS)), dim3(256), 0, s, A, ld, off + jA, TA, rhs_dev + jA);
        const int ntA = tA / CH_NB;
        if (!pairs || p + 1 >= np) {      // single panel: its own K = 128 pass (head = the next panel's column)
            if (int rc = syrk(off + jA + CH_NB, off + jA, CH_NB, ntA, 1)) return rc;
            if (int rc = syrk_rest(off + jA + CH_NB, off + jA, CH_NB, ntA, 1)) return rc;
            ++p;
            continue;
        }
        // A's update of B's block column only
        if (int rc = syrk(off + jA + CH_NB, off + jA, CH_NB, ntA, 1)) return rc;
        const int jB = jA + CH_NB, tB = tA - CH_NB;
        double *TB = work + (size_t)(p + 1) * 2 * CH_NB * CH_NB;
        hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(DG_THREADS), diag_lds, s, A, ld, off + jB, TB, rhs_dev + jB, info_dev);
        if (tB > 0) {
            hipLaunchKernelGGL(chol_trsm_kernel, dim3((unsigned)(tB / TR_ROWS)), dim3(256), 0, s, A, ld, off + jB, TB, rhs_dev + jB);
            const int ntB = tB / CH_NB;
            if (int rc = syrk(off + jB + CH_NB, off + jA, 2 * CH_NB, ntB, 2)) return rc;
            if (int rc = syrk_rest(off + jB + CH_NB, off + jA, 2 * CH_NB, ntB, 2)) return rc;
        }
        p += 2;
    }
    // back substitution
    for (int p = np - 1; p >= 0; --p) {
        const int j = p * CH_NB;
        hipLaunchKernelGGL(chol_bsolve_diag_kernel, dim3(1), dim3(128), 0, s, work + (size_t)p * 2 * CH_NB * CH_NB + CH_NB * CH_NB, rhs_dev + j);
        if (j > 0) hipLaunchKernelGGL(chol_bsolve_update_kernel, dim3((unsigned)((j + 3) / 4)), dim3(256), 0, s, A, ld, off, j, rhs_dev);
    }
    MHS_HIP(hipGetLastError());
    int h_info = 0;
    MHS_HIP(hipMemcpyAsync(&h_info, info_dev, sizeof(int), hipMemcpyDeviceToHost, s));
    MHS_HIP(hipStreamSynchronize(s));
    if (h_info != 0) {
        set_error("mhs_tps_fit: Q2'KQ2 + lambda I is not positive definite (pivot %d)", h_info - off);
        return MHS_ERR_NUMERIC;
    }
    return MHS_OK;
}

}  // namespace mhs
