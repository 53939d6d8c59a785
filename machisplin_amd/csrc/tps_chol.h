// MFMA blocked Cholesky + triangular solves (tps_chol.hip), used by the fixed-lambda route of the spline fit.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "common.h"

namespace mhs {
int chol_padded(int m);                 // m rounded up to the panel width: rows / columns the matrix must provide
size_t chol_work_doubles(int m);        // workspace (the inverted diagonal blocks), in doubles
// Solve (A[off.., off..]) x = rhs for the SPD matrix of order m stored at A (column-major, leading dimension ld, lower
// triangle read; destroyed) -- rhs_dev holds chol_padded(m) entries and receives x.  Row `off` of every column must be
// 16-byte aligned.  Work is enqueued on the lane's two streams; returns once the pivot flag has been read back.
int cholesky_solve_mfma(FitLane &L, double *A, int64_t ld, int off, int m, double *rhs_dev, double *work, int *info_dev);
}  // namespace mhs
