// Batched small thin-plate-spline fits (tps_batch.hip): every spline of a reference-tiled Step 3 (V73:690-738: one
// fields::Tps per tile on the 130-250 stations of its fit box) fitted by ONE kernel launch, one workgroup per spline.
#pragma once
#include <stdint.h>
#include <vector>
#include "common.h"
#include "tps_host.h"

namespace mhs {

constexpr int SB_NMAX = 256;        // distinct stations one workgroup can hold (the matrix lives in its registers)
constexpr int SB_NMIN = 8;

// one spline of a batch as the kernel reads it (device array)
struct SmallJob {
    int n, N, gcv_mode, pad0;
    double lambda;                  // NaN: pick it by GCV
    double pure_ss;
    double htau[3], R[9], w1[3];
    int64_t in_off;                 // doubles into the packed input: u[n], v[n], sw[n], hv0[n], hv1[n], hv2[n], wv[n]
    int64_t perm_off;               // ints: position of knot i in the evaluation's (bin-sorted) knot order, or -1 entries when unused
    int64_t c_off;                  // doubles: the coefficients c[n], natural order
    int64_t knot_off;               // Knots: the evaluation's knot records, sorted order
};
// what a fit hands back (device array, 16 doubles per spline); t_us: microseconds spent in the Gram matrix + projection,
// the tridiagonalisation, the extreme eigenvalues, bracket + grid, the golden section, solve + back-transform (100 MHz counter)
struct SmallResult { double lambda, gcv, eff_df, d[3], status, pad; double t_us[8]; };

// Device buffers of one batch, carved from one grow-only arena of the slot (FitLane::arena of lane `lane`).
struct SmallBatch {
    int count = 0;
    std::vector<SmallJob> jobs;             // host copy
    std::vector<double> in;                 // packed host input
    std::vector<int> perm;                  // packed host permutations
    int nmax = 0;
    int64_t c_total = 0, knot_total = 0;
    // device (valid after small_batch_launch)
    SmallJob *jobs_dev = nullptr;
    double *in_dev = nullptr, *c_dev = nullptr;
    int *perm_dev = nullptr;
    Knot *knots_dev = nullptr;
    SmallResult *res_dev = nullptr;
};

// append one prepared fit; perm may be NULL (knots written in natural order).  Returns the job index.
int small_batch_add(SmallBatch &B, const TpsPrep &P, double lambda, int gcv_mode, const int *perm);
// upload and launch on `s`; the buffers come from the lane's arena (grown if needed: synchronises the device, first calls
// only), with `extra_bytes` more behind them for the caller (*extra_dev)
int small_batch_launch(SmallBatch &B, FitLane &L, hipStream_t s, size_t extra_bytes = 0, char **extra_dev = nullptr);
// after the stream has been synchronised: copy results (and optionally coefficients) back
int small_batch_results(const SmallBatch &B, hipStream_t s, std::vector<SmallResult> &res, std::vector<double> *c);


// Batched grid evaluation of the batch's splines (tps_eval.hip): one window per spline -- a tile's keep window on its
// fit raster (terra::interpolate(terra::rast(rb), tps), V73:726) -- all windows in one nodes launch + one cells launch,
// reading the knot records and the polynomial part where the fit kernel left them on the device.
struct EvalBatch;
EvalBatch *eval_batch_create();
void eval_batch_destroy(EvalBatch *B);
// plan window [r0, r1) x [c0, c1) of `grid` for a spline with these (scaled) knots; perm[i] = position of knot i in the
// evaluation's knot order (what SmallJob::perm_off points at); res_index / knot_off: the spline's job in the fit batch
int eval_batch_add(EvalBatch *B, const double *knots_uv, int n, const double *center, const double *scale, const mhs_grid *grid,
                   int64_t r0, int64_t r1, int64_t c0, int64_t c1, double *out_dev, int64_t ld, int res_index, int64_t knot_off,
                   std::vector<int> &perm);
// the same in two halves: the plan is pure (any host thread), the commit appends to the batch (one thread, in job order)
struct EvalPlanHandle;
EvalPlanHandle *eval_batch_plan(const double *knots_uv, int n, const double *center, const double *scale, const mhs_grid *grid,
                                int64_t r0, int64_t r1, int64_t c0, int64_t c1, double *out_dev, int64_t ld, std::vector<int> &perm, int *rc_out);
void eval_batch_commit(EvalBatch *B, EvalPlanHandle *h, int res_index, int64_t knot_off);
void eval_plan_drop(EvalPlanHandle *h);
size_t eval_batch_device_bytes(const EvalBatch *B);
int eval_batch_launch(EvalBatch *B, char *dev, const Knot *knots_base, const SmallResult *res_base, hipStream_t s);

}  // namespace mhs
