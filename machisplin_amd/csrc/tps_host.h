// Host-side pieces of the TPS fit (tps_gcv_host.hip), shared with tps_fit.hip.
#pragma once
#include <stdint.h>
#include <vector>

namespace mhs {

struct GcvPool;

// GCV machinery on the tridiagonal form; see tps_gcv_host.hip
struct TridiagGcv {
    const double *a = nullptr;  // diagonal of T, m
    const double *b = nullptr;  // off-diagonal of T, m-1
    const double *g = nullptr;  // P' Q2' y, m
    int64_t m = 0, n = 0, N = 0; // m = n - 3 ; n unique stations ; N observations
    double pure_ss = 0.0;
    mutable std::vector<double> work_dp, work_dm, work_q;
    // GCV(lam), trA(lam) and optionally q = (T + lam I)^-1 g
    void eval(double lam, double *gcv, double *tra, double *q_out) const;
    double find_lambda(int mode) const;
};

// The same criterion on a symmetric BANDED form (bandwidth bw): the GPU reduces B only to a band
// (blocked, BLAS-3 style; tps_fit.hip) and every GCV evaluation is a banded Cholesky, a banded
// solve and the trace of the inverse by Takahashi's selected inversion, O(m bw^2).
// ab: lower band, column-major: ab[d + (bw + 1) * j] = M[j + d][j], d = 0..bw.
struct BandGcv {
    const double *ab = nullptr;
    const double *g = nullptr;
    int64_t m = 0, n = 0, N = 0;
    int bw = 1;
    double pure_ss = 0.0;
    int threads = 0;  // host threads for the independent evaluations (0 = the process-wide pool)
    struct GcvPool *pool = nullptr;   // a pool already leased (and awake) for this search, or NULL
    struct Work { std::vector<double> L, Z, q; };
    bool eval(double lam, double *gcv, double *tra, double *q_out, Work &w) const;
    bool finish(double lam, double qq, double tr_inv, double *gcv, double *tra) const;
    double find_lambda(int mode) const;
    int inertia_below(double x, Work &w) const;  // eigenvalues < x (unpivoted banded LDL')
    double eig_kth(int64_t k) const;
};

// Lease the worker pool of a GCV search ahead of time: waking sleeping workers costs about a millisecond, which
// the fit hides behind the tail of the band reduction.  gcv_pool_release(NULL) is a no-op.
GcvPool *gcv_pool_lease(int threads);
void gcv_pool_release(GcvPool *p);

void qr_n3(std::vector<double> &T, int64_t n, std::vector<double> v[3], double tau[3], double R[9]);

// The O(N) host part every fit starts with (fields' Krig.replicates + scale.type = "range" + QR of W^1/2 [1 u v]):
// unique locations in first-appearance order with their means and counts, range-scaled coordinates, the three
// reflectors of the polynomial block and the rotated data Q'(W^1/2 ym).  Shared by mhs_tps_fit (tps_fit.hip) and the
// batched small fits (tps_batch.hip).  Returns MHS_OK or an error code with the message set.
struct TpsPrep {
    int64_t N = 0, n = 0;               // observations, distinct locations
    double pure_ss = 0.0;
    double center[2] = {0, 0}, scale[2] = {1, 1};
    std::vector<double> xm, ym, w;      // unique coordinates (n x 2 column-major), means, counts
    std::vector<double> uv, sw;         // scaled coordinates (n x 2 column-major), sqrt(counts)
    std::vector<double> hv[3];          // reflectors of the QR of W^1/2 [1 u v]
    double htau[3] = {0, 0, 0}, R[9] = {0};
    std::vector<double> wv;             // Q' W^1/2 ym
};
int tps_prepare(const double *xy, const double *y, int64_t N, TpsPrep &P);
void apply_reflector(const std::vector<double> &v, double tau, double *x, int64_t n);

}  // namespace mhs
