// Host-side pieces of the TPS fit (tps_gcv_host.hip), shared with tps_fit.hip.
#pragma once
#include <stdint.h>
#include <vector>

namespace mhs {

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

void qr_n3(std::vector<double> &T, int64_t n, std::vector<double> v[3], double tau[3], double R[9]);
void apply_reflector(const std::vector<double> &v, double tau, double *x, int64_t n);

}  // namespace mhs
