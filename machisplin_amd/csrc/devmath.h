// Device math shared by the TPS kernels: table-driven FP64 log for r^2 log r.
#pragma once
#include "common.h"

namespace mhs {

constexpr double LN2_D = 0.6931471805599453094;
constexpr double LOG_BIAS_D = -1023.0 * 0.6931471805599453094;

// returns log(d2) + 1023 ln2 for finite d2 >= 0 (d2 == 0 gives a finite value, so
// d2 * log(d2) evaluates to 0 at a knot without a branch; fields floors d2 at 1e-20,
// where d2 log d2 = -4.6e-19, far below one ulp of any non-trivial sum).
// tab: LOG_TAB_N x {2^1023 / c_i, log c_i} (runtime.hip), normally staged in LDS.
// r2 = 2^e m, m in [1,2): log m = log c_i + log1p(m/c_i - 1), |m/c_i - 1| <= 2^-11, cubic
// log1p => absolute error < 2e-14.
__device__ __forceinline__ double table_log_biased(double d2, const double2 *tab) {
    const int hi = __double2hiint(d2);
    const int e = hi >> 20;  // biased exponent (d2 >= 0)
    const unsigned off = ((unsigned)hi >> (20 - LOG_TAB_BITS - 4)) & ((LOG_TAB_N - 1) << 4);
    const double2 t = *(const double2 *)((const char *)tab + off);
    // t.x = 2^1023 / c_i: subtracting d2's exponent field from its high word gives 2^-E / c_i, so
    // d2 * that = m / c_i without assembling the mantissa m.  (asm: keeps the two 32-bit ops on
    // the high word; the compiler otherwise widens them into a 64-bit subtract with carry.)
    int ih = __double2hiint(t.x), tmp;
    asm("v_and_b32 %1, 0x7ff00000, %2\n\tv_sub_u32 %0, %0, %1" : "+v"(ih), "=&v"(tmp) : "v"(hi));
    const double invc = __hiloint2double(ih, __double2loint(t.x));
    const double r = fma(d2, invc, -1.0);
    const double q = fma(r, 1.0 / 3.0, -0.5);
    const double r2 = r * r;
    const double lp = fma(r2, q, r);
    const double L = fma((double)e, LN2_D, t.y);
    return L + lp;
}

// d2 * log(d2), exactly 0 for d2 == 0
__device__ __forceinline__ double r2logr2(double d2, const double2 *tab) {
    return d2 * (table_log_biased(d2, tab) + LOG_BIAS_D);
}

__device__ __forceinline__ void stage_log_table(double2 *lds_tab, const double2 *gtab) {
    for (int i = threadIdx.x; i < LOG_TAB_N; i += blockDim.x) lds_tab[i] = gtab[i];
    __syncthreads();
}

// x of another lane through a DPP move (no LDS crossbar traffic); rows outside ROW_MASK read 0.0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double x, int lane) {   // uniform: lands in SGPRs
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane),
                            __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// Sum over the (fully active) wave, in every lane: xor 1, xor 2, mirror within 8, mirror within 16 leave each row of
// 16 lanes with its total; the four row totals are read with v_readlane and added in a fixed order.  23 VALU
// instructions, against 12 ds_bpermute round trips for the butterfly.
__device__ __forceinline__ double wave_sum(double x) {
    x += dpp_fetch<0xB1, 0xf>(x);
    x += dpp_fetch<0x4E, 0xf>(x);
    x += dpp_fetch<0x141, 0xf>(x);
    x += dpp_fetch<0x140, 0xf>(x);
    return (lane_value(x, 0) + lane_value(x, 16)) + (lane_value(x, 32) + lane_value(x, 48));
}

// sum over a block of up to 1024 threads; result valid in every thread
__device__ __forceinline__ double block_sum(double x, double *scratch /* >= 17 doubles in LDS */) {
    x = wave_sum(x);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < nw; ++i) s += scratch[i];
        scratch[16] = s;
    }
    __syncthreads();
    return scratch[16];
}

}  // namespace mhs
