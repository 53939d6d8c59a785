// Does v_mfma_f64_16x16x4_f64 run beside FP64 vector instructions on gfx950, or do the two share the FP64 datapath?
// Three kernels, the same number of waves (2 per SIMD): MFMA only, v_fma_f64 only, both in one loop.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o gpurun_out/mfma_valu && gpurun_out/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4v __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(double *o, int iters, double a, double b) {
    d4v d0 = {a, a, a, a}, d1 = d0, d2 = d0, d3 = d0;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d3, 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fma(x[i], b, a);      // 64 v_fma_f64 = 256 cycles at 4 cycles each
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    o[blockIdx.x * 256 + threadIdx.x] = s + d0[0] + d1[1] + d2[2] + d3[3];
}
template <int MODE> float run(double *o, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<512, 256>>>(o, 10, 1.0, 0.5);
    hipEventRecord(e0);
    k<MODE><<<512, 256>>>(o, iters, 1.0, 0.999);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double *o; hipMalloc(&o, 512 * 256 * 8);
    const int iters = 20000;
    float m1 = run<1>(o, iters), m2 = run<2>(o, iters), m3 = run<3>(o, iters);
    // per SIMD: 2 waves x iters x (4 MFMA | 64 FMA)
    printf("MFMA only  %8.2f ms  -> %.1f cycles per MFMA per SIMD at 2.4 GHz\n", m1, m1 * 1e-3 * 2.4e9 / (2.0 * iters * 4));
    printf("FMA only   %8.2f ms  -> %.2f cycles per v_fma_f64 per SIMD\n", m2, m2 * 1e-3 * 2.4e9 / (2.0 * iters * 64));
    printf("both       %8.2f ms  (sum %.2f, max %.2f)\n", m3, m1 + m2, m1 > m2 ? m1 : m2);
    return 0;
}
