// micro-benchmark: predicate accumulation variants for the gbm LUT kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef float float2v __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters, float tks) {
    float4 key = ((const float4 *)in)[threadIdx.x];
    float tk = tks;   // uniform
    if (V == 0) {
        unsigned i0 = 0, i1 = 0, i2 = 0, i3 = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < 10; ++l) {
                asm volatile(
                    "v_cmp_gt_f32_e64 s[20:21], %4, %5\n"
                    "v_cmp_gt_f32_e64 s[22:23], %4, %6\n"
                    "v_cmp_gt_f32_e64 s[24:25], %4, %7\n"
                    "v_cmp_gt_f32_e64 s[26:27], %4, %8\n"
                    "v_addc_co_u32_e64 %0, s[20:21], %0, %0, s[20:21]\n"
                    "v_addc_co_u32_e64 %1, s[22:23], %1, %1, s[22:23]\n"
                    "v_addc_co_u32_e64 %2, s[24:25], %2, %2, s[24:25]\n"
                    "v_addc_co_u32_e64 %3, s[26:27], %3, %3, s[26:27]\n"
                    : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3)
                    : "s"(tk), "v"(key.x), "v"(key.y), "v"(key.z), "v"(key.w)
                    : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = (float)(i0 + i1 + i2 + i3);
    } else if (V == 1) {
        float2v a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        float2v k01 = {key.x, key.y}, k23 = {key.z, key.w};
        float2v ntk = {-tk, -tk};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int l = 0; l < 10; ++l) {
                float2v p01, p23;
                asm volatile(
                    "v_pk_add_f32 %2, %4, %6 op_sel_hi:[1,0] clamp\n"
                    "v_pk_add_f32 %3, %5, %6 op_sel_hi:[1,0] clamp\n"
                    "v_pk_fma_f32 %0, %0, 2.0, %2 op_sel_hi:[1,0,1]\n"
                    "v_pk_fma_f32 %1, %1, 2.0, %3 op_sel_hi:[1,0,1]\n"
                    : "+v"(a01), "+v"(a23), "=&v"(p01), "=&v"(p23)
                    : "v"(k01), "v"(k23), "s"(ntk));
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = a01.x + a01.y + a23.x + a23.y;
    }
}

int main() {
    float *in, *out;
    CK(hipMalloc(&in, 256 * 16));
    CK(hipMalloc(&out, 1024 * 256 * 4 * 8));
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)(i % 7);
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    const int blocks = 256 * 8;    // 8 blocks of 4 waves per CU = 8 waves/SIMD
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters, 3.0f);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters, 3.0f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // per SIMD: blocks*4 waves / (256 CUs*4 SIMDs) waves, each iters*10 levels of 4 cells
            double waves_per_simd = blocks * 4.0 / 1024.0;
            double lv = waves_per_simd * iters * 10.0;
            if (rep) printf("variant %d: %.3f ms, %.2f ns per wave-level(4 cells) per SIMD = %.2f cycles@2.4GHz\n", v, ms, ms * 1e6 / lv, ms * 1e6 / lv * 2.4);
        }
    }
    float r[4]; CK(hipMemcpy(r, out, 16, hipMemcpyDeviceToHost)); printf("chk %g %g\n", r[0], r[1]);
    return 0;
}
