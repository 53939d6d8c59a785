// What the single-split path of gbm_coherent_kernel costs per (wave, tree), piece by piece (round 5; scratch, not product):
// 16 waves per CU (one 1 024-thread block, as the kernel), every wave looping over "trees".
//   hipcc --offload-arch=gfx950 -O3 tools/micro/readlane_idx_rate.hip -o /tmp/rl && /tmp/rl
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 20000;
typedef float float32v __attribute__((ext_vector_type(32)));

template <int WHAT>
__global__ __launch_bounds__(1024) void k(double *out, int seed) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    float32v keys;
    for (int i = 0; i < 32; ++i) keys[i] = -(float)((lane * 7 + i * 13 + seed) & 255);
    int cl = (lane * 5 + seed) & 255, vl = ((lane + seed) % 5) << 2;
    double d1 = 1.0 + lane * 1e-3;
    double acc[4] = {0, 0, 0, 0};
    int sink = 0;
    for (int it = 0; it < ITER; ++it) {
        const int t = __builtin_amdgcn_readfirstlane((it * 7 + seed) & 63);
        int cc = 3 + (it & 7), vo = (it & 3) << 2;
        double e1 = 1.5;
        if (WHAT == 0 || WHAT == 3) {          // the four readlanes
            cc = __builtin_amdgcn_readlane(cl, t);
            vo = __builtin_amdgcn_readlane(vl, t) & 0xFF;
            e1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d1), t), __builtin_amdgcn_readlane(__double2loint(d1), t));
            if (WHAT == 0) sink += cc + vo + __double2hiint(e1) + __double2loint(e1);
        }
        int h[4] = {0x3F800000, 0, 0x3F800000, 0};
        if (WHAT == 1 || WHAT == 3) {          // index mode + four clamped adds
            asm volatile("s_set_gpr_idx_on %[v], 0x1\n\t"
                         "v_add_f32_e64 %[a], v64, %[c] clamp\n\t"
                         "v_add_f32_e64 %[b], v65, %[c] clamp\n\t"
                         "v_add_f32_e64 %[d], v66, %[c] clamp\n\t"
                         "v_add_f32_e64 %[e], v67, %[c] clamp\n\t"
                         "s_set_gpr_idx_off"
                         : [a] "=&v"(h[0]), [b] "=&v"(h[1]), [d] "=&v"(h[2]), [e] "=&v"(h[3])
                         : "{v[64:95]}"(keys), [v] "s"(vo), [c] "s"(cc));
            if (WHAT == 1) sink += h[0] + h[1] + h[2] + h[3];
        }
        if (WHAT == 4) {                       // the adds without the index mode (fixed predictor)
            asm volatile("v_add_f32_e64 %[a], v64, %[c] clamp\n\t"
                         "v_add_f32_e64 %[b], v65, %[c] clamp\n\t"
                         "v_add_f32_e64 %[d], v66, %[c] clamp\n\t"
                         "v_add_f32_e64 %[e], v67, %[c] clamp"
                         : [a] "=&v"(h[0]), [b] "=&v"(h[1]), [d] "=&v"(h[2]), [e] "=&v"(h[3])
                         : "{v[64:95]}"(keys), [c] "s"(cc));
            sink += h[0] + h[1] + h[2] + h[3];
        }
        if (WHAT == 2 || WHAT == 3) {          // four fp64 fmas with a scalar operand
            if (WHAT == 2) { h[0] ^= it; e1 = __hiloint2double(0x3FF00000 + (it & 15), it); }
            for (int c = 0; c < 4; ++c) acc[c] = fma(__hiloint2double(h[c], 0), e1, acc[c]);
        }
        if (WHAT == 5) {                       // one broadcast LDS read + readfirstlane instead of four readlanes
            const int4 q = *(const int4 *)(smem + (t << 4));
            vo = __builtin_amdgcn_readfirstlane(q.y) & 0xFF;
            sink += q.x + vo + q.z + q.w;
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + sink;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 1024));
    const char *names[] = {"4 v_readlane", "idx on + 4 v_add clamp + idx off", "4 v_fma_f64 (SGPR operand)", "all three (the loop body)",
                           "4 v_add clamp, no index mode", "ds_read_b128 broadcast + v_readfirstlane"};
    void (*ks[])(double *, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>};
    for (int w = 0; w < 6; ++w) {
        CK(hipFuncSetAttribute((const void *)ks[w], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(ks[w], dim3(blocks), dim3(1024), 100 * 1024, 0, out, 3);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(ks[w], dim3(blocks), dim3(1024), 100 * 1024, 0, out, 3);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        // one block of 16 waves per CU: 4 waves per SIMD; cycles of a SIMD per trip of its 4 waves
        printf("%-44s %8.3f ms   %6.1f cycles @ 2.4 GHz per (wave, trip) per SIMD share  (= %5.1f per trip of the SIMD's 4 waves)\n", names[w], ms,
               ms * 1e-3 * 2.4e9 / ITER / 4, ms * 1e-3 * 2.4e9 / ITER);
    }
    return 0;
}
