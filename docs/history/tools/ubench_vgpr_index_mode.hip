// micro-benchmark + semantics check: VGPR-indexed (s_set_gpr_idx) packed-f32 predicate accumulation
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float32v __attribute__((ext_vector_type(32)));

// keys: [8 vars][4 cells] per lane; levels: 5 per "tree" with var index from idx[] (uniform), c from cs[]
__global__ __launch_bounds__(256) void k(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ idxs, const float *__restrict__ cs, int trees) {
    float32v keys;
    for (int i = 0; i < 32; ++i) keys[i] = in[(i * 256 + threadIdx.x)];
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int t = 0; t < trees; ++t) {
        const int *ix = idxs + (t & 63) * 8;
        const float *cc = cs + (t & 63) * 8;
        const int i0 = __builtin_amdgcn_readfirstlane(ix[0]), i1 = __builtin_amdgcn_readfirstlane(ix[1]), i2 = __builtin_amdgcn_readfirstlane(ix[2]), i3 = __builtin_amdgcn_readfirstlane(ix[3]), i4 = __builtin_amdgcn_readfirstlane(ix[4]);
        const unsigned long long c01 = *(const unsigned long long *)(cc), c23 = *(const unsigned long long *)(cc + 2), c45 = *(const unsigned long long *)(cc + 4);
        const unsigned long long a0 = 0x48800000u + ((unsigned)(t & 63) << 5);
        float2v a01, a23, b01, b23;
        asm volatile(
            "s_set_gpr_idx_on %[i0], 0x1\n\t"
            "v_pk_add_f32 %[b01], v[96:97], %[c01] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_add_f32 %[b23], v[98:99], %[c01] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_fma_f32 %[a01], 2.0, %[a0], %[b01] op_sel_hi:[0,0,1]\n\t"
            "v_pk_fma_f32 %[a23], 2.0, %[a0], %[b23] op_sel_hi:[0,0,1]\n\t"
            "s_set_gpr_idx_idx %[i1]\n\t"
            "v_pk_add_f32 %[b01], v[96:97], %[c01] op_sel:[0,1] op_sel_hi:[1,1] clamp\n\t"
            "v_pk_add_f32 %[b23], v[98:99], %[c01] op_sel:[0,1] op_sel_hi:[1,1] clamp\n\t"
            "v_pk_fma_f32 %[a01], 2.0, %[a01], %[b01] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[a23], 2.0, %[a23], %[b23] op_sel_hi:[0,1,1]\n\t"
            "s_set_gpr_idx_idx %[i2]\n\t"
            "v_pk_add_f32 %[b01], v[96:97], %[c23] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_add_f32 %[b23], v[98:99], %[c23] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_fma_f32 %[a01], 2.0, %[a01], %[b01] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[a23], 2.0, %[a23], %[b23] op_sel_hi:[0,1,1]\n\t"
            "s_set_gpr_idx_idx %[i3]\n\t"
            "v_pk_add_f32 %[b01], v[96:97], %[c23] op_sel:[0,1] op_sel_hi:[1,1] clamp\n\t"
            "v_pk_add_f32 %[b23], v[98:99], %[c23] op_sel:[0,1] op_sel_hi:[1,1] clamp\n\t"
            "v_pk_fma_f32 %[a01], 2.0, %[a01], %[b01] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[a23], 2.0, %[a23], %[b23] op_sel_hi:[0,1,1]\n\t"
            "s_set_gpr_idx_idx %[i4]\n\t"
            "v_pk_add_f32 %[b01], v[96:97], %[c45] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_add_f32 %[b23], v[98:99], %[c45] op_sel_hi:[1,0] clamp\n\t"
            "v_pk_fma_f32 %[a01], 2.0, %[a01], %[b01] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[a23], 2.0, %[a23], %[b23] op_sel_hi:[0,1,1]\n\t"
            "s_set_gpr_idx_off"
            : [a01] "=&v"(a01), [a23] "=&v"(a23), [b01] "=&v"(b01), [b23] "=&v"(b23)
            : "{v[96:127]}"(keys), [i0] "s"(i0), [i1] "s"(i1), [i2] "s"(i2), [i3] "s"(i3), [i4] "s"(i4),
              [c01] "s"(c01), [c23] "s"(c23), [c45] "s"(c45), [a0] "s"(a0));
        s0 += (float)(__float_as_int(a01.x) - 0x4B000000); s1 += (float)(__float_as_int(a01.y) - 0x4B000000);
        s2 += (float)(__float_as_int(a23.x) - 0x4B000000); s3 += (float)(__float_as_int(a23.y) - 0x4B000000);
    }
    float *o = out + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
}

int main() {
    float *in, *out, *cs; int *idxs;
    const int blocks = 256 * 4;
    CK(hipMalloc(&in, 32 * 256 * 4)); CK(hipMalloc(&out, (size_t)blocks * 256 * 16));
    CK(hipMalloc(&idxs, 64 * 8 * 4)); CK(hipMalloc(&cs, 64 * 8 * 4));
    static float h[32 * 256]; static int hi[512]; static float hc[512];
    srand(1);
    for (int i = 0; i < 32 * 256; ++i) h[i] = -(float)(rand() % 100);     // -rank
    for (int i = 0; i < 512; ++i) { hi[i] = (rand() % 5) * 4; hc[i] = (float)(rand() % 100 + 1); }
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice)); CK(hipMemcpy(idxs, hi, sizeof(hi), hipMemcpyHostToDevice));
    CK(hipMemcpy(cs, hc, sizeof(hc), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int trees = 6400;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, idxs, cs, trees);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double waves_per_simd = blocks * 4.0 / 1024.0;
        if (rep) printf("gpr-idx: %.3f ms, %.2f cycles@2.4GHz per wave-tree(4 cells) per SIMD\n", ms, ms * 1e6 / (waves_per_simd * trees) * 2.4);
    }
    static float r[256 * 4]; CK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
    // host check for lanes 0..255 of block 0
    int bad = 0;
    for (int lane = 0; lane < 256; ++lane) for (int c = 0; c < 4; ++c) {
        float s = 0;
        for (int t = 0; t < trees; ++t) {
            int tt = t & 63; unsigned idx = 0;
            for (int q = 0; q < 5; ++q) { int v4 = hi[tt * 8 + q]; float key = h[(v4 + c) * 256 + lane]; float d = key + hc[tt * 8 + q]; idx = idx * 2 + (d >= 1.f ? 1 : 0); }
            s += (float)((tt << 5) + idx);
        }
        if (s != r[lane * 4 + c]) { if (bad < 5) printf("mismatch lane %d c %d: %g vs %g\n", lane, c, r[lane * 4 + c], s); ++bad; }
    }
    printf("bad %d\n", bad);
    return 0;
}
