// Does the LDS-DMA destination base (M0) reach LDS addresses beyond 64 KB on gfx950 (160 KB of LDS per CU)?  And when is the
// data visible to ANOTHER wave of the block: at the issuing wave's vmcnt(0)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma_range.hip -o /tmp/lds_dma_range && /tmp/lds_dma_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory");
}

// wave 1 copies 1 KB chunks to `dst` by LDS-DMA, waits vmcnt(0) and raises a flag in LDS; wave 0 polls the flag and reads the chunk back
__global__ __launch_bounds__(128) void probe(const unsigned *src, unsigned *out, unsigned dst, int chunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned flag = 163840 - 16;
    if (threadIdx.x == 0) *(volatile unsigned *)(smem + flag) = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (threadIdx.x >= 64) {
        for (int q = 0; q < chunks; ++q)
            glds16((const char *)src + q * 1024 + lane * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + q * 1024)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(flag), "v"(1u) : "memory");
    } else {
        unsigned v;
        do {
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag) : "memory");
        } while (__builtin_amdgcn_readfirstlane((int)v) == 0);
        for (int q = 0; q < chunks; ++q)
            for (int k = 0; k < 4; ++k) {
                unsigned a = dst + q * 1024 + lane * 16 + k * 4, r;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
                out[q * 256 + lane * 4 + k] = r;
            }
    }
}

int main() {
    const int chunks = 8;
    std::vector<unsigned> h(chunks * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0xA5000000u + (unsigned)i;
    unsigned *src, *out;
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&out, h.size() * 4));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    for (unsigned dst : {0u, 32768u, 60416u, 65536u, 76800u, 102400u, 150000u & ~1023u}) {
        CK(hipMemset(out, 0, h.size() * 4));
        probe<<<1, 128, 163840>>>(src, out, dst, chunks);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> g(h.size());
        CK(hipMemcpy(g.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h.size(); ++i) bad += g[i] != h[i];
        printf("LDS-DMA to byte %6u .. %6u: %zu of %zu words wrong (first read back %08x, want %08x)\n", dst, dst + chunks * 1024, bad, h.size(), g[0], h[0]);
    }
    // repeated launches: a race between vmcnt(0) and the LDS write would show up as sporadic mismatches
    size_t bad_total = 0;
    for (int rep = 0; rep < 2000; ++rep) {
        probe<<<1, 128, 163840>>>(src, out, 76800u, chunks);
        if (rep % 100 == 99) {
            CK(hipDeviceSynchronize());
            std::vector<unsigned> g(h.size());
            CK(hipMemcpy(g.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < h.size(); ++i) bad_total += g[i] != h[i];
        }
    }
    {   // source 8-byte but not 16-byte aligned (a forest's trees start at any record)
        CK(hipMemset(out, 0, h.size() * 4));
        probe<<<1, 128, 163840>>>(src + 2, out, 25600u, chunks - 1);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> g(h.size());
        CK(hipMemcpy(g.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < (size_t)(chunks - 1) * 256; ++i) bad += g[i] != h[i + 2];
        printf("source at +8 bytes (not 16-byte aligned): %zu words wrong (first %08x, want %08x)\n", bad, g[0], h[2]);
    }
    printf("2000 launches, flag after vmcnt(0): %zu words wrong in the sampled read-backs\n", bad_total);
    return 0;
}
