// Does a CU mask on the stream of a grid-filling kernel leave compute units that ANOTHER stream's small kernels get at once?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cu_mask_probe.hip -o exp/cu_mask_probe && exp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <thread>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void busy(double *o, int iters) {      // ~ms per block, 96 VGPR-ish, fills the chip
    double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) { x = fma(x, 0.999999, y); y = fma(y, 0.999999, x * 1e-9); }
    o[(size_t)blockIdx.x * 256 + threadIdx.x] = x + y;
}
__global__ __launch_bounds__(512) void small(double *o) { o[threadIdx.x] = threadIdx.x; }
int main() {
    int ncu = 0; hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); ncu = prop.multiProcessorCount;
    double *o; CK(hipMalloc(&o, (size_t)200000 * 256 * 8));
    double *o2; CK(hipMalloc(&o2, 4096));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sb; CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    for (int reserve : {0, 8, 16, 32, 64}) {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int i = 0; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
        if (reserve) { int stride = ncu / reserve + 1; for (int k = 0, i = 0; k < reserve; ++k, i = (i + stride) % ncu) mask[i / 32] &= ~(1u << (i % 32)); }
        hipStream_t sa; CK(hipExtStreamCreateWithCUMask(&sa, (uint32_t)mask.size(), mask.data()));
        hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
        busy<<<100000, 256, 0, sa>>>(o, 2000);                         // warm-up
        CK(hipStreamSynchronize(sa));
        CK(hipEventRecord(a0, sa));
        busy<<<100000, 256, 0, sa>>>(o, 20000);
        CK(hipEventRecord(a1, sa));
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        // 200 dependent small launches on the other stream while the big kernel runs
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 200; ++k) small<<<1, 512, 0, sb>>>(o2);
        CK(hipStreamSynchronize(sb));
        double chain_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipStreamSynchronize(sa));
        float big_ms; CK(hipEventElapsedTime(&big_ms, a0, a1));
        printf("reserve %2d CUs: grid-filling kernel %7.1f ms ; 200 dependent 1-block launches beside it: %8.2f ms (%.1f us each)\n",
               reserve, big_ms, chain_ms, chain_ms * 1e3 / 200);
        CK(hipStreamDestroy(sa));
    }
    return 0;
}
