// instruction-rate microbenchmarks for gfx950 (scratch; not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int ITER = 4096;
typedef double d4 __attribute__((ext_vector_type(4)));

template<int OP> __global__ __launch_bounds__(256) void k_rate(double* out, double a, double b, int ai) {
  double x[8]; int xi[8];
  for (int i=0;i<8;++i){ x[i] = a + threadIdx.x*1e-3 + i; xi[i] = ai + threadIdx.x + i; }
  __shared__ double2 tab[1024];
  for (int i=threadIdx.x;i<1024;i+=256) tab[i]=make_double2(i*1e-3, i*2e-3);
  __syncthreads();
  for (int it=0; it<ITER; ++it) {
#pragma unroll
    for (int i=0;i<8;++i) {
      if (OP==0) x[i] = fma(x[i], a, b);
      if (OP==1) x[i] = x[i]*a;
      if (OP==2) x[i] = x[i]+b;
      if (OP==3) { x[i] = (double)xi[i]; xi[i] += __double2loint(x[i]) ; }  // cvt_f64_i32 + 1 i32 add
      if (OP==4) { xi[i] = (xi[i] >> 3) + ai; }  // 2 i32 ops
      if (OP==5) { double2 t = tab[(xi[i]) & 1023]; xi[i] += __double2loint(t.x); } // random-ish ds_read_b128 + and + add
      if (OP==6) { double2 t = tab[(threadIdx.x + it) & 1023]; x[i] += t.x; } // sequential ds_read + add_f64
      if (OP==7) { x[i] = (x[i] < a) ? b : x[i]; } // cmp + cndmask x2
      if (OP==8) { float f = __builtin_amdgcn_logf((float)xi[i]); xi[i] += __float_as_int(f);} // v_log_f32 + cvt + add
      if (OP==9) { x[i] = fmax(x[i], a); }
      if (OP==10) { x[i] = __builtin_amdgcn_rcp(x[i]); }
    }
  }
  double s=0; for(int i=0;i<8;++i) s += x[i] + xi[i];
  out[blockIdx.x*256+threadIdx.x]=s;
}

__global__ __launch_bounds__(256) void k_mfma(double* out, double a) {
  d4 acc[4]; for (int i=0;i<4;++i) acc[i] = (d4){0,0,0,0};
  double x = a + threadIdx.x, y = a - threadIdx.x;
  for (int it=0; it<ITER; ++it) {
#pragma unroll
    for (int i=0;i<4;++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0,0,0);
  }
  double s=0; for(int i=0;i<4;++i) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*256+threadIdx.x]=s;
}
// mfma + valu concurrently
__global__ __launch_bounds__(256) void k_mfma_valu(double* out, double a, double b) {
  d4 acc[2]; for (int i=0;i<2;++i) acc[i] = (d4){0,0,0,0};
  double x[8]; for (int i=0;i<8;++i) x[i]=a+threadIdx.x+i;
  double xx = a + threadIdx.x, y = a - threadIdx.x;
  for (int it=0; it<ITER; ++it) {
#pragma unroll
    for (int i=0;i<2;++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(xx, y, acc[i], 0,0,0);
#pragma unroll
    for (int r=0;r<2;++r)
#pragma unroll
    for (int i=0;i<8;++i) x[i] = fma(x[i], a, b);
  }
  double s=0; for(int i=0;i<2;++i) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  for (int i=0;i<8;++i) s+=x[i];
  out[blockIdx.x*256+threadIdx.x]=s;
}

template<typename F> double timeit(F f, int reps=5) {
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for(int r=0;r<reps;++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms,e0,e1)); return ms/reps*1e-3;
}
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  int blocks = p.multiProcessorCount*8; double* out; CK(hipMalloc(&out, sizeof(double)*blocks*256));
  const char* names[] = {"fma_f64","mul_f64","add_f64","cvt_f64_i32+cvt_i32+add","2x i32 (ashr,add)","ds_read_b128 rand +2 i32","ds_read_b128 seq + add_f64","cmp_f64+cndmask","log_f32+cvt+add","max_f64","rcp_f64"};
  auto run=[&](auto kern, const char* nm, double ops_per_iter){
    double t = timeit([&]{ hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9, 3); });
    double waveops = (double)blocks*4*ITER*ops_per_iter; // wave-instructions
    double per_simd_cycles = t*2.4e9 / (waveops/(p.multiProcessorCount*4));
    printf("%-32s %.3f ms  %.2f cycles@2.4GHz per wave-iter-op (per SIMD)\n", nm, t*1e3, per_simd_cycles);
  };
  run(k_rate<0>, names[0], 8); run(k_rate<1>, names[1], 8); run(k_rate<2>, names[2], 8);
  run(k_rate<3>, names[3], 8); run(k_rate<4>, names[4], 8); run(k_rate<5>, names[5], 8);
  run(k_rate<6>, names[6], 8); run(k_rate<7>, names[7], 8); run(k_rate<8>, names[8], 8);
  run(k_rate<9>, names[9], 8); run(k_rate<10>, names[10], 8);
  {
    double t = timeit([&]{ hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, 1.0); });
    double n = (double)blocks*4*ITER*4; printf("mfma_f64_16x16x4: %.3f ms, %.2f cycles/mfma/SIMD, %.1f TFLOP/s\n", t*1e3, t*2.4e9/(n/(p.multiProcessorCount*4)), n*2048/t*1e-12);
    t = timeit([&]{ hipLaunchKernelGGL(k_mfma_valu, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9); });
    printf("mfma x2 + 16 fma_f64 per iter: %.3f ms -> %.2f cycles per iter per SIMD (mfma alone 2x, valu alone 16x)\n", t*1e3, t*2.4e9/((double)blocks*4*ITER/(p.multiProcessorCount*4)));
  }
  return 0;
}
