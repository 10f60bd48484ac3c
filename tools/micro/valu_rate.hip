// Developer microbenchmark (not part of the library): issue rate of v_exp_f32 / v_fma_f32 / v_pk_fma_f32 and whether
// transcendental and plain VALU work of one wave overlap on gfx950.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
  float c = 0.999f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      if (MODE == 2) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[(i + 8) & 15]) : "v"(c)); }
      if (MODE == 3) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                       asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[(i + 5) & 15]) : "v"(c));
                       asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[(i + 9) & 15]) : "v"(c));
                       asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[(i + 13) & 15]) : "v"(c)); }
      if (MODE == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int ops_per_iter) {
  float* d;
  hipMalloc(&d, 256 * 1024 * 4 * 4);
  const int iters = 20000;
  // one workgroup of 256 threads = 4 waves = 1 wave per SIMD; waves_per_simd workgroups per CU (256 CUs)
  dim3 grid(256 * waves_per_simd), block(256);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, 100);
  hipEventRecord(s);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  // cycles per wave-instruction per SIMD at 2.4 GHz
  double instr = (double)iters * ops_per_iter * waves_per_simd;
  printf("%-28s waves/SIMD %d : %.3f ms  -> %.2f cycles per wave-instruction per SIMD (2.4 GHz assumed)\n", name,
         waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 8; w *= 2) {
    run<0>("v_exp_f32", w, 16);
    run<4>("v_rcp_f32", w, 16);
    run<1>("v_fma_f32", w, 16);
    run<2>("exp+fma 1:1", w, 32);
    run<3>("exp+fma 1:3", w, 64);
  }
  return 0;
}
