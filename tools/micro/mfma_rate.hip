// Developer microbenchmark (not part of the library): sustained v_mfma_f32_32x32x16_bf16 rate on gfx950 with 1, 2 and 4
// waves per SIMD and no memory traffic -- the practical ceiling that the GEMM kernels' MFMA fractions should be read against.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int n = 0; n < NACC; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[n][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int wps) {
  float* d;
  (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 20000;
  dim3 grid(256 * wps), block(256);
  hipEvent_t s, e;
  (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, d, 100);
  (void)hipEventRecord(s);
  hipLaunchKernelGGL(k<NACC>, grid, block, 0, 0, d, iters);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms;
  (void)hipEventElapsedTime(&ms, s, e);
  const double flops = 2.0 * 32 * 32 * 16 * (double)NACC * iters * 4 * 256 * wps;
  printf("independent accumulators %2d, waves/SIMD %d: %.3f ms -> %.0f TFLOP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC, wps, ms,
         flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)NACC * iters * wps));
  (void)hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); run<10>(w); }
  return 0;
}
