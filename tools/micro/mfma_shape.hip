// Developer microbenchmark (not part of the library): the SAME wave tile (64 x 160 outputs per wave, K advanced 32 per
// iteration, every operand fragment re-read from LDS by ds_read_b128, 2 waves per SIMD, every CU busy, random bf16 data)
// computed with v_mfma_f32_32x32x16_bf16 (20 per iteration) and with v_mfma_f32_16x16x32_bf16 (40 per iteration).
// MI355X_MICROARCH.md "DVFS give-back" item 7 reports the 16x16x32 form at 1.12-1.15x the FLOP/s under load at equal
// cycles per FLOP (the chip holds a higher clock): this checks it on the box at hand, in the GEMM kernels' regime, and
// prints the in-kernel clock (s_memtime / s_memrealtime) next to the rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kRows = 576, kSlot = kRows * 64, kSlots = 4;   // the 256x320 tile's ring: 4 x (256 + 320) rows of 64 B

template <int SHAPE, int NT, int ORDER = 0>
__global__ __launch_bounds__(NT) void k(const unsigned short* __restrict__ src, float* out, unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  for (int i = threadIdx.x; i < kSlots * kSlot / 16; i += NT) ((uint4*)smem)[i] = ((const uint4*)src)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;
    const int r32 = lane & 31, hsel = lane >> 5, sw = (r32 >> 2) & 3;
    const int a_frag = (wr * 64 + r32) * 64, b_frag = 256 * 64 + (wc * 160 + r32) * 64;
    for (int it = 0; it < iters; ++it) {
      const char* st = smem + (it & 3) * kSlot;
      bf16x8 af[2][2], wf[5][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int co = ((ks * 2 + hsel) ^ sw) * 16;
#pragma unroll
        for (int n = 0; n < 5; ++n) wf[n][ks] = *(const bf16x8*)(st + b_frag + n * 2048 + co);
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i][ks] = *(const bf16x8*)(st + a_frag + i * 2048 + co);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 5; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n][ks], af[i][ks], acc[i][n], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][n][e];
  } else {
    f32x4 acc[4][10];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 10; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][n][e] = 0.f;
    const int r16 = lane & 15, g = lane >> 4;
    // conflict-free chunk permutation for this access shape: physical = logical ^ perm[(row >> 2) & 3], perm = {0, 2, 3, 1}
    const int sw = (0x1320 >> (((r16 >> 2) & 3) * 4)) & 3;
    const int co = (g ^ sw) * 16;
    const int a_frag = (wr * 64 + r16) * 64 + co, b_frag = 256 * 64 + (wc * 160 + r16) * 64 + co;
    for (int it = 0; it < iters; ++it) {
      const char* st = smem + (it & 3) * kSlot;
      bf16x8 af[4], wf[10];
#pragma unroll
      for (int n = 0; n < 10; ++n) wf[n] = *(const bf16x8*)(st + b_frag + n * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(st + a_frag + i * 1024);
      if constexpr (ORDER == 0) {          // i outer: 10 consecutive MFMAs share the B operand (af[i])
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < 10; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[i], acc[i][n], 0, 0, 0);
      } else if constexpr (ORDER == 1) {   // n outer: 4 consecutive MFMAs share the A operand (wf[n])
#pragma unroll
        for (int n = 0; n < 10; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[i], acc[i][n], 0, 0, 0);
      } else {                             // diagonal: consecutive MFMAs share no operand
#pragma unroll
        for (int t = 0; t < 40; ++t) {
          const int i = t & 3, n = (t + (t >> 2) * 3) % 10;
          acc[i][(n + i * 0) % 10] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[i], acc[i][n], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 10; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc[i][n][e];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}


// 16 waves per workgroup (4 per SIMD, <= 128 VGPRs each), wave tile 64 x 80 (4 x 5 blocks of 16 x 16): the register budget of
// a 4-waves-per-SIMD ping-pong (two waves of every SIMD in their MFMA phase at any time).  9 fragment reads per 20 MFMAs.
__global__ __launch_bounds__(1024) void k16w(const unsigned short* __restrict__ src, float* out, unsigned long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  for (int i = threadIdx.x; i < kSlots * kSlot / 16; i += 1024) ((uint4*)smem)[i] = ((const uint4*)src)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x4 acc[4][5];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][n][e] = 0.f;
  const int r16 = lane & 15, g = lane >> 4;
  const int sw = (0x1320 >> (((r16 >> 2) & 3) * 4)) & 3;
  const int co = (g ^ sw) * 16;
  const int a_frag = (wr * 64 + r16) * 64 + co, b_frag = 256 * 64 + (wc * 80 + r16) * 64 + co;
  for (int it = 0; it < iters; ++it) {
    const char* st = smem + (it & 3) * kSlot;
    bf16x8 af[4], wf[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) wf[n] = *(const bf16x8*)(st + b_frag + n * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(st + a_frag + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < 5; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[i], acc[i][n], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < 5; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[i][n][e];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

void run16w(const unsigned short* src, float* out, unsigned long long* clk, int iters) {
  const int smem = kSlots * kSlot;
  (void)hipFuncSetAttribute((const void*)k16w, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t s, e;
  (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  (void)hipEventRecord(s);
  hipLaunchKernelGGL(k16w, dim3(256), dim3(1024), smem, 0, src, out, clk, iters);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms;
  (void)hipEventElapsedTime(&ms, s, e);
  static unsigned long long h[512];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double ghz = 0;
  for (int b = 0; b < 256; ++b) ghz += (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
  ghz /= 256;
  const double flops = 2.0 * 64 * 80 * 32 * (double)iters * 16 * 256;
  printf("16 waves (4/SIMD), wave tile 64x80, 16x16x32: %8.3f ms  %7.0f TFLOP/s  in-kernel clock %.3f GHz  -> %.2f cycles per iteration per SIMD (matrix pipe alone: 1280)\n",
         ms, flops / ms / 1e9, ghz, ms * 1e-3 * ghz * 1e9 / iters);
}

template <int SHAPE, int NT, int ORDER = 0>
double run(const unsigned short* src, float* out, unsigned long long* clk, int iters, bool print) {
  const int smem = kSlots * kSlot;
  (void)hipFuncSetAttribute((const void*)k<SHAPE, NT, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t s, e;
  (void)hipEventCreate(&s); (void)hipEventCreate(&e);
  (void)hipEventRecord(s);
  hipLaunchKernelGGL((k<SHAPE, NT, ORDER>), dim3(256), dim3(NT), smem, 0, src, out, clk, iters);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms;
  (void)hipEventElapsedTime(&ms, s, e);
  static unsigned long long h[512];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double ghz = 0;
  for (int b = 0; b < 256; ++b) ghz += (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
  ghz /= 256;
  const double flops = 2.0 * 64 * 160 * 32 * (double)iters * (NT / 64) * 256;
  if (print)
    printf("shape %2d order %d, %d waves/SIMD: %8.3f ms  %7.0f TFLOP/s  in-kernel clock %.3f GHz  -> %.2f cycles per iteration per SIMD (matrix pipe alone: %d)\n", SHAPE,
           NT / 256, ms, flops / ms / 1e9, ghz, ms * 1e-3 * ghz * 1e9 / iters, 640 * (NT / 256));
  return flops / ms / 1e9;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 40000;
  const int n = kSlots * kSlot / 2;
  unsigned short* h = (unsigned short*)malloc(n * 2);
  srand(1);
  for (int i = 0; i < n; ++i) {   // uniform [-1, 1) as bf16
    float f = (float)rand() / RAND_MAX * 2.f - 1.f;
    unsigned u; memcpy(&u, &f, 4);
    h[i] = (unsigned short)(u >> 16);
  }
  unsigned short* src; float* out; unsigned long long* clk;
  (void)hipMalloc(&src, n * 2); (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&clk, 512 * 8);
  (void)hipMemcpy(src, h, n * 2, hipMemcpyHostToDevice);
  run<32, 512>(src, out, clk, 2000, false); run<16, 512>(src, out, clk, 2000, false);
  for (int r = 0; r < 3; ++r) {      // interleaved rounds on one device (guide rule 24); each launch runs ~0.1-0.2 s
    run<32, 512>(src, out, clk, iters, true);
    run<16, 512>(src, out, clk, iters, true);
    run<32, 256>(src, out, clk, iters, true);      // one wave per SIMD: what a ping-pong schedule's compute phase gets
    run<16, 256>(src, out, clk, iters, true);
    run<16, 256, 1>(src, out, clk, iters, true);
    run<16, 256, 2>(src, out, clk, iters, true);
    run<16, 512, 1>(src, out, clk, iters, true);
    run16w(src, out, clk, iters);
  }
  return 0;
}
