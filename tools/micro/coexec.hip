// Developer microbenchmark: do MFMA and VALU / SALU / LDS instructions of the two waves sharing a SIMD overlap on gfx950?
// One 512-thread workgroup per CU (waves w and w+4 share a SIMD).  Roles per wave: M = 16 independent MFMA 32x32x16 per
// iteration, V = 64 independent v_fma_f32, S = 64 s_add_u32, L = 16 ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void role_m(f32x16 (&acc)[8], bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
}
__device__ __forceinline__ void role_v(float (&v)[16], float c) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
}
__device__ __forceinline__ void role_s(unsigned& s) {
#pragma unroll
  for (int r = 0; r < 64; ++r) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
}
__device__ __forceinline__ void role_l(const char* lds, f32x4 (&l)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) l[i] = *(const f32x4*)(lds + i * 1024);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// mode: role of waves 0-3 in the high nibble, role of waves 4-7 in the low nibble: 0 idle, 1 M, 2 V, 3 S, 4 L, 5 = M and V interleaved
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int wid = threadIdx.x >> 6;
  const int role = wid < 4 ? (mode >> 4) : (mode & 15);
  f32x16 acc[8];
  bf16x8 a, b;
  float v[16];
  f32x4 l[16];
  unsigned s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x + i; l[i] = f32x4{0, 0, 0, 0}; }
  for (int i = threadIdx.x; i < 8192; i += 512) ((float*)lds)[i] = i;
  __syncthreads();
  const char* lp = lds + (threadIdx.x & 63) * 16;
  if (role == 1) for (int it = 0; it < iters; ++it) role_m(acc, a, b);
  if (role == 2) for (int it = 0; it < iters; ++it) role_v(v, 0.999f);
  if (role == 3) for (int it = 0; it < iters; ++it) role_s(s);
  if (role == 4) for (int it = 0; it < iters; ++it) role_l(lp, l);
  if (role == 5) for (int it = 0; it < iters; ++it) { role_m(acc, a, b); role_v(v, 0.999f); }
  float r = (float)s;
#pragma unroll
  for (int n = 0; n < 8; ++n) r += acc[n][0] + acc[n][7];
#pragma unroll
  for (int i = 0; i < 16; ++i) r += v[i] + l[i].x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 512 * 4);
  const int iters = 4000;
  struct { const char* name; int mode; } cases[] = {
      {"M | idle", 0x10}, {"V | idle", 0x20}, {"L | idle", 0x40}, {"M | M", 0x11}, {"V | V", 0x22},
      {"M | V", 0x12},    {"M | L", 0x14},    {"M+V same wave | idle", 0x50}, {"M+V | M+V", 0x55}, {"V | L", 0x24}};
  for (auto& c : cases) {
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, 10, c.mode);
    (void)hipEventRecord(s);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, c.mode);
    (void)hipEventRecord(e);
    (void)hipEventSynchronize(e);
    float ms;
    (void)hipEventElapsedTime(&ms, s, e);
    printf("%-24s %.3f ms  -> %.0f cycles per iteration at 2.4 GHz\n", c.name, ms, ms * 1e-3 * 2.4e9 / iters);
    fflush(stdout);
  }
  return 0;
}
