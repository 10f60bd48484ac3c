// Developer microbenchmark (not part of the library): the instruction mix of a FUSED feed-forward pair at C = 320
// (GEGLU projection 320 -> 2 x 1280, then 1280 -> 320) in which the 4C-wide intermediate never leaves the CU:
//   256-row tile per workgroup, 8 waves x 32 rows; per 16-column chunk of the intermediate and per wave:
//   20 MFMAs (x . W1 chunk, K = 320) -> 8 GEGLU outputs per lane (table GELU) -> packed straight into the B operand of
//   10 MFMAs (h chunk . W2 chunk -> 32 x 320 output accumulators, 160 registers, live for the whole tile).
//   x: k-steps 0..9 in registers (40), 10..19 in LDS (80 KB); W1 / W2 chunks (20 + 10 KB) stream through a 2-slot
//   LDS ring by LDS-DMA (30 one-KiB pieces per chunk, 8 waves).
// Question: what MFMA rate does this mix sustain (random data, all CUs, 2 waves per SIMD) compared with the ~1.5 ms the two
// separate GEMM launches take for M = 460 800 (1.13 TFLOP)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(unsigned)(unsigned long long)(p))

constexpr int X_HI = 8 * 10 * 1024;            // 80 KB: [wave][ks 10..19][32 rows][32 B]
constexpr int W1_SLOT = 20 * 1024, W2_SLOT = 10 * 1024, SLOT = W1_SLOT + W2_SLOT + 2048;   // + 2 dummy KiB (32 pieces / 8 waves)
constexpr int TAB = X_HI + 2 * SLOT;
constexpr int SMEM = TAB + 8192;

__device__ __forceinline__ float geglu1(float a, float g, const char* tab) {
  float t = __builtin_fmaf(g, 100.0f, 512.0f);
  t = __builtin_amdgcn_fmed3f(t, 0.0f, 1023.99994f);
  const float fr = __builtin_amdgcn_fractf(t);
  const float2 e = *(const float2*)(tab + (int)t * 8);
  return a * (g * __builtin_fmaf(fr, e.y, e.x));
}

template <int MODE>   // 0 = chunk by chunk; 1 = GEGLU + second GEMM of chunk c interleaved with the first GEMM of chunk c + 1
__global__ __launch_bounds__(512) void k(const unsigned short* __restrict__ w, const unsigned short* __restrict__ x, float* out,
                                         unsigned long long* clk, int chunks) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r32 = lane & 31, hsel = lane >> 5;
  for (int i = threadIdx.x; i < 1024; i += 512) *(float2*)(smem + TAB + i * 8) = make_float2(0.5f + 0.0004f * (i - 512), 0.0004f);
  // x: this wave's 32 rows; k-steps 0..9 into registers, 10..19 into LDS
  bf16x8 xr[10];
  const unsigned short* xw = x + ((long)blockIdx.x * 8 + wid) * (32 * 320);
#pragma unroll
  for (int ks = 0; ks < 10; ++ks) xr[ks] = *(const bf16x8*)(xw + ks * 512 + lane * 8);
  for (int ks = 10; ks < 20; ++ks)
    *(bf16x8*)(smem + (wid * 10 + ks - 10) * 1024 + lane * 16) = *(const bf16x8*)(xw + ks * 512 + lane * 8);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0x7fffffff, 0x00020000);
  auto dma = [&](int chunk, int piece) {   // piece 0..3 of this wave: KiB number piece * 8 + wid of the chunk's 32 (30 real)
    char* dst = smem + X_HI + (chunk & 1) * SLOT + (piece * 8 + wid) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst), 16, lane * 16, ((chunk % 80) * 30 + (piece * 8 + wid) % 30) * 1024, 0, 0);
  };
#pragma unroll
  for (int p = 0; p < 4; ++p) dma(0, p);
  f32x16 acc[10];
#pragma unroll
  for (int n = 0; n < 10; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const char* tab = smem + TAB;
  const char* xhi = smem + wid * 10 * 1024 + lane * 16;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (MODE == 0) {
    for (int c = 0; c < chunks; ++c) {
      const char* st = smem + X_HI + (c & 1) * SLOT + lane * 16;
      f32x16 a1;
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        const bf16x8 wf = *(const bf16x8*)(st + ks * 1024);
        const bf16x8 xf = ks < 10 ? xr[ks] : *(const bf16x8*)(xhi + (ks - 10) * 1024);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, a1, 0, 0, 0);
        if (ks % 5 == 0) dma(c + 1, ks / 5);
      }
      float h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = geglu1(a1[e], a1[8 + e], tab);
      bf16x8 hf;
#pragma unroll
      for (int e = 0; e < 8; ++e) hf[e] = (__bf16)h[e];
#pragma unroll
      for (int n = 0; n < 10; ++n) {
        const bf16x8 wf = *(const bf16x8*)(st + W1_SLOT + n * 1024);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hf, acc[n], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    // software pipeline: a1 of chunk c + 1 is accumulated while chunk c's GEGLU (VALU) and second GEMM run; the W2 part of a
    // slot is read one chunk late, so the ring here has its W2 halves in a third region (reuses the dummy KiBs' neighbourhood)
    f32x16 a1;
#pragma unroll
    for (int e = 0; e < 16; ++e) a1[e] = 0.f;
    {
      const char* st = smem + X_HI + lane * 16;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        const bf16x8 wf = *(const bf16x8*)(st + ks * 1024);
        const bf16x8 xf = ks < 10 ? xr[ks] : *(const bf16x8*)(xhi + (ks - 10) * 1024);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, a1, 0, 0, 0);
      }
    }
    for (int c = 0; c < chunks; ++c) {
      const char* st = smem + X_HI + (c & 1) * SLOT + lane * 16;
      const char* sn = smem + X_HI + ((c + 1) & 1) * SLOT + lane * 16;
      const f32x16 a0 = a1;
#pragma unroll
      for (int p = 0; p < 4; ++p) dma(c + 1, p);     // (timing only: overwrites the slot being read by the look-ahead GEMM)
      float h[8];
      bf16x8 hf;
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        const bf16x8 wf = *(const bf16x8*)(sn + ks * 1024);
        const bf16x8 xf = ks < 10 ? xr[ks] : *(const bf16x8*)(xhi + (ks - 10) * 1024);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, a1, 0, 0, 0);
        if (ks < 8) h[ks] = geglu1(a0[ks], a0[8 + ks], tab);
        if (ks == 8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) hf[e] = (__bf16)h[e];
        }
        if (ks >= 10) {
          const int n = ks - 10;
          const bf16x8 w2 = *(const bf16x8*)(st + W1_SLOT + n * 1024);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, hf, acc[n], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < 10; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[n][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  const int nwg = 256 * 4, chunks = 80 * 8;       // 80 chunks = one 256-row tile at C = 320
  unsigned short *w, *x; float* out; unsigned long long* clk;
  const size_t wbytes = 80 * 30 * 1024 + (1 << 20), xbytes = (size_t)nwg * 8 * 32 * 320 * 2;
  hipMalloc(&w, wbytes); hipMalloc(&x, xbytes); hipMalloc(&out, nwg * 512 * 4); hipMalloc(&clk, nwg * 16);
  unsigned short* h = (unsigned short*)malloc(xbytes > wbytes ? xbytes : wbytes);
  srand(1);
  for (size_t i = 0; i < wbytes / 2; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(w, h, wbytes, hipMemcpyHostToDevice);
  for (size_t i = 0; i < xbytes / 2; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(x, h, xbytes, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    auto fn = mode == 0 ? k<0> : k<1>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(fn, dim3(nwg), dim3(512), SMEM, 0, w, x, out, clk, chunks);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double flops = (double)nwg * 8 * chunks * 30 * (32.0 * 32 * 16 * 2);
      printf("mode %d: %.3f ms  %.0f TFLOP/s   %.0f cycles per chunk (30 MFMAs per wave, 2 waves per SIMD: 1920 = 100%%)  clock %.2f GHz\n",
             mode, ms, flops / ms / 1e9, (double)c[0] / chunks, (double)c[0] / (c[1] * 10.0) );
    }
  }
  printf("smem %d bytes; %s\n", SMEM, hipGetErrorString(hipGetLastError()));
  return 0;
}
