// Developer microbenchmark (not part of the library): what a GEMM epilogue's output stores cost the memory side as a function of
// how many contiguous bytes of a row ONE store instruction covers.  A 460800 x 320 bf16 output (the 72 x 128 level, 640-byte
// rows) is written once per launch by 1800 "tiles" of 256 x 320 on persistent 512-thread workgroups, each wave its 64 x 160
// part with buffer-less 16-byte stores, as
//   LPR = 2: 32 rows x 32 B per instruction  (the v_permlane32_swap epilogue, profiles/r04_epilogue_permlane_ab.txt)
//   LPR = 4: 16 rows x 64 B                  (the LDS-staged epilogue of the ping-pong kernels)
//   LPR = 8:  8 rows x 128 B                 (two sub-tiles staged side by side: whole cache lines)
//   LPR = 20: 3.2 rows x 320 B               (a wave's whole 160-column row segment)
// build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int LPR>
__global__ __launch_bounds__(512) void k(uint4* __restrict__ out, int tiles) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  constexpr int RPI = 64 / LPR;                       // rows per instruction (LPR = 20: 3 rows, 4 lanes idle)
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long row0 = (long)t * 256 + wr * 64;
    const uint4 v = make_uint4(t, lane, wid, 7);
    if constexpr (LPR == 20) {
      const int r = lane / 20, c = lane % 20;
      for (int p = 0; p < 64; p += 3)
        if (lane < 60 && p + r < 64) out[(row0 + p + r) * 40 + wc * 20 + c] = v;
    } else {
      constexpr int CB = 20 / LPR;                    // column blocks of LPR uint4 (LPR = 8: two blocks + a 4-wide rest)
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int p = 0; p < 64; p += RPI) out[(row0 + p + lane / LPR) * 40 + wc * 20 + j * LPR + lane % LPR] = v;
      if constexpr (LPR == 8) {
#pragma unroll
        for (int p = 0; p < 64; p += 16) out[(row0 + p + lane / 4) * 40 + wc * 20 + 16 + lane % 4] = v;
      }
    }
  }
}

template <int LPR>
void run(uint4* buf, int tiles) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<LPR>, dim3(225), dim3(512), 0, 0, buf, tiles);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<LPR>, dim3(225), dim3(512), 0, 0, buf, tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%3d lanes per row (%4d B per row and instruction): %7.1f us per launch, %5.2f TB/s\n", LPR, LPR * 16, ms / n * 1e3,
         (double)tiles * 256 * 640 / (ms / n * 1e-3) / 1e12);
}

int main() {
  const int tiles = 1800;
  uint4* buf;
  hipMalloc(&buf, (size_t)tiles * 256 * 640 * 3);
  for (int rep = 0; rep < 2; ++rep) {
    run<2>(buf, tiles); run<4>(buf, tiles); run<8>(buf, tiles); run<20>(buf, tiles);
  }
  return 0;
}
