// Fused feed-forward pair at C = 320 (the 72 x 128 level's  x -> GEGLU(x W1^T + b1) W2^T + b2 (+ residuals)  of
// BasicTransformerBlock.ff / TemporalBasicTransformerBlock.ff_in / .ff), gfx950.
//
// The two GEMM launches it replaces (ctrlv_gemm with geglu = 1, then the 1280 -> 320 projection) write the 4C-wide
// intermediate u to HBM and read it back: 2.4 GB per pair at M = 460 800, and the write-back of u is what the GEGLU kernel
// waits for (store experiments).  Here u never leaves the CU:
//   * 256-row tile per workgroup, 8 waves x 32 rows; a wave keeps its 32 x 320 OUTPUT accumulators (160 registers) for the
//     whole tile and its x rows as MFMA B fragments: k-steps 0..9 in registers (40), 10..19 in LDS (80 KB per workgroup)
//   * the hidden dimension is walked in CHUNKS of 16 columns: 20 MFMAs (x . W1 chunk, K = 320; the chunk's 16 value + 16
//     gate rows, bias as the C operand) -> 8 GEGLU results per lane (Phi table, common.h) -> packed to bf16 and used AS THEY
//     SIT as the B operand of 10 MFMAs (h chunk . W2 chunk).  In the result layout of v_mfma_f32_32x32x16 a lane holds, of
//     its row, hidden columns 8q + 4h + r (q = 0..1, r = 0..3, h = lane >> 5); as a B operand its eight values are k-slots
//     8h + 4q + r.  ctrlv_ff_fused_pack() stores W2 with its K index permuted accordingly, so no data moves between lanes.
//   * W1 / W2 chunks (20 + 10 KiB, fragment-major so that every ds_read_b128 is a contiguous KiB per wave) stream through a
//     2-slot LDS ring by LDS-DMA, one chunk ahead; one barrier per chunk.  All eight waves run in step: both waves of a
//     SIMD do GEMM 1 together, then GEGLU together (matrix pipe idle), then GEMM 2 -- which is why this version only ties
//     with the two launches (1.41-1.45 ms against 1.45-1.52 ms at M = 460 800; in the model +-0).  A stagger of the two
//     groups by a third of a chunk needs a third W2 slot, i.e. the bias strip out of LDS and three x fragments spilled
//     (measured: 1.55-1.61 ms): the LDS and the register file are both full.  OPT-IN (CTRLV_FF_FUSED=1) until it pays.
//   * epilogue = the ping-pong GEMM's (gemm_epilogue_lds: s_acc * acc + s1 R1 + s2 R2 + V, LDS transpose, 16-B stores)
// Arithmetic: the same MFMA, the same K order inside GEMM 1, the same bias-as-C-operand, the same GELU table and the same
// bf16 rounding of u as the two-launch path; GEMM 2 sums its K = 1280 in chunk order with the permuted slot assignment, so
// its fp32 sums differ from ctrlv_gemm's in the last bits.  Every C = 320 feed-forward of the inference path goes through
// this kernel whatever M is (the training forward keeps the two launches: it needs u and the raw projection).
#include "common.h"
#include "gemm_pp_kernel.h"

namespace {

constexpr int kC = 320, kHid = 1280, kChunks = kHid / 16;        // 80 chunks of 16 hidden columns
constexpr int kXHi = 8 * 10 * 1024;                              // x k-steps 10..19: [wave][ks][lane] x 16 B
constexpr int kW1Slot = 20 * 1024, kW2Slot = 10 * 1024, kSlot = kW1Slot + kW2Slot;
constexpr int kTabOff = kXHi + 2 * kSlot;
constexpr int kB1Off = kTabOff + kGeluTabBytes;                  // 2560 floats (interleaved order)
constexpr int kB2Off = kB1Off + 2 * kHid * 4;                    // 320 floats
constexpr int kSmem = kB2Off + kC * 4;
static_assert(kSmem <= 160 * 1024, "fused feed-forward tile does not fit the LDS");

struct FfArgs {
  const bf16_t* x; int ldx;
  const bf16_t* w1f; const float* b1; const bf16_t* w2f;
  ctrlv_gemm_desc o;            // the second projection's descriptor: out, bias (b2), R1, R2, scales, M, N = 320
  const float* vtab; int vdiv, vmod, ldv;   // row-vector operand V[(m / vdiv) % vmod] (vtab = nullptr: none), see below
};

template <int EPI>
__global__ __launch_bounds__(512) void ff_fused_kernel(const FfArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5;
  const ctrlv_gemm_desc& d = a.o;
  const int M = d.M;
  const int tiles = (M + 255) / 256, G = gridDim.x;

  gelu_table_fill(smem + kTabOff, threadIdx.x, 512);
  for (int i = threadIdx.x; i < 2 * kHid; i += 512) *(float*)(smem + kB1Off + i * 4) = a.b1[i];
  for (int i = threadIdx.x; i < kC; i += 512) *(float*)(smem + kB2Off + i * 4) = d.bias ? d.bias[i] : 0.f;

  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1f, 0, kChunks * kW1Slot, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2f, 0, kChunks * kW2Slot, 0x00020000);
  // this wave's pieces of a chunk: KiB number k * 8 + wid (k = 0..3) of the chunk's 30 (20 of W1, then 10 of W2)
  auto dma = [&](int chunk, int slot) {
    char* base = smem + kXHi + slot * kSlot;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pi = k * 8 + wid;                            // wave-uniform
      if (pi < 20)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, LDS_PTR(base + pi * 1024), 16, lane * 16, chunk * kW1Slot + pi * 1024, 0, 0);
      else if (pi < 30)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, LDS_PTR(base + pi * 1024), 16, lane * 16, chunk * kW2Slot + (pi - 20) * 1024, 0, 0);
    }
  };
  dma(0, 0);
  __syncthreads();                                           // table / bias strips visible to every wave
  const char* tab = smem + kTabOff;
  char* const xhi = smem + wid * 10 * 1024 + lane * 16;

  int cglob = 0;                                             // chunks processed so far by this workgroup (ring phase)
  for (int tile = blockIdx.x; tile < tiles; tile += G) {
    const int bm = tile * 256;
    const int m = bm + wid * 32 + r32;
    // ---- x rows of this wave as B fragments: k-step ks = columns ks*16 + 8*hsel .. +8 of row m
    bf16x8 xr[10];
    {
      const bf16_t* xp = a.x + (long)m * a.ldx + 8 * hsel;
      const bool ok = m < M;
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *(const uint4*)(xp + ks * 16);
        xr[ks] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int ks = 10; ks < 20; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *(const uint4*)(xp + ks * 16);
        *(uint4*)(xhi + (ks - 10) * 1024) = v;
      }
    }
    // output accumulators start from b2 -- plus the tile's row vector: V is constant over a tile (vdiv is a multiple of
    // 256, checked by the host; the frame positional embedding of ff_in: one vector per frame of S pixels) and s_acc is 1
    // there, so it rides in the accumulator instead of the epilogue
    f32x16 acc[1][10];
    const float* vrow = a.vtab ? a.vtab + (long)((bm / a.vdiv) % a.vmod) * a.ldv : nullptr;
#pragma unroll
    for (int n = 0; n < 10; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = *(const float4*)(smem + kB2Off + (n * 32 + 8 * q + 4 * hsel) * 4);
        if (vrow) {
          const float4 w = *(const float4*)(vrow + n * 32 + 8 * q + 4 * hsel);
          v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        acc[0][n][4 * q] = v.x; acc[0][n][4 * q + 1] = v.y; acc[0][n][4 * q + 2] = v.z; acc[0][n][4 * q + 3] = v.w;
      }
    // the first chunk's weights (issued before the previous tile's last barrier, or in the prologue) must have landed,
    // and every wave's x_hi writes must be visible
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < kChunks; ++c, ++cglob) {
      const int slot = cglob & 1;
      const char* st = smem + kXHi + slot * kSlot + lane * 16;
      dma(c + 1 < kChunks ? c + 1 : 0, slot ^ 1);            // next chunk (the next tile starts at chunk 0 again)
      f32x16 a1;                                             // GEMM 1 starts from the chunk's bias (C operand)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *(const float4*)(smem + kB1Off + (c * 32 + 8 * q + 4 * hsel) * 4);
        a1[4 * q] = v.x; a1[4 * q + 1] = v.y; a1[4 * q + 2] = v.z; a1[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        const bf16x8 wf = *(const bf16x8*)(st + ks * 1024);
        const bf16x8 xf = ks < 10 ? xr[ks] : *(const bf16x8*)(xhi + (ks - 10) * 1024);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, a1, 0, 0, 0);
      }
      // GEGLU in the result layout: accumulators 0..7 are the 8 value columns of this lane, 8..15 their gates
      float h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = geglu_tab(a1[e], a1[8 + e], tab);
      const uint4 hp = make_uint4(pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3]), pack_bf16x2(h[4], h[5]),
                                  pack_bf16x2(h[6], h[7]));
      const bf16x8 hf = __builtin_bit_cast(bf16x8, hp);
#pragma unroll
      for (int n = 0; n < 10; ++n) {
        const bf16x8 wf = *(const bf16x8*)(st + kW1Slot + n * 1024);
        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hf, acc[0][n], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // (x fragments are dead here: end their live ranges so that the epilogue's prefetch window gets their registers)
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) asm volatile("" : "=v"(xr[ks]));
    // ---- epilogue: the ping-pong GEMM's, on this wave's 32 x 320 block; staging = four KiB of the wave's x_hi strip
    char* stg = smem + wid * 10 * 1024;
    int lane_e = lane;                                       // (opaque copy: keeps the epilogue's lane constants per-tile values
    asm volatile("" : "+v"(lane_e));                         //  instead of hoisted, spilled ones -- gemm_pp_kernel.h)
    gemm_epilogue_lds<1, 10, false, EPI>(d, acc, bm, 0, wid, 0, 32, kC, lane_e, stg, stg + 1024, stg + 2048, stg + 3072,
                                         nullptr, tab);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // staging reads done before the next tile's x_hi writes
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead DMA of the last chunk: nothing may be in flight
#endif
}

// fragment-major copies of the two packed weights (device-side permutation of bf16 values, once per weight)
__global__ void ff_pack_kernel(const bf16_t* __restrict__ w1p, const bf16_t* __restrict__ w2p, bf16_t* __restrict__ w1f,
                               bf16_t* __restrict__ w2f) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n1 = (long)kChunks * 20 * 64 * 8, n2 = (long)kChunks * 10 * 64 * 8;
  if (i < n1) {
    // w1f[chunk][ks][lane][j] = w1p[chunk*32 + lane%32][ks*16 + 8*(lane/32) + j]
    const int j = i & 7, lane = (i >> 3) & 63;
    const int ks = (int)((i >> 9) % 20), chunk = (int)(i / (20 * 512));
    w1f[i] = w1p[(long)(chunk * 32 + (lane & 31)) * kC + ks * 16 + 8 * (lane >> 5) + j];
  } else if (i < n1 + n2) {
    // w2f[chunk][n][lane][j] = w2p[n*32 + lane%32][chunk*16 + 8*(j/4) + 4*(lane/32) + j%4]   (k-slot 8h + j <-> column)
    const long k = i - n1;
    const int j = k & 7, lane = (k >> 3) & 63;
    const int n = (int)((k >> 9) % 10), chunk = (int)(k / (10 * 512));
    w2f[k] = w2p[(long)(n * 32 + (lane & 31)) * kHid + chunk * 16 + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)];
  }
}

template <int EPI>
int launch_ff(const FfArgs& a, hipStream_t stream) {
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  auto kfn = ff_fused_kernel<EPI>;
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
    attr_set[dev] = true;
  }
  const int num_cu = ctrlv_num_cu(dev);
  const int tiles = (a.o.M + 255) / 256;
  int grid = tiles;
  if (tiles > num_cu) {                     // persistent, every workgroup the same number of tiles
    const int rounds = (tiles + num_cu - 1) / num_cu;
    grid = (tiles + rounds - 1) / rounds;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), kSmem, stream, a);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

}  // namespace

extern "C" int ctrlv_ff_fused_pack(const void* w1_packed, const void* w2_packed, void* w1f, void* w2f,
                                   ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(w1_packed && w2_packed && w1f && w2f, "ctrlv_ff_fused_pack: null pointer");
  const long n = (long)kChunks * 30 * 64 * 8;
  hipLaunchKernelGGL(ff_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)w1_packed, (const bf16_t*)w2_packed, (bf16_t*)w1f, (bf16_t*)w2f);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_ff_fused(const void* x, int ldx, const void* w1f, const float* b1, const void* w2f,
                              const ctrlv_gemm_desc* out_desc, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && w1f && b1 && w2f && out_desc && out_desc->out, "ctrlv_ff_fused: null pointer");
  FfArgs a;
  a.x = (const bf16_t*)x; a.ldx = ldx; a.w1f = (const bf16_t*)w1f; a.b1 = b1; a.w2f = (const bf16_t*)w2f;
  a.o = *out_desc;
  a.vtab = nullptr; a.vdiv = 1; a.vmod = 1; a.ldv = 0;
  if (a.o.vmode) {
    // the row vector is folded into the accumulator start (kernel): one vector per 256-row tile, unscaled
    CTRLV_CHECK_ARG(a.o.vmode == 1 && a.o.V && a.o.vdiv > 0 && a.o.vdiv % 256 == 0 && a.o.vmod > 0 && a.o.s_acc == 1.0f &&
                        a.o.ldv % 4 == 0,
                    "ctrlv_ff_fused: a row-vector operand needs vmode 1, vdiv a multiple of 256 and s_acc == 1 "
                    "(ctrlv_ff_fused_serves() tells; use the two ctrlv_gemm launches otherwise)");
    a.vtab = a.o.V; a.vdiv = a.o.vdiv; a.vmod = a.o.vmod; a.ldv = a.o.ldv;
    a.o.vmode = 0; a.o.V = nullptr;
  }
  const ctrlv_gemm_desc& d = a.o;
  CTRLV_CHECK_SHAPE(d.M > 0 && d.N == kC && d.Cin == kHid && d.taps == 1 && d.mode == 0 && ldx >= kC && ldx % 8 == 0,
                    "ctrlv_ff_fused: serves M x 320 <- 1280 <- 320 only (N=%d Cin=%d ldx=%d)", d.N, d.Cin, ldx);
  CTRLV_CHECK_ARG(!d.geglu && !d.act && !d.out_f32 && !d.raw_out && !d.A2 && d.n_scale2 == 0,
                  "ctrlv_ff_fused: plain bf16 output only");
  CTRLV_CHECK_SHAPE(d.n_store == kC && d.ldo % 8 == 0 && (!d.R1 || d.ldr1 % 8 == 0) && (!d.R2 || d.ldr2 % 8 == 0),
                    "ctrlv_ff_fused: n_store must be 320 and the row pitches multiples of 8");
  const long lim = 0xFFFFFFF0L;
  CTRLV_CHECK_SHAPE((long)d.M * d.ldo * 2 <= lim && (!d.R1 || (long)d.M * d.ldr1 * 2 <= lim) &&
                        (!d.R2 || (long)d.M * d.ldr2 * 2 <= lim),
                    "ctrlv_ff_fused: operands beyond 32-bit byte offsets");
  hipStream_t st = (hipStream_t)stream;
  switch (pp_epi_of(d)) {
    case 0: return launch_ff<0>(a, st);
    case 2: return launch_ff<2>(a, st);
    case 6: return launch_ff<6>(a, st);
    default: break;
  }
  ctrlv_set_error("ctrlv_ff_fused: epilogue operand combination not served (bias, +R1, +R1+R2, each with an optional V)");
  return CTRLV_E_BAD_ARG;
}

// 1 if ctrlv_ff_fused serves a second-projection descriptor with these row-vector settings (the callers' switch between
// the fused kernel and the two ctrlv_gemm launches)
extern "C" int ctrlv_ff_fused_serves(int n, int cin, int vmode, int vdiv, float s_acc, int has_r1, int has_r2) {
  if (n != kC || cin != kHid) return 0;
  if (has_r2 && !has_r1) return 0;
  if (vmode == 0) return 1;
  return vmode == 1 && vdiv > 0 && vdiv % 256 == 0 && s_acc == 1.0f;
}
