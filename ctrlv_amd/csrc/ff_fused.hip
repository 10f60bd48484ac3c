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
//   * W1 / W2 chunks (21 + 10 KiB, fragment-major so that every ds_read_b128 is a contiguous KiB per wave) stream through
//     LDS rings by LDS-DMA, two chunks ahead of the window boundary; one barrier per chunk.  The two waves of a SIMD run a
//     third of a chunk apart (see the kernel) so that one's GEGLU runs beside the other's MFMAs; in step, the first version
//     only tied with the two launches (1.35-1.45 ms at M = 460 800; this one 1.29-1.30 against 1.44-1.52; -1.35 % per step)
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
constexpr int kW1Pieces = 21;                                    // 20 k-steps of K = 320 + one carrying the bias (see below)
constexpr int kW1Slot = kW1Pieces * 1024, kW2Slot = 10 * 1024;
constexpr int kW1Off = kXHi;                                     // W1 ring: 2 slots (chunk c, chunk c + 1 in flight)
constexpr int kW2Off = kW1Off + 2 * kW1Slot;                     // W2 ring: 3 slots (c - 1 for the late group, c, c + 1)
constexpr int kTabOff = kW2Off + 3 * kW2Slot;
constexpr int kSmem = kTabOff + kGeluTabBytes;
static_assert(kSmem <= 160 * 1024, "fused feed-forward tile does not fit the LDS");

struct FfArgs {
  const el_t* x; int ldx;
  const el_t* w1f; const el_t* w2f;
  ctrlv_gemm_desc o;            // the second projection's descriptor: out, bias (b2), R1, R2, scales, M, N = 320
  const float* vtab; int vdiv, vmod, ldv;   // row-vector operand V[(m / vdiv) % vmod] (vtab = nullptr: none), see below
  // optional LayerNorm of the input rows (ln_g = nullptr: x is used as it is): x' = LN(x + lnv[(m / ln_vdiv) % ln_vmod])
  const float* ln_g; const float* ln_b; float ln_eps;
  const float* lnv; int ln_vdiv, ln_vmod, ln_ldv;
};

template <int EPI, bool LO = false>
__global__ __launch_bounds__(512) void ff_fused_kernel(const FfArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5;
  const ctrlv_gemm_desc& d = a.o;
  const int M = d.M;
  const int tiles = (M + 255) / 256, G = gridDim.x;

  CTRLV_CLOCK_BEGIN();
  gelu_table_fill(smem + kTabOff, threadIdx.x, 512);

  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1f, 0, kChunks * kW1Slot, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2f, 0, kChunks * kW2Slot, 0x00020000);
  // The 31 KiB pieces of a chunk (21 of W1, then 10 of W2) are all issued by waves 0-3 (piece k * 4 + wid, k = 0..7): by
  // the stamps (tools/ff_stamp.py) that group reached every window boundary ~900 cycles before waves 4-7 and waited at
  // the barrier, while waves 4-7 -- the arbitration losers of their SIMDs -- spent 540 cycles issuing their four pieces
  // right behind it (waves 0-3: 230).  `cg` = the workgroup's running chunk count (ring phase), `chunk` = which of the 80.
  auto dma = [&](int chunk, int cg) {
    if (wid >= 4) return;                                    // (wave-uniform)
    char* s1 = smem + kW1Off + (cg & 1) * kW1Slot;
    char* s2 = smem + kW2Off + (cg % 3) * kW2Slot;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pi = k * 4 + wid;
      if (pi < kW1Pieces)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, LDS_PTR(s1 + pi * 1024), 16, lane * 16, chunk * kW1Slot + pi * 1024, 0, 0);
      else if (pi < kW1Pieces + 10)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, LDS_PTR(s2 + (pi - kW1Pieces) * 1024), 16, lane * 16,
                                                 chunk * kW2Slot + (pi - kW1Pieces) * 1024, 0, 0);
    }
  };
  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                           // chunk 0 and the table visible to every wave
  dma(1, 1);                                                 // (every later DMA is issued at a window boundary)
  const char* tab = smem + kTabOff;
  char* const xhi = smem + wid * 10 * 1024 + lane * 16;
  // The two waves of a SIMD (w, w + 4) run A THIRD OF A CHUNK APART: between two barriers group 0 does [GEMM 1, GEGLU,
  // GEMM 2] of chunk c, group 1 [GEMM 2 of chunk c - 1, GEMM 1, GEGLU of chunk c] -- so one wave's GEGLU (VALU + table
  // reads) runs beside its partner's MFMAs instead of beside its partner's GEGLU (in step, the matrix pipe idled through
  // both: the first version of this kernel only tied with the two launches).  W2 chunks therefore live for two windows
  // (3-slot ring).  The LDS for the third slot comes from the GEMM-1 bias: it is not a 10 KB strip read as the C operand
  // but a 21st K step -- the packed W1 chunk carries (bf16(b), bf16(b - bf16(b))) in two K slots against a constant-one x
  // fragment, which adds b to within 2^-17 |b| in the fp32 accumulator -- and from b2, read from global once per tile.
  const int grp = wid >> 2;
  const uint4 xone_u = hsel == 0 ? make_uint4(CTRLV_ELEM_DTYPE == 1 ? 0x3C003C00u : 0x3F803F80u, 0, 0, 0) : make_uint4(0, 0, 0, 0);
  const elx8 xone = __builtin_bit_cast(elx8, xone_u);

  int cglob = 0;                                             // chunks processed so far by this workgroup (ring phase)
  for (int tile = blockIdx.x; tile < tiles; tile += G) {
    const int bm = tile * 256;
    const int m = bm + wid * 32 + r32;
    // ---- x rows of this wave as B fragments: k-step ks = columns ks*16 + 8*hsel .. +8 of row m
    elx8 xr[10];
    {
      const el_t* xp = a.x + (long)m * a.ldx + 8 * hsel;
      const bool ok = m < M;
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *(const uint4*)(xp + ks * 16);
        xr[ks] = __builtin_bit_cast(elx8, v);
      }
#pragma unroll
      for (int ks = 10; ks < 20; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *(const uint4*)(xp + ks * 16);
        *(uint4*)(xhi + (ks - 10) * 1024) = v;
      }
    }
    if (a.ln_g) {
      // LayerNorm of the rows in place (the norm3 / norm_in in front of every feed-forward: one kernel launch and one
      // write + read of the activation less).  A row's 320 values sit in its two lanes (hsel = 0 / 1, 160 each: 80 in xr,
      // 80 in the wave's x_hi strip); statistics about the row's first value as pilot (shifted sums: no cancellation), one
      // lane exchange; the normalised values are rounded to bf16 like ctrlv_layernorm's output and overwrite the raw ones.
      const float* lv = a.lnv ? a.lnv + (long)((m / a.ln_vdiv) % a.ln_vmod) * a.ln_ldv + 8 * hsel : nullptr;
      auto raw8 = [&](int ks, float* f) {
        const uint4 v = ks < 10 ? __builtin_bit_cast(uint4, xr[ks]) : *(const uint4*)(xhi + (ks - 10) * 1024);
        unpack_elx8(v, f);
        if (lv) {
          const float4 p = *(const float4*)(lv + ks * 16), q = *(const float4*)(lv + ks * 16 + 4);
          f[0] += p.x; f[1] += p.y; f[2] += p.z; f[3] += p.w; f[4] += q.x; f[5] += q.y; f[6] += q.z; f[7] += q.w;
        }
      };
      float pilot;
      {
        float f0[8];
        raw8(0, f0);
        pilot = __shfl(f0[0], r32);                          // the row's column 0 (lane hsel = 0)
      }
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        float f[8];
        raw8(ks, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dl = f[e] - pilot; sm += dl; sq += dl * dl; }
        if (ks % 5 == 4) __builtin_amdgcn_sched_barrier(0);  // (five steps' loads in flight at a time, not all twenty)
      }
      sm += __shfl_xor(sm, 32);
      sq += __shfl_xor(sq, 32);
      const float dm = sm * (1.0f / kC);                     // mean - pilot
      const float var = sq * (1.0f / kC) - dm * dm;
      const float mean = pilot + dm, rstd = rsqrtf((var > 0.f ? var : 0.f) + a.ln_eps);
      const float* gp = a.ln_g + 8 * hsel;
      const float* bp = a.ln_b + 8 * hsel;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        float f[8];
        raw8(ks, f);
        const float4 g0 = *(const float4*)(gp + ks * 16), g1 = *(const float4*)(gp + ks * 16 + 4);
        const float4 b0 = *(const float4*)(bp + ks * 16), b1v = *(const float4*)(bp + ks * 16 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * gg[e] + bb[e];
        uint4 o = pack_elx8(f);
        if (!(m < M)) o = make_uint4(0, 0, 0, 0);
        if (ks < 10) xr[ks] = __builtin_bit_cast(elx8, o);
        else *(uint4*)(xhi + (ks - 10) * 1024) = o;
        if (ks % 5 == 4) __builtin_amdgcn_sched_barrier(0);
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
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) v = *(const float4*)(d.bias + n * 32 + 8 * q + 4 * hsel);
        if (vrow) {
          const float4 w = *(const float4*)(vrow + n * 32 + 8 * q + 4 * hsel);
          v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        acc[0][n][4 * q] = v.x; acc[0][n][4 * q + 1] = v.y; acc[0][n][4 * q + 2] = v.z; acc[0][n][4 * q + 3] = v.w;
      }
    // One instruction stream for both groups -- GEMM 1, GEGLU, GEMM 2 per chunk -- and ONE window boundary per chunk (wait
    // for the DMA issued a window ago, barrier, issue the DMA two chunks ahead), which group 0 takes after GEMM 2 and
    // group 1 between GEGLU and GEMM 2.  Ring safety: after boundary k the W1 slot of chunk k + 2 was last read by GEMM 1
    // of chunk k (both groups: before their boundary k), its W2 slot by group 1's GEMM 2 of chunk k - 1 (between its
    // boundaries k - 1 and k).  x_hi, the staging and the accumulators are wave-private: a new tile needs no barrier.
#ifdef CTRLV_FF_STAMP      // diagnostic build (tools/ff_stamp.py): cycles per phase, summed per wave, written to a.lnv
#define FSTAMP(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
    unsigned long long st_g1 = 0, st_ge = 0, st_b = 0, st_g2 = 0;
#else
#define FSTAMP(v)
#endif
#ifdef CTRLV_FF_STAMP
    unsigned long long st_w = 0, st_bar = 0, st_dma = 0;
#endif
    auto boundary = [&](int c) {
      FSTAMP(b0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FSTAMP(b1);
      __syncthreads();
      FSTAMP(b2);
      dma((c + 2) % kChunks, cglob + 2);
#ifdef CTRLV_FF_STAMP
      FSTAMP(b3);
      st_w += b1 - b0; st_bar += b2 - b1; st_dma += b3 - b2;
#endif
    };
    for (int c = 0; c < kChunks; ++c, ++cglob) {
      FSTAMP(t0);
      const char* s1 = smem + kW1Off + (cglob & 1) * kW1Slot + lane * 16;
      f32x16 a1;
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 20; ++ks) {
        const elx8 wf = *(const elx8*)(s1 + ks * 1024);
        const elx8 xf = ks < 10 ? xr[ks] : *(const elx8*)(xhi + (ks - 10) * 1024);
        a1 = mfma_32x32x16(wf, xf, a1);
      }
      a1 = mfma_32x32x16(*(const elx8*)(s1 + 20 * 1024), xone, a1);   // + bias
      FSTAMP(t1);
      // GEGLU in the result layout: accumulators 0..7 are the 8 value columns of this lane, 8..15 their gates
      float h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = geglu_tab(a1[e], a1[8 + e], tab);
      const uint4 hp = make_uint4(pack_elx2(h[0], h[1]), pack_elx2(h[2], h[3]), pack_elx2(h[4], h[5]),
                                  pack_elx2(h[6], h[7]));
      const elx8 hf = __builtin_bit_cast(elx8, hp);
      FSTAMP(t2);
      if (grp == 1) boundary(c);
      FSTAMP(t3);
      const char* s2 = smem + kW2Off + (cglob % 3) * kW2Slot + lane * 16;
#pragma unroll
      for (int n = 0; n < 10; ++n) {
        const elx8 wf = *(const elx8*)(s2 + n * 1024);
        acc[0][n] = mfma_32x32x16(wf, hf, acc[0][n]);
      }
      FSTAMP(t4);
      if (grp == 0) boundary(c);
#ifdef CTRLV_FF_STAMP
      FSTAMP(t5);
      st_g1 += t1 - t0; st_ge += t2 - t1; st_b += (t3 - t2) + (t5 - t4); st_g2 += t4 - t3;
#endif
    }
#ifdef CTRLV_FF_STAMP
    if (lane == 0 && a.lnv) {
      unsigned long long* o = (unsigned long long*)a.lnv + ((long)blockIdx.x * 8 + wid) * 4;
      o[0] += st_g1; o[1] += st_w; o[2] += st_bar; o[3] += st_dma;      // (variant: the boundary split up)
      (void)st_ge; (void)st_b; (void)st_g2;
    }
#endif
    // (x fragments are dead here: end their live ranges so that the epilogue's prefetch window gets their registers)
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) asm volatile("" : "=v"(xr[ks]));
    // ---- epilogue: the ping-pong GEMM's, on this wave's 32 x 320 block; staging = four KiB of the wave's x_hi strip
    char* stg = smem + wid * 10 * 1024;
    int lane_e = lane;                                       // (opaque copy: keeps the epilogue's lane constants per-tile values
    asm volatile("" : "+v"(lane_e));                         //  instead of hoisted, spilled ones -- gemm_pp_kernel.h)
    gemm_epilogue_lds<1, 10, false, EPI, false, false, LO>(d, acc, bm, 0, wid, 0, 32, kC, lane_e, stg, stg + 1024, stg + 2048,
                                                           stg + 3072, nullptr, tab);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // staging reads done before the next tile's x_hi writes
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead DMA of the last chunks: nothing may be in flight
  CTRLV_CLOCK_END();
#endif
}

// fragment-major copies of the two packed weights (device-side permutation of bf16 values, once per weight); the 21st
// K step of every W1 chunk carries the GEMM-1 bias as (bf16(b), bf16(b - bf16(b))) in K slots 0 and 1
__global__ void ff_pack_kernel(const el_t* __restrict__ w1p, const float* __restrict__ b1, const el_t* __restrict__ w2p,
                               el_t* __restrict__ w1f, el_t* __restrict__ w2f) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n1 = (long)kChunks * kW1Pieces * 64 * 8, n2 = (long)kChunks * 10 * 64 * 8;
  if (i < n1) {
    // w1f[chunk][ks][lane][j] = w1p[chunk*32 + lane%32][ks*16 + 8*(lane/32) + j]
    const int j = i & 7, lane = (i >> 3) & 63;
    const int ks = (int)((i >> 9) % kW1Pieces), chunk = (int)(i / (kW1Pieces * 512));
    el_t v = 0;
    if (ks < 20) {
      v = w1p[(long)(chunk * 32 + (lane & 31)) * kC + ks * 16 + 8 * (lane >> 5) + j];
    } else if (lane < 32 && j < 2) {
      const float b = b1[chunk * 32 + lane];
      const el_t hi = f32_to_el(b);
      v = j == 0 ? hi : f32_to_el(b - el_to_f32(hi));
    }
    w1f[i] = v;
  } else if (i < n1 + n2) {
    // w2f[chunk][n][lane][j] = w2p[n*32 + lane%32][chunk*16 + 8*(j/4) + 4*(lane/32) + j%4]   (k-slot 8h + j <-> column)
    const long k = i - n1;
    const int j = k & 7, lane = (k >> 3) & 63;
    const int n = (int)((k >> 9) % 10), chunk = (int)(k / (10 * 512));
    w2f[k] = w2p[(long)(n * 32 + (lane & 31)) * kHid + chunk * 16 + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)];
  }
}

template <int EPI, bool LO = false>
int launch_ff(const FfArgs& a, hipStream_t stream) {
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  auto kfn = ff_fused_kernel<EPI, LO>;
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

CTRLV_CLOCK_READER(ff_fused)

extern "C" int ctrlv_ff_fused_w1f_bytes(void) { return kChunks * kW1Slot; }

extern "C" int ctrlv_ff_fused_pack(const void* w1_packed, const float* b1, const void* w2_packed, void* w1f, void* w2f,
                                   ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(w1_packed && b1 && w2_packed && w1f && w2f, "ctrlv_ff_fused_pack: null pointer");
  const long n = (long)kChunks * (kW1Pieces + 10) * 64 * 8;
  hipLaunchKernelGGL(ff_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const el_t*)w1_packed, b1, (const el_t*)w2_packed, (el_t*)w1f, (el_t*)w2f);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

// Every condition under which the fused kernel serves a second-projection descriptor -- ONE function for the launcher
// and for the callers' switch (ctrlv_ff_fused_serves), so that a layer the switch accepts is never refused at launch.
// `fold` (out): the row vector rides in the accumulator start.  report = false: no error text (a "no" is not an error).
static int ff_check(const ctrlv_gemm_desc& d, int ldx, bool report, bool* fold) {
#define FF_REQ(cond, code, ...)                        \
  do {                                                 \
    if (!(cond)) {                                     \
      if (report) ctrlv_set_error(__VA_ARGS__);        \
      return code;                                     \
    }                                                  \
  } while (0)
  *fold = false;
  FF_REQ(d.out != nullptr, CTRLV_E_BAD_ARG, "ctrlv_ff_fused: null pointer");
  FF_REQ(d.M > 0 && d.N == kC && d.Cin == kHid && d.taps == 1 && d.mode == 0 && ldx >= kC && ldx % 8 == 0, CTRLV_E_BAD_SHAPE,
         "ctrlv_ff_fused: serves M x 320 <- 1280 <- 320 only (N=%d Cin=%d ldx=%d)", d.N, d.Cin, ldx);
  FF_REQ(!d.geglu && !d.act && !d.out_f32 && !d.raw_out && !d.A2 && d.n_scale2 == 0, CTRLV_E_BAD_ARG,
         "ctrlv_ff_fused: plain element-type output only");
  FF_REQ(d.n_store == kC && d.ldo % 8 == 0 && (!d.R1 || d.ldr1 % 8 == 0) && (!d.R2 || d.ldr2 % 8 == 0), CTRLV_E_BAD_SHAPE,
         "ctrlv_ff_fused: n_store must be 320 and the row pitches multiples of 8");
  const long lim = 0xFFFFFFF0L;
  FF_REQ((long)d.M * d.ldo * 2 <= lim && (long)d.M * ldx * 2 <= lim && (!d.R1 || (long)d.M * d.ldr1 * 2 <= lim) &&
             (!d.R2 || (long)d.M * d.ldr2 * 2 <= lim),
         CTRLV_E_BAD_SHAPE, "ctrlv_ff_fused: operands beyond 32-bit byte offsets");
  FF_REQ(!d.R2 || d.R1, CTRLV_E_BAD_ARG, "ctrlv_ff_fused: R2 without R1");
  if (pp_split_io(d)) {      // split trunk planes (include/ctrlv_hip.h): the fp16 element library, {R1} and {R1, R2} epilogues
    FF_REQ(CTRLV_ELEM_DTYPE == 1 && d.R1 && (!d.R1_lo || d.R1) && (!d.R2_lo || d.R2), CTRLV_E_BAD_ARG,
           "ctrlv_ff_fused: split trunk planes need the fp16 element library and an R1 operand");
  }
  if (d.vmode) {
    FF_REQ((d.vmode == 1 || d.vmode == 2) && d.V && d.vdiv > 0 && d.vmod > 0 && d.ldv >= kC && d.ldv % 4 == 0 &&
               (d.vmode == 1 || d.vS > 0),
           CTRLV_E_BAD_ARG, "ctrlv_ff_fused: bad row-vector operand");
    // one vector per 256-row tile, unscaled (the frame embedding of ff_in): it rides in the accumulator start (kernel)
    // and the epilogue has no row-vector reads; anything else is the shared epilogue's row-vector operand (EPI bit 0)
    *fold = d.vmode == 1 && d.vdiv % 256 == 0 && d.s_acc == 1.0f;
    FF_REQ(*fold || !d.R2, CTRLV_E_BAD_ARG,
           "ctrlv_ff_fused: R1 + R2 + a row vector is served only in the per-tile form (vmode 1, vdiv %% 256 == 0, s_acc 1)");
    FF_REQ(*fold || !pp_split_io(d), CTRLV_E_BAD_ARG, "ctrlv_ff_fused: split trunk planes with a row vector: per-tile form only");
  }
  return CTRLV_OK;
#undef FF_REQ
}

extern "C" int ctrlv_ff_fused_ln(const void* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                 const float* ln_V, int ln_vdiv, int ln_vmod, int ln_ldv, const void* w1f, const void* w2f,
                                 const ctrlv_gemm_desc* out_desc, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && w1f && w2f && out_desc && out_desc->out, "ctrlv_ff_fused: null pointer");
  CTRLV_CHECK_ARG((ln_gamma == nullptr) == (ln_beta == nullptr), "ctrlv_ff_fused: LayerNorm needs gamma and beta");
#ifndef CTRLV_FF_STAMP        // (the stamped diagnostic build receives its output buffer through ln_V)
  CTRLV_CHECK_ARG(!ln_V || (ln_gamma && ln_vdiv > 0 && ln_vmod > 0 && ln_ldv >= 320 && ln_ldv % 4 == 0),
                  "ctrlv_ff_fused: bad LayerNorm row-vector table");
#endif
  FfArgs a;
  a.ln_g = ln_gamma; a.ln_b = ln_beta; a.ln_eps = ln_eps;
  a.lnv = ln_V; a.ln_vdiv = ln_V ? ln_vdiv : 1; a.ln_vmod = ln_V ? ln_vmod : 1; a.ln_ldv = ln_ldv;
  a.x = (const el_t*)x; a.ldx = ldx; a.w1f = (const el_t*)w1f; a.w2f = (const el_t*)w2f;
  a.o = *out_desc;
  a.vtab = nullptr; a.vdiv = 1; a.vmod = 1; a.ldv = 0;
  bool fold = false;
  const int rc = ff_check(a.o, ldx, true, &fold);
  if (rc != CTRLV_OK) return rc;
  if (fold) {
    a.vtab = a.o.V; a.vdiv = a.o.vdiv; a.vmod = a.o.vmod; a.ldv = a.o.ldv;
    a.o.vmode = 0; a.o.V = nullptr;
  }
  const ctrlv_gemm_desc& d = a.o;
  hipStream_t st = (hipStream_t)stream;
  // (EPI = 1 / 3: the row-vector operand through the epilogue.  The R1 + V instantiation at this wave shape -- TM = 1,
  // TN = 10 -- is the one that "intermittently stored zero dwords" in round 3: the store-data hazard of gemm_pp_kernel.h
  // (store_data_hazard_guard), a zero-initialisation of the next sub-tile's row-vector registers scheduled right behind a
  // buffer store whose data registers it reused.  Guarded, it is back in the build; tests: 48-run bit-stability.)
#ifdef CTRLV_ELEM_F16
  if (pp_split_io(d)) {
    if (pp_epi_of(d) == 2) return launch_ff<2, true>(a, st);
    if (pp_epi_of(d) == 6) return launch_ff<6, true>(a, st);
  }
#endif
  switch (pp_split_io(d) ? -1 : pp_epi_of(d)) {
    case 0: return launch_ff<0>(a, st);
    case 1: return launch_ff<1>(a, st);
    case 2: return launch_ff<2>(a, st);
    case 3: return launch_ff<3>(a, st);
    case 6: return launch_ff<6>(a, st);
    default: break;
  }
  ctrlv_set_error("ctrlv_ff_fused: epilogue operand combination not served (bias [+ V], + R1 [+ V], + R1 + R2 [+ a per-tile V])");
  return CTRLV_E_BAD_ARG;
}

extern "C" int ctrlv_ff_fused(const void* x, int ldx, const void* w1f, const void* w2f,
                              const ctrlv_gemm_desc* out_desc, ctrlv_stream_t stream) {
  return ctrlv_ff_fused_ln(x, ldx, nullptr, nullptr, 0.f, nullptr, 1, 1, 0, w1f, w2f, out_desc, stream);
}

// 1 if ctrlv_ff_fused serves this second-projection descriptor with input rows of pitch ldx (the callers' switch between
// the fused kernel and the two ctrlv_gemm launches): exactly the launcher's own conditions
extern "C" int ctrlv_ff_fused_serves(const ctrlv_gemm_desc* out_desc, int ldx) {
  if (!out_desc) return 0;
  bool fold = false;
  if (ff_check(*out_desc, ldx, false, &fold) != CTRLV_OK) return 0;
  return (out_desc->act || (out_desc->out_f32 & 1)) ? 0 : 1;
}
