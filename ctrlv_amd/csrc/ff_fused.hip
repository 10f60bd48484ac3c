// Fused feed-forward pair at C = 320 (the 72 x 128 level's  x -> GEGLU(x W1^T + b1) W2^T + b2 (+ residuals)  of
// BasicTransformerBlock.ff / TemporalBasicTransformerBlock.ff_in / .ff), gfx950.
//
// The two GEMM launches it replaces (ctrlv_gemm with geglu = 1, then the 1280 -> 320 projection) write the 4C-wide
// intermediate u to HBM and read it back: 2.4 GB per pair at M = 460 800.  Here u never leaves the CU.
//
// Round 5 structure ("pair" kernel; rounds 3-4 ran 8 waves x (32 rows x 320 output columns), whose 160 accumulator + 40
// x registers left the compiler no room: fragment reads two at a time right in front of their MFMAs, the LDS latency of
// every pair exposed, matrix pipe 47 % busy):
//   * 128-row tile per workgroup, 8 waves = 4 row groups x 2 halves.  The two waves of a row group (w, w + 4: the two
//     waves of one SIMD) hold the SAME 32 x rows, all 20 k-steps in registers (80; no x strip in LDS), and each keeps
//     32 x 160 OUTPUT accumulators (80 registers) -- half the output columns.
//   * the hidden dimension is walked in chunks of 16 columns; GEMM 1 + GEGLU of a chunk (21 MFMAs: K = 320 plus the bias
//     step, then the Phi table) is done ONCE per row group -- even chunks by half 0, odd chunks by half 1 -- and the 32 x 16
//     result h (bf16, already in B-operand form: 1 KiB) is handed to the partner through LDS; GEMM 2 of a chunk PAIR
//     (2 x 5 MFMAs per wave: its 160 columns) is done by both.  Same MFMAs, same K order, same chunk order of the GEMM 2
//     sum as the round 3-4 kernel: bit-identical results.
//   * time is cut into SLOTS separated by ONE workgroup barrier.  In slot t the half (t & 1) has its A slot of chunk t
//     (the 21-MFMA chain, then six of the lane's eight GEGLU columns); the other half its B slot: the last two GEGLU
//     columns of the chunk of ITS A slot before (h -> registers + LDS), the slot's LDS-DMA issue, 10 MFMAs of the chunk
//     pair whose two h are visible.  On every SIMD one wave is in its chain while its partner finishes a GEGLU, issues
//     DMA and runs its 10 MFMAs: 31 MFMAs per SIMD and slot, the two roles about equally long.
//   * W1 chunks (21 KiB, fragment-major) through a 3-deep LDS-DMA ring, issued two slots ahead; W2 chunk pairs (2 x 10 KiB)
//     through a 4-chunk ring, issued at the even slot two before their first use.  Fragment reads run SIX MFMAs ahead
//     through rotating registers, every step fenced (sched_barrier) so that the compiler keeps that distance.
//   * epilogue = the ping-pong GEMM's (gemm_epilogue_lds: s_acc * acc + s1 R1 + s2 R2 + V, LDS transpose, 16-B stores) on
//     the wave's 32 x 160 block.
// Arithmetic: the same MFMA, the same K order inside GEMM 1, the same bias step, the same GELU table and the same bf16
// rounding of u as the two-launch path; GEMM 2 sums its K = 1280 in chunk order with the permuted slot assignment, so
// its fp32 sums differ from ctrlv_gemm's in the last bits.  Every C = 320 feed-forward of the inference path goes through
// this kernel whatever M is (the training forward keeps the two launches: it needs u and the raw projection).
#include <type_traits>

#include "common.h"
#include "gemm_pp_kernel.h"

namespace {

constexpr int kC = 320, kHid = 1280, kChunks = kHid / 16;        // 80 chunks of 16 hidden columns
constexpr int kTileM = 128;
#ifndef CTRLV_FF_GEA
#define CTRLV_FF_GEA 6
#endif
constexpr int kGeA = CTRLV_FF_GEA;                               // GEGLU columns done in the A slot (the rest in the wave's next B slot)
constexpr int kNQ = 6;                                           // W1 fragment reads in flight ahead of the MFMA chain
constexpr int kW1Pieces = 21;                                    // 20 k-steps of K = 320 + one carrying the bias (see below)
constexpr int kW1Slot = kW1Pieces * 1024, kW2Slot = 10 * 1024;
constexpr int kStgOff = 0;                                       // epilogue staging: 8 waves x 4 KiB
constexpr int kHfOff = kStgOff + 8 * 4096;                       // h exchange: [row group][chunk & 3] x 1 KiB
constexpr int kW1Off = kHfOff + 16 * 1024;                       // W1 ring: 3 chunks
constexpr int kW2Off = kW1Off + 3 * kW1Slot;                     // W2 ring: 4 chunks (two pairs)
constexpr int kTabOff = kW2Off + 4 * kW2Slot;
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

template <int EPI, bool LO = false, bool LN = false>
__global__ __launch_bounds__(512) void ff_fused_kernel(const FfArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rg = wid & 3, hh = wid >> 2;                      // row group, half (waves w and w + 4 share a SIMD)
  const int r32 = lane & 31, hsel = lane >> 5;
  const ctrlv_gemm_desc& d = a.o;
  const int M = d.M;
  const int tiles = (M + kTileM - 1) / kTileM, G = gridDim.x;

  CTRLV_CLOCK_BEGIN();
  gelu_table_fill(smem + kTabOff, threadIdx.x, 512);

  const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1f, 0, kChunks * kW1Slot, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2f, 0, kChunks * kW2Slot, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((long)d.M * a.ldx * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB2 = __builtin_amdgcn_make_buffer_rsrc((void*)(d.bias ? d.bias : (const float*)a.w1f), 0,
                                                                         d.bias ? kC * 4 : 0, 0x00020000);
  // LDS-DMA of one W1 chunk (21 KiB pieces) / one W2 chunk (10): the four waves of the half that is in its B segment
  // take piece k * 4 + rg.  `g` = the workgroup's running chunk count (ring phase), `chunk` = which of the 80.
  auto dma_w1 = [&](int chunk, int g) {
    char* s1 = smem + kW1Off + (g % 3) * kW1Slot;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int pi = k * 4 + rg;
      if (pi < kW1Pieces)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW1, LDS_PTR(s1 + pi * 1024), 16, lane * 16, chunk * kW1Slot + pi * 1024, 0, 0);
    }
  };
  auto dma_w2 = [&](int chunk, int g) {
    char* s2 = smem + kW2Off + (g & 3) * kW2Slot;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int pi = k * 4 + rg;
      if (pi < 10)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW2, LDS_PTR(s2 + pi * 1024), 16, lane * 16, chunk * kW2Slot + pi * 1024, 0, 0);
    }
  };
  // the W2 ring starts as zeros: the empty pair of the first tile multiplies it by zero h fragments (0 x NaN bits would not be 0)
  for (int i = threadIdx.x * 16; i < 4 * kW2Slot; i += 512 * 16) *(uint4*)(smem + kW2Off + i) = make_uint4(0, 0, 0, 0);
  // kernel start: W1 chunks 0 and 1 of the first tile (later tiles find theirs prefetched by the tile before)
  if (hh == 0) dma_w1(0, 0); else dma_w1(1, 1);
  wait_vmcnt<0>();
  __syncthreads();                                           // chunks 0 / 1 and the table visible to every wave
  const char* tab = smem + kTabOff;
  const unsigned tab_lds = (unsigned)(unsigned long)LDS_PTR(smem + kTabOff);
  char* const hfx = smem + kHfOff + rg * 4096 + lane * 16;   // this row group's four exchange buffers
  // GEMM-1 bias: not a strip read as the C operand but a 21st K step -- the packed W1 chunk carries (bf16(b), bf16(b -
  // bf16(b))) in two K slots against a constant-one x fragment, which adds b to within 2^-17 |b| in the fp32 accumulator
  const unsigned xone1 = hsel == 0 ? (CTRLV_ELEM_DTYPE == 1 ? 0x3C003C00u : 0x3F803F80u) : 0u;   // (1, 1) in K slots 0, 1

  int gbase = 0;                                             // chunks started so far by this workgroup (ring phase)
  // The x rows (20 loads per lane) and the accumulator start (bias + the tile's row vector: one float4 for 40 lanes of a
  // wave) of the NEXT tile are requested behind a tile's last A segment and arrive under its closing slots and epilogue: a
  // tile used to open with 15-20 thousand cycles of exposed memory latency (16 % of its time with the epilogue).
  elx8 xr[20];
  u32x4_t ini_b, ini_v;
  const __amdgpu_buffer_rsrc_t rsVt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.vtab ? a.vtab : (const float*)a.w1f), 0, a.vtab ? (int)((long)a.vmod * a.ldv * 4) : 0, 0x00020000);
  auto tile_loads = [&](int tile) {
    // (through buffer descriptors: rows >= M lie behind the end and read as zeros, an absent bias / row vector is an empty
    //  descriptor -- no branch; 22 vector-memory operations: the closing slot barrier counts on that)
    const int bm = tile * kTileM;
    const unsigned xoff = (unsigned)(bm + rg * 32 + r32) * (unsigned)(a.ldx * 2) + 16 * hsel;
#pragma unroll
    for (int ks = 0; ks < 20; ++ks)
      xr[ks] = __builtin_bit_cast(elx8, __builtin_amdgcn_raw_buffer_load_b128(rsX, xoff, ks * 32, 0));
    const unsigned col4 = lane < 40 ? (unsigned)(hh * 160 + lane * 4) * 4 : 0xFFFFFFFFu;
    ini_b = __builtin_amdgcn_raw_buffer_load_b128(rsB2, col4, 0, 0);
    ini_v = __builtin_amdgcn_raw_buffer_load_b128(rsVt, col4, a.vtab ? ((bm / a.vdiv) % a.vmod) * (a.ldv * 4) : 0, 0);
  };
  tile_loads(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles; tile += G, gbase += kChunks) {
    const int bm = tile * kTileM;
    const int m = bm + rg * 32 + r32;
#ifdef CTRLV_FF_STAMP
    unsigned long long st_tb; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_tb)::"memory");
#endif
    // ---- x rows of this row group as B fragments (both halves hold them): k-step ks = columns ks*16 + 8*hsel .. +8 of row m
    if constexpr (LN) {
      // LayerNorm of the rows in place (the norm3 / norm_in in front of every feed-forward: one kernel launch and one
      // write + read of the activation less).  A row's 320 values sit in its two lanes (hsel = 0 / 1, 160 each);
      // statistics about the row's first value as pilot (shifted sums: no cancellation), one lane exchange; the
      // normalised values are rounded to the element type like ctrlv_layernorm's output and overwrite the raw ones.
      const float* lv = a.lnv ? a.lnv + (long)((m / a.ln_vdiv) % a.ln_vmod) * a.ln_ldv + 8 * hsel : nullptr;
      auto raw8 = [&](int ks, float* f) {
        unpack_elx8(__builtin_bit_cast(uint4, xr[ks]), f);
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
        xr[ks] = __builtin_bit_cast(elx8, o);
        if (ks % 5 == 4) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // output accumulators (this half's 160 columns) start from b2 -- plus the tile's row vector: V is constant over a tile
    // (vdiv is a multiple of 256, checked by the host; the frame positional embedding of ff_in: one vector per frame of S
    // pixels) and s_acc is 1 there, so it rides in the accumulator instead of the epilogue
    f32x16 acc[1][5];
    char* const stg = smem + kStgOff + wid * 4096;          // this wave's epilogue staging; here: the strip of its 160 start values
    {
      const f32x4 b = __builtin_bit_cast(f32x4, ini_b), w = __builtin_bit_cast(f32x4, ini_v);
      if (lane < 40) *(float4*)(stg + lane * 16) = make_float4(b.x + w.x, b.y + w.y, b.z + w.z, b.w + w.w);
      __builtin_amdgcn_wave_barrier();                       // (compiler-only: the strip is exchanged between lanes of this wave)
#pragma unroll
      for (int i = 0; i < 20; ++i) {
        const int n = i >> 2, q = i & 3;
        const float4 v = *(const float4*)(stg + (n * 32 + 8 * q + 4 * hsel) * 4);
        acc[0][n][4 * q] = v.x; acc[0][n][4 * q + 1] = v.y; acc[0][n][4 * q + 2] = v.z; acc[0][n][4 * q + 3] = v.w;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the strip is read before the epilogue reuses the staging)
    }
#ifdef CTRLV_FF_STAMP      // diagnostic build (tools/ff_stamp.py): cycles per phase, summed per wave, written to a.lnv
#define FSTAMP(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
    unsigned long long st_g1 = 0, st_ge = 0, st_b = 0, st_bar = 0, st_dma = 0, st_tile0;
    { FSTAMP(tt0); st_tile0 = tt0; }
#else
#define FSTAMP(v)
#endif
    // ---- segment A of chunk t (this half's turn): GEMM 1 (21 chained MFMAs), GEGLU, h -> own registers + exchange buffer
    elx8 hcur, hprev;                                        // this half's h of its last two chunks
    {
      const uint4 z = make_uint4(0, 0, 0, 0);
      hcur = hprev = __builtin_bit_cast(elx8, z);
    }
    // GEGLU of elements E0..E1-1 of the chain's result, in the result layout: accumulators 0..7 are the 8 value columns of
    // this lane, 8..15 their gates.  (Table reads as asm statements with their own wait: the compiler puts a vmcnt(0) in
    // front of an LDS read it knows of when LDS-DMA of this wave is in flight, and the wave stood there.)
    auto geglu_part = [&](auto e0c, auto e1c, const float* av, const float* gv, float* h) {
      constexpr int E0 = decltype(e0c)::value, E1 = decltype(e1c)::value, NE = E1 - E0;
      float fr[NE ? NE : 1];
      f32x2_t te[NE ? NE : 1];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
#pragma clang fp contract(off)
        float tq = __builtin_fmaf(gv[E0 + e], 100.0f, 512.0f);
        tq = __builtin_amdgcn_fmed3f(tq, 0.0f, 1023.99994f);
        fr[e] = __builtin_amdgcn_fractf(tq);
        const unsigned addr = tab_lds + (unsigned)((int)tq) * 8u;
        asm volatile("ds_read_b64 %0, %1" : "=v"(te[e]) : "v"(addr));
      }
      static_assert(NE == 0 || NE == 2 || NE == 4 || NE == 6 || NE == 8, "columns per part");
      if constexpr (NE == 8)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(te[0]), "+v"(te[1]), "+v"(te[2]), "+v"(te[3]), "+v"(te[4]), "+v"(te[5]), "+v"(te[6]), "+v"(te[7]));
      if constexpr (NE == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(te[0]), "+v"(te[1]));
      if constexpr (NE == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(te[0]), "+v"(te[1]), "+v"(te[2]), "+v"(te[3]));
      if constexpr (NE == 6)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(te[0]), "+v"(te[1]), "+v"(te[2]), "+v"(te[3]), "+v"(te[4]), "+v"(te[5]));
#pragma unroll
      for (int e = 0; e < NE; ++e) h[E0 + e] = geglu_tab_finish(av[E0 + e], gv[E0 + e], fr[e], make_float2(te[e].x, te[e].y));
    };
    // ---- A slot of chunk t (this half's turn): GEMM 1 (21 chained MFMAs), then kGeA = 6 of the lane's eight GEGLU columns.
    // The last two wait for the wave's NEXT slot (its B role): the slot's two roles then take about equally long -- chain +
    // 6 columns against 2 columns + DMA issue + 10 MFMAs -- behind ONE barrier per slot.  Measured at M = 460 800 on one
    // device (tools/ab_build.py, -DCTRLV_FF_GEA): 8 + 0 columns 1.139 ms, 6 + 2 1.122, 4 + 4 1.138, 2 + 6 1.152; with a
    // second barrier per slot between the chain and a whole GEGLU (the partner's MFMAs beside the GEGLU, not inside the
    // chain) 1.17 ms.  (The first measurements of the one-barrier forms read 1.25-1.43 ms: the h store sat BEHIND the DMA
    // issue and the compiler puts a vmcnt(0) in front of an LDS store while LDS-DMA of the wave is in flight.)
    uint32_t hpk[kGeA / 2];                                         // packed h columns 0..3 of the chunk of this wave's last A slot
    float ak[kGeA < 8 ? 8 - kGeA : 1], gk[kGeA < 8 ? 8 - kGeA : 1];                                      // its raw columns 4..7 (value, gate)
    auto seg_g1 = [&](int t) {
      FSTAMP(t0);
      // (the chain is the slot's critical path: its wave outranks the partner)
      __builtin_amdgcn_s_setprio(2);
      const char* s1 = smem + kW1Off + ((gbase + t) % 3) * kW1Slot + lane * 16;
      elx8 wq[kNQ];
#pragma unroll
      for (int i = 0; i < kNQ; ++i) wq[i] = *(const elx8*)(s1 + i * 1024);
      f32x16 a1;
#pragma unroll
      for (int e = 0; e < 16; ++e) a1[e] = 0.f;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 21; ++ks) {                      // k-step 20: the bias against the constant-one fragment
        elx8 xf;
        if (ks == 20) {                                      // (built here: four registers for one step, not for the loop)
          unsigned one = xone1;
          asm volatile("" : "+v"(one));
          xf = __builtin_bit_cast(elx8, make_uint4(one, 0, 0, 0));
        } else {
          xf = xr[ks];
        }
        a1 = mfma_32x32x16(wq[ks % kNQ], xf, a1);
        if (ks + kNQ < 21) wq[ks % kNQ] = *(const elx8*)(s1 + (ks + kNQ) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
      FSTAMP(t1);
      float av[8], gv[8], h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { av[e] = a1[e]; gv[e] = a1[8 + e]; }
      geglu_part(std::integral_constant<int, 0>{}, std::integral_constant<int, kGeA>{}, av, gv, h);
#pragma unroll
      for (int e = 0; e < kGeA / 2; ++e) hpk[e] = pack_elx2(h[2 * e], h[2 * e + 1]);
#pragma unroll
      for (int e = 0; e < 8 - kGeA; ++e) { ak[e] = av[kGeA + e]; gk[e] = gv[kGeA + e]; }
      __builtin_amdgcn_s_setprio(0);
#ifdef CTRLV_FF_STAMP
      FSTAMP(t2);
      st_g1 += t1 - t0; st_ge += t2 - t1;
#endif
    };
    // ---- the last columns of chunk t (B slot): h -> own registers + the exchange buffer the partner reads a slot on
    auto seg_ge = [&](int t) {
      FSTAMP(t1b);
      float av[8], gv[8], h[8];
#pragma unroll
      for (int e = 0; e < 8 - kGeA; ++e) { av[kGeA + e] = ak[e]; gv[kGeA + e] = gk[e]; }
      geglu_part(std::integral_constant<int, kGeA>{}, std::integral_constant<int, 8>{}, av, gv, h);
      uint32_t hw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hw[e] = e < kGeA / 2 ? hpk[e] : pack_elx2(h[2 * e], h[2 * e + 1]);
      const uint4 hp = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *(uint4*)(hfx + (t & 3) * 1024) = hp;
      hprev = hcur;
      hcur = __builtin_bit_cast(elx8, hp);
#ifdef CTRLV_FF_STAMP
      FSTAMP(t2);
      st_ge += t2 - t1b;
#endif
    };
    // the slot's LDS-DMA issue (the half in its B role, first thing: the W1 chunk read two slots on, in even slots the W2
    // pair read from two slots on; the ring buffers written here were last read in slot t - 1)
    auto seg_dma = [&](int t) {
      FSTAMP(t0);
      dma_w1((t + 2) % kChunks, gbase + t + 2);              // chunks 80, 81 = the next tile's 0, 1
      if ((t & 1) == 0) { dma_w2(t, gbase + t); dma_w2(t + 1, gbase + t + 1); }
#ifdef CTRLV_FF_STAMP
      FSTAMP(t1);
      st_dma += t1 - t0;
#endif
    };
    // ---- segment B: this half's 2 x 5 MFMAs of chunk pair j (chunks 2j, 2j + 1).  A pair outside 0..39 (the first B
    // segment of half 0 and the last one of half 1: see the slot plan) runs on zero h fragments -- 10 MFMAs that add
    // nothing, beside the partner's A segment, and in exchange the accumulators are updated in ONE straight line
    auto seg_b = [&](int j) {
      FSTAMP(t0);
      const int c0 = 2 * j;
      const bool live = j >= 0 && j < kChunks / 2;
      const elx8 hpart = *(const elx8*)(hfx + ((c0 + (hh ^ 1)) & 3) * 1024);   // the partner's chunk of the pair
      const char* s2a = smem + kW2Off + ((gbase + c0) & 3) * kW2Slot + (5 * hh) * 1024 + lane * 16;
      const char* s2b = smem + kW2Off + ((gbase + c0 + 1) & 3) * kW2Slot + (5 * hh) * 1024 + lane * 16;
      elx8 vq[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) vq[i] = *(const elx8*)((i < 5 ? s2a + i * 1024 : s2b + (i - 5) * 1024));
      // own h of the pair: half 0 computed chunk 2j two A segments ago (2j + 2 came in between), half 1 chunk 2j + 1 last
      const uint4 hz = make_uint4(0, 0, 0, 0);
      const uint4 hp_u = __builtin_bit_cast(uint4, hpart);
      const uint4 own_u = hh == 0 ? __builtin_bit_cast(uint4, hprev) : __builtin_bit_cast(uint4, hcur);
      const uint4 h0_u = !live ? hz : hh == 0 ? own_u : hp_u;
      const uint4 h1_u = !live ? hz : hh == 0 ? hp_u : own_u;
      const elx8 h0 = __builtin_bit_cast(elx8, h0_u), h1 = __builtin_bit_cast(elx8, h1_u);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        acc[0][i % 5] = mfma_32x32x16(vq[i % 6], i < 5 ? h0 : h1, acc[0][i % 5]);
        if (i + 6 < 10) vq[i % 6] = *(const elx8*)(s2b + (i + 6 - 5) * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef CTRLV_FF_STAMP
      FSTAMP(t1);
      st_b += t1 - t0;
#endif
    };
    auto slot_barrier = [&](bool dma_wave) {
      FSTAMP(b0);
      if (dma_wave) wait_vmcnt<0>();                         // (its DMA of the slot before last: landed before this barrier)
      lds_done_barrier();
#ifdef CTRLV_FF_STAMP
      FSTAMP(b1);
      st_bar += b1 - b0;
#endif
    };
    // Slot plan, t = 0..81, one barrier per slot: half (t & 1) has its A slot of chunk t (t < 80): the chain and six columns
    // of the GEGLU; the other half its B slot: the last two columns of the chunk of its A slot before, the slot's DMA issue,
    // 10 MFMAs of a chunk pair.
    //   half 0: chain(2k) in slot 2k,     h(2k), B(pair k - 1) in slot 2k + 1  (k = 0: the empty pair -1); B(39) in slot 81
    //   half 1: chain(2k + 1) in slot 2k + 1, h(2k + 1), B(pair k) in slot 2k + 2;                 the empty B(40) in slot 81
    // Both halves run ONE program, half 1 a slot behind half 0 (with the two roles as branches of one loop, or one loop per
    // half, the register allocator put the accumulators of the paths into different registers: 40 v_mov_b64 per slot and
    // spilled x fragments).  h of chunk c is written in slot c + 1 and read by the partner in slot c + 2.  A wave waits for
    // its own LDS-DMA of slot t (vmcnt) in front of the barrier of slot t + 2, its next B slot.
    if (hh == 1) { slot_barrier(true); seg_dma(0); }
    for (int k = 0; k < kChunks / 2; ++k) {
      slot_barrier(false);
      seg_g1(2 * k + hh);
      slot_barrier(true);
      seg_ge(2 * k + hh);                                    // (in front of the DMA issue: the compiler puts a vmcnt(0) before an
      if (2 * k + 1 + hh < kChunks) seg_dma(2 * k + 1 + hh); //  LDS store -- h -- while LDS-DMA of the wave is in flight)
      seg_b(k - 1 + hh);
    }
    tile_loads(tile + G);                                    // (the x registers are free: both halves are past their last chain)
    slot_barrier(false);
    if (hh == 0) {
      FSTAMP(b0);
      wait_vmcnt<22>();                                      // its DMA of slot 79 -- everything but the 22 loads just issued
      lds_done_barrier();
#ifdef CTRLV_FF_STAMP
      FSTAMP(b1);
      st_bar += b1 - b0;
#endif
    }
    hprev = hcur;                                            // (half 0: no GEGLU came after chunk 78's)
    seg_b(kChunks / 2 - 1 + hh);
#ifdef CTRLV_FF_STAMP
    if (lane == 0 && a.lnv) {
      FSTAMP(tt1);
      unsigned long long* o = (unsigned long long*)a.lnv + ((long)blockIdx.x * 8 + wid) * 8;
      o[0] += st_g1; o[1] += st_ge; o[2] += st_dma; o[3] += st_b; o[4] += st_bar; o[5] += tt1 - st_tile0;
      o[6] += st_tile0 - st_tb;
    }
    unsigned long long st_e0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_e0)::"memory");
#endif
    // ---- epilogue: the ping-pong GEMM's, on this wave's 32 x 160 block; staging = four KiB of this wave's own
    int lane_e = lane;                                       // (opaque copy: keeps the epilogue's lane constants per-tile values
    asm volatile("" : "+v"(lane_e));                         //  instead of hoisted, spilled ones -- gemm_pp_kernel.h)
    gemm_epilogue_lds<1, 5, false, EPI, false, false, LO>(d, acc, bm, 0, rg, hh, 32, 160, lane_e, stg, stg + 1024, stg + 2048,
                                                          stg + 3072, nullptr, tab);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // staging reads done before the next tile's writes
#ifdef CTRLV_FF_STAMP
    if (lane == 0 && a.lnv) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long st_e1; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_e1)::"memory");
      ((unsigned long long*)a.lnv + ((long)blockIdx.x * 8 + wid) * 8)[7] += st_e1 - st_e0;
    }
#endif
  }
  wait_vmcnt<0>();                                           // the look-ahead DMA of the last slots: nothing may be in flight
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

template <int EPI, bool LO, bool LN>
int launch_ff_ln(const FfArgs& a, hipStream_t stream) {
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  auto kfn = ff_fused_kernel<EPI, LO, LN>;
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
    attr_set[dev] = true;
  }
  const int num_cu = ctrlv_num_cu(dev);
  const int tiles = (a.o.M + kTileM - 1) / kTileM;
  int grid = tiles;
  if (tiles > num_cu) {                     // persistent, every workgroup the same number of tiles
    const int rounds = (tiles + num_cu - 1) / num_cu;
    grid = (tiles + rounds - 1) / rounds;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), kSmem, stream, a);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

// (the LayerNorm prologue is a kernel of its own: its registers and spills stay out of the plain kernel)
template <int EPI, bool LO = false>
int launch_ff(const FfArgs& a, hipStream_t stream) {
  return a.ln_g ? launch_ff_ln<EPI, LO, true>(a, stream) : launch_ff_ln<EPI, LO, false>(a, stream);
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
