// Fused temporal self-attention block at C = 320 (the 72 x 128 level's TemporalBasicTransformerBlock.attn1 with its residual
// and the one-key cross-attention vector; SURVEY A.4, reference instantiation sites src/ctrlv/models/controlnet.py:157-170), gfx950:
//
//     g1 = g0 + to_out( softmax_f( q k^T / 8 ) v ) + b_o + V[clip],      (q | k | v) = tt W_qkv^T,   tt = LayerNorm(g0)
//
// per PIXEL over its F <= 32 frames.  The three launches it replaces (q|k|v projection, ctrlv_attention_temporal, output
// projection) write the 3C-wide q|k|v tensor and the attention output to HBM and read them back: 2.4 GB of the 4.1 GB the
// block moves per instance at M = 460 800.  Here nothing but tt, g0 and g1 touches HBM.
//
// Structure: ONE WAVE = ONE PIXEL.  A pixel's F frames are the 32 columns of every MFMA (rows (b F + f) S + s of the
// channels-last activation: the (b f) s c <-> (b s) f c permutes are address arithmetic, as everywhere in this library); its
// 32 x 320 input rows sit in registers as B-operand fragments (80 registers, the pair kernel's form, ff_fused.hip) and
// everything between them and the output accumulators STAYS IN REGISTERS -- no activation goes through LDS:
//   * q_h, k_h (per 32-channel block): acc = W . x  (weight block = A operand)  -> lane = frame, 16 channels 8 q + 4 hsel + r.
//     Both come out in the SAME channel-to-slot assignment, and a contraction does not care which channel sits in which K
//     slot: rounded to the element type, k is the A operand and q the B operand of S^T = K Q^T as they stand (4 MFMAs / head).
//   * softmax over the keys of a query = over the lane's 16 scores and its half-wave partner's (one v_permlane32_swap each for
//     max and sum); keys >= F are masked.  The exponentials, rounded, are the B operand of O^T = V^T P as they stand.
//   * v_h is computed with the operand ROLES SWAPPED (x = A operand, weight block = B operand: the same registers and the same
//     LDS fragments, the two arguments of the MFMA exchanged): lane = channel, 16 frames 8 q + 4 hsel + r -- which IS the A
//     operand V^T of O^T = V^T P with the key-to-slot assignment P has.  No transpose anywhere.
//   * O^T (lane = frame, channels 8 q + 4 hsel + r) rounded is the B operand of the output projection; its channel-to-slot
//     assignment is baked into the K order of the packed W_o (ctrlv_temporal_fused_pack), and the ROW order of every W_o block
//     is chosen so that a lane ends up with 16 CONSECUTIVE output channels: residual reads and stores are 2 x 16 B per lane
//     and block straight from / to the accumulator registers (no LDS transpose).
// Weights: 40 chunks of 20 KiB (32 rows x K = 320, fragment-major: one KiB per MFMA, lane-linear) stream in PAIRS through a
// 3-slot LDS-DMA ring (120 KiB), issued two pairs ahead; the 8 waves of a workgroup (8 consecutive pixels) walk the pairs in
// two groups half a slot apart (ping-pong: one group's 40 MFMAs beside the other's roundings / softmax / stores / loads).  Order: per head (q b0, k b0), (q b1, k b1), (v b0, v b1), then the 10 blocks of
// W_o.  The next pixel group's input rows are requested when the last head is done and land under the output projection.
// Frames are padded to 32 MFMA columns: at F = 25 78 % of the matrix work is real (the attention itself is 5 % of it).
#include "common.h"
#include "gemm_pp_kernel.h"      // wait_vmcnt / raw_barrier / pp_store_out (store-data hazard guard) / pp_split_io

namespace {

constexpr int kC = 320, kHeads = kC / 64, kKS = kC / 16;         // 20 K steps of 16
constexpr int kChunk = kKS * 1024;                               // 20 KiB: 32 weight rows x 320, one KiB per K step
constexpr int kQkvChunks = 6 * kHeads, kOutChunks = kC / 32, kChunks = kQkvChunks + kOutChunks;   // 30 + 10
constexpr int kPair = 2 * kChunk, kPairs = kChunks / 2;          // the ring moves PAIRS of chunks: one barrier per 40 MFMAs
constexpr int kSlots = 3;
constexpr int kIniOff = kSlots * kPair;                          // per wave: the pixel's 320 start values (bias + row vector), fp32
constexpr int kLnOff = kIniOff + 8 * kC * 4;                      // LayerNorm prologue: gamma | beta (fp32)
constexpr int kSmem = kLnOff + 2 * kC * 4;                        // 120 KiB ring + 10 KiB of start values + 2.5 KiB
constexpr int kNQ = 6;                                           // fragment reads in flight ahead of the MFMAs
constexpr int kDmaPerWave = 2 * kKS / 8;                         // 5 one-KiB pieces per wave and pair
constexpr float kScaleLog2 = 0.125f * 1.44269504088896340736f;   // (1 / sqrt(64)) log2(e)

// Diagnostic build only (-DCTRLV_TA_STAMP, tools/ta_bench.py --stamp): per-wave cycle sums of the phases, written to a buffer
// of their own (TaArgs.stamp; the product build has neither the field's use nor a stamp instruction).
#ifdef CTRLV_TA_STAMP
#define TSTAMP(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#define TSTAMP_ADD(acc, t0, t1) acc += (t1) - (t0)
#else
#define TSTAMP(v)
#define TSTAMP_ADD(acc, t0, t1)
#endif

struct TaArgs {
  const el_t* x; int ldx;                 // LayerNorm output rows [M][ldx]
  const el_t* wf;                         // kChunks x 20 KiB (ctrlv_temporal_fused_pack)
  const float* bias;                      // [320] output-projection bias or null
  const el_t* r1; const el_t* r1_lo; int ldr1;
  const float* vtab; int vmode, vdiv, vmod, vS, ldv, vrows;     // vrows: table rows the launch can address
  el_t* out; el_t* out_lo; int ldo;
  int B, F, S;
  const float* ln_g; const float* ln_b; float ln_eps;     // LN: x holds the RAW rows, normalised in the kernel (else null)
  unsigned long long* stamp;              // diagnostic build: [workgroup][wave][8] cycle sums (else unused)
};

__device__ __forceinline__ elx8 pack_half(const f32x16& a, int ks) {          // accumulators 8 ks .. 8 ks + 7 -> one operand
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = a[8 * ks + e];
  return __builtin_bit_cast(elx8, pack_elx8(f));
}
// the value is materialised HERE: left alone, the scheduler sinks the rounding of a head's q / P / attention output (needed
// only chains later) behind those chains and keeps the fp32 accumulators alive instead -- 64 registers over budget
__device__ __forceinline__ void pin(elx8& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ float swap_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swap_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LN: the block's LayerNorm (norm1 of TemporalBasicTransformerBlock) in the kernel -- x holds the raw rows, a row's 320 values
// sit in its two lanes (hsel = 0 / 1, 160 each: the B fragments as loaded), the two-pass arithmetic of ln_rows_kernel
// (norm.hip: mean, then sum (x - mean)^2; o = (x - mean) rstd gamma + beta, rounded to the element type like that kernel's
// output) runs on them in the Y phase in front of a pixel group's first chain, one half-wave exchange per sum.
template <bool LO, bool LN>
__global__ __launch_bounds__(512) void temporal_fused_kernel(const TaArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  CTRLV_CLOCK_BEGIN();
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int f32_ = lane & 31, hsel = lane >> 5;
  const int F = a.F, S = a.S;
  const long npix = (long)a.B * S;
  const long ngroups = (npix + 7) / 8;
  const int G = gridDim.x;
  constexpr unsigned kOOB = 0xFFFFFFFFu;
  constexpr int kFlags = 0x00020000;
  const long M = (long)a.B * F * S;

  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wf, 0, kChunks * kChunk, kFlags);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(M * a.ldx * 2), kFlags);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(a.r1 ? a.r1 : a.x), 0, a.r1 ? (int)(M * a.ldr1 * 2) : 0, kFlags);
  const __amdgpu_buffer_rsrc_t rsRL = __builtin_amdgcn_make_buffer_rsrc((void*)((LO && a.r1_lo) ? a.r1_lo : a.x), 0,
                                                                        (LO && a.r1_lo) ? (int)(M * a.ldr1) : 0, kFlags);      // (lo planes: one byte per element)
  const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(M * a.ldo * 2), kFlags);
  const __amdgpu_buffer_rsrc_t rsOL = __builtin_amdgcn_make_buffer_rsrc((void*)((LO && a.out_lo) ? a.out_lo : a.out), 0,
                                                                        (LO && a.out_lo) ? (int)(M * a.ldo) : 0, kFlags);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? (const void*)a.bias : (const void*)a.wf), 0,
                                                                       a.bias ? kC * 4 : 0, kFlags);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(a.vmode ? (const void*)a.vtab : (const void*)a.wf), 0,
                                                                       a.vmode ? (int)((long)a.vrows * a.ldv * 4) : 0, kFlags);

  // LDS-DMA of chunk pair `c` (0 .. kPairs - 1, cyclic) into ring slot `g % 3` (g = the workgroup's running pair count): wave w
  // takes pieces w, w + 8, .., w + 32 of the pair's 40
  auto dma = [&](int c, int g) {
#ifdef CTRLV_TA_NODMA       // (diagnostic build: timing without the weight stream; results are garbage)
    return;
#endif
    char* slot = smem + (g % kSlots) * kPair;
#pragma unroll
    for (int k = 0; k < kDmaPerWave; ++k) {
      const int pi = k * 8 + wid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(slot + pi * 1024), 16, (unsigned)(lane * 16), c * kPair + pi * 1024, 0, 0);
    }
  };
  int g = 0;                                                     // finished slots of this workgroup (ring phase)
#ifdef CTRLV_TA_STAMP
  unsigned long long c_bar = 0, c_dma = 0, c_chain = 0, c_wait = 0, c_soft = 0, c_epi = 0, c_x = 0;
  TSTAMP(t_begin);
#endif
  // PING-PONG (the schedule of gemm_pp_kernel.h): the two waves of a SIMD (w, w + 4) are in different GROUPS and group 1 runs
  // HALF A SLOT behind group 0.  A slot is  [barrier] X: the pair's 40 MFMAs  [barrier] Y: everything else -- the roundings /
  // softmax / epilogue stores of the pair, the next pair's loads, the LDS-DMA issue --  and both groups run the SAME program,
  // group 1 behind one extra barrier: while one group's chain has the matrix pipe to itself (36 cycles per MFMA), the other
  // does its Y work beside it.  In lockstep (first versions) both waves of a SIMD were in the same phase: the chains shared
  // the pipe and NOTHING ran beside the Y work -- 47 % of the critical path (stamps, tools/ta_bench.py --stamp).
  // Ring: pair t is read in X(t) -- group 0 in interval 2t, group 1 in 2t + 1 --; DMA(t + 2) refills the slot of pair t - 1
  // and is the LAST thing of Y(t) (interval >= 2t + 1 > every read of pair t - 1); a wave waits for its own pieces (and for
  // everything else it has in flight: vmcnt(0)) at the end of X(t + 1), two barriers in front of the first read of pair t + 2.
  // (the chain's first kNQ fragments: requested in front of the slot's barrier where pair g is known to be complete by then
  //  -- see x_begin -- so that the first MFMA does not wait for the LDS behind the barrier)
  elx8 wq[kNQ];
  auto chain_prefetch = [&]() {
    const char* s1 = smem + (g % kSlots) * kPair + lane * 16;
#pragma unroll
    for (int i = 0; i < kNQ; ++i) wq[i] = *(const elx8*)(s1 + i * 1024);
  };
  // (group 1 only: its X(t) follows the barrier that ends interval 2t, and every wave has waited for its pieces of pair t by
  //  the barrier that ends interval 2t - 1 -- a whole interval earlier.  Group 0's X(t) opens interval 2t itself: group 1's
  //  pieces are only guaranteed behind THAT barrier, so group 0 reads after it.  Reading early in group 0 was a race: seen as
  //  run-to-run differences at the full 72 x 128 size, tests/test_fullsize_gpu.py.)
  const bool early_frags = wid >= 4;
  auto x_begin = [&]() {
    TSTAMP(t0);
    if (early_frags) chain_prefetch();
    lds_done_barrier();
    if (!early_frags) chain_prefetch();
    TSTAMP(t1);
    TSTAMP_ADD(c_bar, t0, t1);
  };
  auto y_begin = [&]() {
    TSTAMP(t0);
    wait_vmcnt<0>();
    TSTAMP(t1);
    lds_done_barrier();
    TSTAMP(t2);
    TSTAMP_ADD(c_wait, t0, t1);
    TSTAMP_ADD(c_bar, t1, t2);
  };
  auto y_end = [&]() {                                           // the slot's DMA issue; g counts finished slots
    TSTAMP(t0);
    dma((g + 2) % kPairs, g + 2);
    ++g;
    TSTAMP(t1);
    TSTAMP_ADD(c_dma, t0, t1);
  };
  // One pair = 2 x 20 MFMAs over K = 320 against the wave's fragments `xb`: accA takes the pair's first chunk, accB its
  // second.  W_IS_A: acc = W . x (lane = frame); else the roles swapped: acc = x . W (lane = channel).  Fragment reads run
  // kNQ MFMAs ahead through rotating registers, straight through the chunk boundary; `mid()` is issued four MFMAs into the
  // second chunk (the first one's result is complete by then: its rounding / dependent MFMAs ride beside the second chain).
  auto chain2 = [&](auto w_is_a, const elx8 (&xb)[kKS], f32x16& accA, f32x16& accB, auto&& mid) {
    constexpr bool W_IS_A = decltype(w_is_a)::value;
    const char* s1 = smem + (g % kSlots) * kPair + lane * 16;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2 * kKS; ++ks) {
      if (ks < kKS) accA = W_IS_A ? mfma_32x32x16(wq[ks % kNQ], xb[ks], accA) : mfma_32x32x16(xb[ks], wq[ks % kNQ], accA);
      else accB = W_IS_A ? mfma_32x32x16(wq[ks % kNQ], xb[ks - kKS], accB) : mfma_32x32x16(xb[ks - kKS], wq[ks % kNQ], accB);
      if (ks + kNQ < 2 * kKS) wq[ks % kNQ] = *(const elx8*)(s1 + (ks + kNQ) * 1024);
      __builtin_amdgcn_sched_barrier(0);
      if (ks == kKS + 3) {
        mid();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // x rows as B fragments: K step ks = channels 16 ks + 8 hsel .. + 7 of the lane's row (frames >= F, pixels past the end: zeros)
  auto x_offset = [&](long grp) {
    const long pix = grp * 8 + wid;
    const bool live = pix < npix;
    const int b = live ? (int)(pix / S) : 0, s = live ? (int)(pix - (long)b * S) : 0;
    const long row = ((long)b * F + f32_) * S + s;
    return (live && f32_ < F) ? (unsigned)(row * a.ldx * 2) + 16u * hsel : kOOB;
  };
  auto x_loads = [&](elx8 (&xr)[kKS], unsigned xoff, int k0, int k1) {
#pragma unroll
    for (int ks = k0; ks < k1; ++ks)
      xr[ks] = __builtin_bit_cast(elx8, __builtin_amdgcn_raw_buffer_load_b128(rsX, xoff, ks * 32, 0));
  };

  const unsigned ln_lds = (unsigned)(unsigned long)LDS_PTR(smem + kLnOff);
  if constexpr (LN) {
    for (int i = threadIdx.x; i < kC; i += 512) {
      *(float*)(smem + kLnOff + i * 4) = a.ln_g[i];
      *(float*)(smem + kLnOff + (kC + i) * 4) = a.ln_b[i];
    }
    __syncthreads();                                             // (before any LDS-DMA is in flight)
  }
  auto layer_norm = [&](elx8 (&xr)[kKS], bool keep) {
    float sm = 0.f;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
      float f[8];
      unpack_elx8(__builtin_bit_cast(uint4, xr[ks]), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) sm += f[e];
      __builtin_amdgcn_sched_barrier(0);                         // (one K step's eight values alive at a time)
    }
    float mean = swap_sum(sm) * (1.0f / kC);
    asm volatile("" : "+v"(mean));
    float sq = 0.f;
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
      // (opaque between the passes: the compiler otherwise keeps the 160 unpacked values of the pass before alive -- 345
      //  spilled registers -- instead of unpacking again)
      pin(xr[ks]);
      float f[8];
      unpack_elx8(__builtin_bit_cast(uint4, xr[ks]), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dl = f[e] - mean; sq += dl * dl; }
      __builtin_amdgcn_sched_barrier(0);
    }
    float rstd = rsqrtf(swap_sum(sq) * (1.0f / kC) + a.ln_eps);
    asm volatile("" : "+v"(rstd));
    const unsigned ga = ln_lds + 32u * hsel;                     // gamma of channels 16 ks + 8 hsel .. + 7 at ga + 64 ks
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) {
      // (asm reads with their own wait: the compiler puts a vmcnt(0) in front of an LDS read it knows of while LDS-DMA of this
      //  wave is in flight, gemm_pp_kernel.h)
      f32x4 g0, g1, b0, b1;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g0) : "v"(ga), "n"(ks * 64) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g1) : "v"(ga), "n"(ks * 64 + 16) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b0) : "v"(ga), "n"(kC * 4 + ks * 64) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b1) : "v"(ga), "n"(kC * 4 + ks * 64 + 16) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g0), "+v"(g1), "+v"(b0), "+v"(b1));
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      pin(xr[ks]);
      float f[8], o[8];
      unpack_elx8(__builtin_bit_cast(uint4, xr[ks]), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f[e] - mean) * rstd * gg[e] + bb[e];
      uint4 pk = pack_elx8(o);
      if (!keep) pk = make_uint4(0, 0, 0, 0);                    // frames >= F / pixels past the end stay zero rows
      xr[ks] = __builtin_bit_cast(elx8, pk);
      pin(xr[ks]);                                               // (computed HERE, not sunk to the chain that uses it)
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: pairs 0 and 1 in flight, the first pixel group's rows requested
  dma(0, 0);
  dma(1, 1);
  elx8 xr[kKS];
  x_loads(xr, x_offset(blockIdx.x), 0, kKS);
  wait_vmcnt<0>();
  if (wid >= 4) raw_barrier();                                   // group 1 starts half a slot behind

  for (long grp = blockIdx.x; grp < ngroups; grp += G) {
    const long pix = grp * 8 + wid;
    const bool live = pix < npix;
    const int b = live ? (int)(pix / S) : 0, s = live ? (int)(pix - (long)b * S) : 0;
    const long row = ((long)b * F + f32_) * S + s;               // this lane's frame of this wave's pixel
    const bool ok = live && f32_ < F;
    elx8 ap[kKS];                                                // attention output, the output projection's B fragments
    unsigned xoff_next = kOOB;
    if constexpr (LN) layer_norm(xr, ok);
    // The output projection's start values (bias + the pixel's row vector; 320 floats per pixel) are requested HERE -- four loads
    // that return under the first chains -- and parked in this wave's LDS strip.  Requested per block in front of its chain
    // (first version), they stood behind the previous block's STORES in the in-order vmcnt queue: every out-projection slot
    // waited for a write acknowledgement before its chain could start (stamps: 6 600 cycles per slot, 27 % of the kernel).
    u32x4_t ib[2], iv[2];
    {
      unsigned vrow = 0;
      if (a.vmode) {
        const long m0 = ((long)b * F) * S + s;
        const long vi = a.vmode == 1 ? (m0 / a.vdiv) % a.vmod : ((m0 / a.vdiv) * a.vS + (m0 % a.vS)) % a.vmod;
        vrow = (unsigned)(vi * a.ldv * 4);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c4 = (t * 64 + lane) * 4;                      // floats c4 .. c4 + 3 (t = 1: lanes 0 .. 15)
        const bool in = c4 < kC;
        ib[t] = __builtin_amdgcn_raw_buffer_load_b128(rsB, in ? (unsigned)(c4 * 4) : kOOB, 0, 0);
        iv[t] = __builtin_amdgcn_raw_buffer_load_b128(rsV, (in && a.vmode) ? vrow + (unsigned)(c4 * 4) : kOOB, 0, 0);
      }
    }
    const unsigned ini_lds = (unsigned)(unsigned long)LDS_PTR(smem + kIniOff + wid * (kC * 4));
#pragma unroll
    for (int h = 0; h < kHeads; ++h) {
      f32x16 sacc = zero16;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        x_begin();
        f32x16 qa = zero16, ka = zero16;
        elx8 q0, q1;
        TSTAMP(t0);
        chain2(std::true_type{}, xr, qa, ka, [&]() {
          q0 = pack_half(qa, 0); q1 = pack_half(qa, 1);          // (rounded beside the k chain: 8 registers, not 16)
          pin(q0); pin(q1);
        });
        TSTAMP(t1);
        TSTAMP_ADD(c_chain, t0, t1);
        y_begin();
        TSTAMP(t2);
        // S^T += K_blk Q_blk^T: both operands as they come out of the chains (same channel-to-slot assignment)
        sacc = mfma_32x32x16(pack_half(ka, 0), q0, sacc);
        sacc = mfma_32x32x16(pack_half(ka, 1), q1, sacc);
        if (h == 0 && blk == 0) {
          // start values -> the wave's strip, behind the group's first chains (asm: the compiler would put a vmcnt(0) in front
          // of an LDS store it knows of while LDS-DMA of this wave is in flight, gemm_pp_kernel.h; the strip's last readers
          // were this wave's own loads of the previous group: LDS operations of one wave execute in order)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const f32x4 sum = {__uint_as_float(ib[t][0]) + __uint_as_float(iv[t][0]), __uint_as_float(ib[t][1]) + __uint_as_float(iv[t][1]),
                               __uint_as_float(ib[t][2]) + __uint_as_float(iv[t][2]), __uint_as_float(ib[t][3]) + __uint_as_float(iv[t][3])};
            if (t == 0 || lane < 16)
              asm volatile("ds_write_b128 %0, %1" ::"v"(ini_lds + (unsigned)((t * 64 + lane) * 16)), "v"(sum) : "memory");
          }
        }
        if (blk == 0) {
          TSTAMP(t3);
          TSTAMP_ADD(c_soft, t2, t3);
          y_end();
        }
      }
      // ---- softmax over the keys (accumulator e of lane (query, hsel) = key 8 (e >> 2) + 4 hsel + (e & 3))
      TSTAMP(ts0);
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * hsel;
        if (key >= F) sacc[e] = -INFINITY;
        mx = fmaxf(mx, sacc[e]);
      }
      mx = swap_max(mx) * kScaleLog2;
      float rs_ = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(sacc[e] * kScaleLog2 - mx);
        sacc[e] = p;
        rs_ += p;
      }
      const float inv = __builtin_amdgcn_rcpf(swap_sum(rs_));
      elx8 pf0 = pack_half(sacc, 0), pf1 = pack_half(sacc, 1);
      pin(pf0); pin(pf1);
      TSTAMP(ts1);
      TSTAMP_ADD(c_soft, ts0, ts1);
      y_end();
      {
        x_begin();
        f32x16 va = zero16, vb = zero16, oa;                      // V^T blocks: lane = channel, frames 8 q + 4 hsel + r
        TSTAMP(t0);
        chain2(std::false_type{}, xr, va, vb, [&]() {
          oa = mfma_32x32x16(pack_half(va, 0), pf0, zero16);
          oa = mfma_32x32x16(pack_half(va, 1), pf1, oa);
        });
        TSTAMP(t1);
        TSTAMP_ADD(c_chain, t0, t1);
        y_begin();
        TSTAMP(t2);
        f32x16 ob = mfma_32x32x16(pack_half(vb, 0), pf0, zero16);
        ob = mfma_32x32x16(pack_half(vb, 1), pf1, ob);
#pragma unroll
        for (int e = 0; e < 16; ++e) { oa[e] *= inv; ob[e] *= inv; }
        ap[h * 4] = pack_half(oa, 0); ap[h * 4 + 1] = pack_half(oa, 1);
        ap[h * 4 + 2] = pack_half(ob, 0); ap[h * 4 + 3] = pack_half(ob, 1);
        pin(ap[h * 4]); pin(ap[h * 4 + 1]); pin(ap[h * 4 + 2]); pin(ap[h * 4 + 3]);
        TSTAMP(t3);
        TSTAMP_ADD(c_soft, t2, t3);
        if (h == kHeads - 1) {
          // the NEXT pixel group's rows: the x registers are free from here on -- four loads per slot over this slot and the
          // first four of the output projection (the CU's vector-memory path takes 64 B per clock: all twenty in one Y phase
          // made that phase six times as long as the chain beside it), so that they have landed a slot before the next group
          xoff_next = x_offset(grp + G);
          if constexpr (!LO) x_loads(xr, xoff_next, 0, 4);
        }
        y_end();
      }
    }
    // ---- output projection, 10 blocks of 32 channels in pairs; lane (frame, hsel) ends with channels 32 nb + 16 hsel .. + 15
    const unsigned rbase = ok ? (unsigned)(row * a.ldr1 * 2) + 32u * hsel : kOOB;
    const unsigned obase = ok ? (unsigned)(row * a.ldo * 2) + 32u * hsel : kOOB;
#pragma unroll
    for (int np = 0; np < kOutChunks / 2; ++np) {
      TSTAMP(te0);
      // start values (bias + the pixel's row vector) and the residual rows of both blocks: requested in the Y phase in front of
      // the chains' barrier
      f32x16 acc[2];
      u32x4_t rr[2][2];
      u32x2_t rl[2][2];                                      // (lo planes: 8 bytes per 8 elements, byte offsets halved)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          rr[t][hq] = __builtin_amdgcn_raw_buffer_load_b128(rsR, rbase, (2 * np + t) * 64 + hq * 16, 0);
          if (LO) rl[t][hq] = __builtin_amdgcn_raw_buffer_load_b64(rsRL, rbase == kOOB ? kOOB : (rbase >> 1), (2 * np + t) * 32 + hq * 8, 0);
        }
      {
        // the lane's 2 x 16 start values from the strip (channels 32 nb + 16 hsel .. + 15 of the two blocks)
        f32x4 iq[2][4];
        const unsigned ia = ini_lds + (unsigned)((2 * np * 32 + 16 * hsel) * 4);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(iq[t][q]) : "v"(ia), "n"(t * 128 + q * 16) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(iq[0][0]), "+v"(iq[0][1]), "+v"(iq[0][2]), "+v"(iq[0][3]), "+v"(iq[1][0]), "+v"(iq[1][1]), "+v"(iq[1][2]),
                       "+v"(iq[1][3]));
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[t][4 * q] = iq[t][q].x; acc[t][4 * q + 1] = iq[t][q].y; acc[t][4 * q + 2] = iq[t][q].z; acc[t][4 * q + 3] = iq[t][q].w;
          }
      }
      TSTAMP(te1);
      TSTAMP_ADD(c_epi, te0, te1);
      x_begin();
      TSTAMP(tc1);
      chain2(std::true_type{}, ap, acc[0], acc[1], []() {});
      TSTAMP(te2);
      TSTAMP_ADD(c_chain, tc1, te2);
      y_begin();
      TSTAMP(te2b);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          float o[8], rf[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = acc[t][8 * hq + e];
          unpack_elx8(make_uint4(rr[t][hq][0], rr[t][hq][1], rr[t][hq][2], rr[t][hq][3]), rf);
          {
#pragma clang fp contract(off)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = o[e] + rf[e];
          }
          if (LO) add_lo8(o, make_uint2(rl[t][hq][0], rl[t][hq][1]));
          const uint4 pk = pack_elx8(o);
          const u32x4_t pv = {pk.x, pk.y, pk.z, pk.w};
#ifndef CTRLV_TA_NOSTORE
          pp_store_out(pv, rsO, obase, (2 * np + t) * 64 + hq * 16);
#else
          asm volatile("" ::"v"(pv));
#endif
          if (LO) {
            const uint2 pl = split_lo8(o, pk);
            const u32x2_t pvl = {pl.x, pl.y};
            pp_store_out_lo(pvl, rsOL, obase == kOOB ? kOOB : (obase >> 1), (2 * np + t) * 32 + hq * 8);
          }
        }
      TSTAMP(te3);
      TSTAMP_ADD(c_epi, te2b, te3);
      // (split planes: two more residual / output planes are alive across the chains and the x registers are needed for them
      //  -- the rows are requested behind the last epilogue instead, where nothing else is alive)
      if constexpr (!LO) { if (np < kOutChunks / 2 - 1) x_loads(xr, xoff_next, np * 4 + 4, np * 4 + 8); }
      else { if (np == kOutChunks / 2 - 1) x_loads(xr, xoff_next, 0, kKS); }
      TSTAMP(te4);
      TSTAMP_ADD(c_x, te3, te4);
      y_end();
    }
  }
  if (wid < 4) raw_barrier();                                    // group 0's closing barrier (pairs with group 1's last one)
  wait_vmcnt<0>();                                               // the look-ahead DMA: nothing may be in flight at exit
  CTRLV_CLOCK_END();
#ifdef CTRLV_TA_STAMP
  TSTAMP(t_end);
  if (lane == 0 && a.stamp) {
    unsigned long long* o = a.stamp + ((long)blockIdx.x * 8 + wid) * 8;
    o[0] = t_end - t_begin; o[1] = c_bar; o[2] = c_dma; o[3] = c_chain; o[4] = c_wait; o[5] = c_soft; o[6] = c_epi; o[7] = c_x;
  }
#endif
#endif
}

// fragment-major copy of the packed weights.  Chunk c < 30: head h = c / 6, kind = c % 6 = (q b0, k b0, q b1, k b1, v b0,
// v b1): rows of the fused [3C][C] projection.  Chunk 30 + nb: block nb of W_o with its rows permuted (A row 8 q + 4 hs + r
// holds output channel 32 nb + 16 hs + 4 q + r) and its K steps in the order the attention output arrives:
// step jj = (head, block, ks), slot (hs', s) <-> input channel 64 head + 32 block + 8 (2 ks + (s >> 2)) + 4 hs' + (s & 3).
__global__ void temporal_pack_kernel(const el_t* __restrict__ wqkv, int ld_qkv, const el_t* __restrict__ wo, int ld_o,
                                     el_t* __restrict__ wf) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)kChunks * kKS * 512) return;
  const int e = i & 7, lane = (i >> 3) & 63, ks = (int)((i >> 9) % kKS), c = (int)(i / (kKS * 512));
  const int i32 = lane & 31, hs = lane >> 5;
  el_t v;
  if (c < kQkvChunks) {
    const int h = c / 6, kind = c % 6;
    const int which = kind >= 4 ? 2 : (kind & 1), blk = kind >= 4 ? kind - 4 : kind >> 1;
    const int rowi = which * kC + h * 64 + blk * 32 + i32;
    v = wqkv[(long)rowi * ld_qkv + ks * 16 + 8 * hs + e];
  } else {
    const int nb = c - kQkvChunks;
    const int q = i32 >> 3, hs_r = (i32 >> 2) & 1, r = i32 & 3;            // A row i32 = 8 q + 4 hs_r + r
    const int n = nb * 32 + 16 * hs_r + 4 * q + r;
    const int head = ks >> 2, blk = (ks >> 1) & 1, k2 = ks & 1;
    const int d = 64 * head + 32 * blk + 8 * (2 * k2 + (e >> 2)) + 4 * hs + (e & 3);
    v = wo[(long)n * ld_o + d];
  }
  wf[i] = v;
}

}  // namespace

CTRLV_CLOCK_READER(temporal_fused)

extern "C" size_t ctrlv_temporal_fused_weight_bytes(void) { return (size_t)kChunks * kChunk; }

extern "C" int ctrlv_temporal_fused_pack(const void* wqkv_packed, int ld_qkv, const void* wo_packed, int ld_o, void* wf,
                                         ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(wqkv_packed && wo_packed && wf, "ctrlv_temporal_fused_pack: null pointer");
  CTRLV_CHECK_SHAPE(ld_qkv >= kC && ld_o >= kC, "ctrlv_temporal_fused_pack: the packed weights must have K >= 320");
  const long n = (long)kChunks * kKS * 512;
  hipLaunchKernelGGL(temporal_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const el_t*)wqkv_packed, ld_qkv, (const el_t*)wo_packed, ld_o, (el_t*)wf);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

// the launcher's conditions (one function for it and for the callers' switch, ctrlv_temporal_fused_serves)
static int ta_check(const ctrlv_temporal_fused_desc& d, bool report) {
#define TA_REQ(cond, code, ...)                        \
  do {                                                 \
    if (!(cond)) {                                     \
      if (report) ctrlv_set_error(__VA_ARGS__);        \
      return code;                                     \
    }                                                  \
  } while (0)
  TA_REQ(d.C == kC, CTRLV_E_BAD_SHAPE, "ctrlv_temporal_fused: serves C = 320 only (C=%d)", d.C);
  TA_REQ(d.B > 0 && d.F > 0 && d.F <= 32 && d.S > 0, CTRLV_E_BAD_SHAPE, "ctrlv_temporal_fused: B, S > 0 and 0 < F <= 32 (F=%d)", d.F);
  TA_REQ(d.ldx >= kC && d.ldx % 8 == 0 && d.ldo >= kC && d.ldo % 8 == 0 && (!d.R1 || (d.ldr1 >= kC && d.ldr1 % 8 == 0)),
         CTRLV_E_BAD_SHAPE, "ctrlv_temporal_fused: row pitches must be >= 320 and multiples of 8");
  const long M = (long)d.B * d.F * d.S, lim = 0xFFFFFFF0L;
  TA_REQ(M * d.ldx * 2 <= lim && M * d.ldo * 2 <= lim && (!d.R1 || M * d.ldr1 * 2 <= lim), CTRLV_E_BAD_SHAPE,
         "ctrlv_temporal_fused: operands beyond 32-bit byte offsets");
  TA_REQ((!d.R1_lo || d.R1) && ((!d.R1_lo && !d.out_lo) || CTRLV_ELEM_DTYPE == 1), CTRLV_E_BAD_ARG,
         "ctrlv_temporal_fused: split trunk planes need R1 and the fp16 element library");
  TA_REQ((d.ln_gamma == nullptr) == (d.ln_beta == nullptr), CTRLV_E_BAD_ARG, "ctrlv_temporal_fused: LayerNorm needs gamma and beta");
  if (d.vmode) {
    // the row vector must be constant over a pixel's frames: one table row per clip (vmode 1) or per (pixel, clip) (vmode 2)
    TA_REQ((d.vmode == 1 || d.vmode == 2) && d.V && d.vmod > 0 && d.ldv >= kC && d.ldv % 4 == 0 && d.vdiv == d.F * d.S &&
               (d.vmode == 1 || d.vS == d.S),
           CTRLV_E_BAD_ARG, "ctrlv_temporal_fused: the row vector must be per clip (vdiv = F S; vmode 2: vS = S)");
    const long vrows = d.vmode == 1 ? (long)d.B : (long)d.B * d.S;       // table rows the launch can address (vmod may be "none")
    TA_REQ((vrows < d.vmod ? vrows : (long)d.vmod) * d.ldv * 4 <= lim, CTRLV_E_BAD_SHAPE, "ctrlv_temporal_fused: row-vector table too large");
  }
  return CTRLV_OK;
#undef TA_REQ
}

#ifdef CTRLV_TA_STAMP
static unsigned long long* g_ta_stamp = nullptr;
extern "C" void ctrlv_temporal_fused_set_stamp(unsigned long long* buf) { g_ta_stamp = buf; }
#endif

extern "C" int ctrlv_temporal_fused_serves(const ctrlv_temporal_fused_desc* d) {
  return (d && ta_check(*d, false) == CTRLV_OK) ? 1 : 0;
}

extern "C" int ctrlv_temporal_fused(const ctrlv_temporal_fused_desc* dp, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(dp && dp->x && dp->wf && dp->out, "ctrlv_temporal_fused: null pointer");
  const ctrlv_temporal_fused_desc& d = *dp;
  const int rc = ta_check(d, true);
  if (rc != CTRLV_OK) return rc;
  TaArgs a;
  a.x = (const el_t*)d.x; a.ldx = d.ldx; a.wf = (const el_t*)d.wf; a.bias = d.bias;
  a.r1 = (const el_t*)d.R1; a.r1_lo = (const el_t*)d.R1_lo; a.ldr1 = d.R1 ? d.ldr1 : d.ldx;
  a.vtab = d.V; a.vmode = d.V ? d.vmode : 0; a.vdiv = d.vdiv > 0 ? d.vdiv : 1; a.vmod = d.vmod > 0 ? d.vmod : 1;
  a.vS = d.vS > 0 ? d.vS : 1; a.ldv = d.ldv;
  a.vrows = 0;
  if (a.vmode) {
    const long vrows = a.vmode == 1 ? (long)d.B : (long)d.B * d.S;
    a.vrows = (int)(vrows < a.vmod ? vrows : (long)a.vmod);
  }
  a.out = (el_t*)d.out; a.out_lo = (el_t*)d.out_lo; a.ldo = d.ldo;
  a.B = d.B; a.F = d.F; a.S = d.S;
  a.ln_g = d.ln_gamma; a.ln_b = d.ln_beta; a.ln_eps = d.ln_eps;
  a.stamp = nullptr;
#ifdef CTRLV_TA_STAMP
  a.stamp = g_ta_stamp;
#endif
  const int dev = ctrlv_current_device();
  const long groups = ((long)d.B * d.S + 7) / 8;
  const int num_cu = ctrlv_num_cu(dev);
  long grid = groups;
  if (groups > num_cu) {                    // persistent, every workgroup the same number of pixel groups
    const long rounds = (groups + num_cu - 1) / num_cu;
    grid = (groups + rounds - 1) / rounds;
  }
  const bool lo = d.R1_lo || d.out_lo;
  const bool ln = d.ln_gamma != nullptr;
  static bool attr_set[2][2][CTRLV_MAX_DEVICES] = {};
#define TA_LAUNCH(LOV, LNV)                                                                                             \
  do {                                                                                                                  \
    auto kfn = temporal_fused_kernel<LOV, LNV>;                                                                         \
    if (!attr_set[LOV][LNV][dev]) {                                                                                     \
      CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));          \
      attr_set[LOV][LNV][dev] = true;                                                                                   \
    }                                                                                                                   \
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), kSmem, (hipStream_t)stream, a);                            \
  } while (0)
#ifdef CTRLV_ELEM_F16
  if (lo && ln) TA_LAUNCH(true, true);     // (the in-kernel LayerNorm normalises the rows of x: with a split trunk, its hi plane)
  else if (lo) TA_LAUNCH(true, false);
  else if (ln) TA_LAUNCH(false, true);
  else TA_LAUNCH(false, false);
#else
  (void)lo;
  if (ln) TA_LAUNCH(false, true); else TA_LAUNCH(false, false);
#endif
#undef TA_LAUNCH
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
