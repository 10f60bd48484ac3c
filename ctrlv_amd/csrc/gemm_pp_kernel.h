// Ping-pong gather-GEMM for gfx950: 256 x BN x 32 half-steps, 8 waves, 4-slot LDS-DMA ring, counted vmcnt.
//
// Same contract as gemm.hip (ctrlv_gemm_desc), different schedule.  The 2-stage kernel of gemm.hip drains
// `vmcnt(0)` and barriers once per K-step with both waves of every SIMD in the same phase, which caps it at
// ~0.3 of the MFMA peak (profiles/r01_*).  Here:
//   * the two waves that share a SIMD (wave w and w+4) are put in different GROUPS and group 1 runs one barrier
//     slot behind group 0: while one group issues its 16-20 MFMAs from registers, the other does its LDS fragment
//     reads, its LDS-DMA issue and its waits -- matrix pipe and memory path of a SIMD alternate between its two waves
//     (MI355X_MICROARCH "Two waves per SIMD"; guide T3/T4).
//   * K advances in half-steps of 32; a ring of 4 LDS slots (4 x (256+BN) x 64 B) keeps 3 half-steps of LDS-DMA in
//     flight across the raw s_barriers; waits are COUNTED (`s_waitcnt vmcnt(2*per_wave)`), never 0 in steady state.
//   * hazards: the ring slot of half-step j is read in phase L_j (group 0 in barrier slot 2j, group 1 in 2j+1) and
//     re-filled by the DMA of half-step j+4, issued in L_{j+1} / C_{j+1}; every wave retires its own DMA(j+1) before
//     the barrier that ends its L_j and finishes its ds_reads (lgkmcnt(0)) before that same barrier -- RAW and WAR are
//     each separated by >= 1 barrier that all 8 waves pass.  The issue stream simply runs three half-steps past the
//     block's last tile (all rows out of range: zeros, no memory traffic), so the loop has no tail cases.
//   * the hot loop is instruction-issue-bound: lane offsets are per tile, a half-step adds scalars.
//   * 64-B LDS rows (32 bf16): physical 16-B chunk = logical ^ ((row>>2)&3) keeps ds_read_b128 conflict-free; the
//     swizzle is applied on the per-lane DMA source address.
//   * PERSISTENT: one workgroup per CU walks its tiles (XCD-contiguous windows) and the DMA ring runs straight
//     through tile boundaries -- the first three half-steps of tile T+1 are already in flight while tile T finishes and
//     its epilogue (stores, GEGLU) runs, so short-K layers (K = 320: only 10 half-steps) no longer pay an exposed
//     prologue per tile.  Group 1 stays one barrier slot behind group 0 across tiles.
// Tiles: BN = 256 (waves 2x4, wave tile 128x64) and BN = 320 (waves 4x2, wave tile 64x160) -- the latter makes
// N = 320 / 640 / 960 / 1920 exact multiples (the C = 320 / 640 levels of the UNet).
#pragma once
#include <type_traits>

#include "common.h"


// K order (dy, 32-channel block, dx) + row-halo staging for this 3x3 conv?  One decision for every kernel that may serve
// the layer (defined in gemm_pp_m0.hip: conv_halo_geometry() below and the CTRLV_CONV_HALO switch, default on).
bool ctrlv_conv_halo_order(const ctrlv_gemm_desc& d);

namespace {

// Diagnostic build only (-DCTRLV_PP_STAMP, tools/gemm_stamp.py): per-wave cycle sums of the phases of the schedule,
// written to the buffer passed in d.V with vmode == 0 (unused by the profiled launch).  The shipped library contains no stamp.
#ifdef CTRLV_PP_STAMP
#define STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define STAMP_ADD(acc, a, b) acc += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(acc, a, b)
#endif

// counted LDS wait that "defines" the four registers it covers (the consumer cannot be scheduled above it)
template <int N>
__device__ __forceinline__ void wait_lgkm_tied(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N < 15 ? N : 15));
}
// the TN x 4 strip reads of a tile's bias start (one address register, the displacement as the instruction's offset)
template <int TN, int N = 0>
__device__ __forceinline__ void bias_read_blocks(f32x4 (&bq)[TN][4], unsigned base) {
  if constexpr (N < TN) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[N][0]) : "v"(base), "n"(N * 128) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[N][1]) : "v"(base), "n"(N * 128 + 32) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[N][2]) : "v"(base), "n"(N * 128 + 64) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[N][3]) : "v"(base), "n"(N * 128 + 96) : "memory");
    bias_read_blocks<TN, N + 1>(bq, base);
  }
}
// sub-tile n of TN bias blocks (four reads each, issued in order n = 0..TN-1): wait until block n has landed
template <int TN, int N = 0>
__device__ __forceinline__ void bias_wait_blocks(f32x4 (&bq)[TN][4]) {
  if constexpr (N < TN) {
    wait_lgkm_tied<4 * (TN - 1 - N)>(bq[N][0], bq[N][1], bq[N][2], bq[N][3]);
    bias_wait_blocks<TN, N + 1>(bq);
  }
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_done_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void raw_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Epilogue staging accesses as asm statements.  While LDS-DMA of a wave is in flight the compiler puts an `s_waitcnt
// vmcnt(0)` in front of the first LDS store (and of LDS reads it cannot tell apart from the DMA's targets) it knows of --
// and at the top of an epilogue the next tile's first pieces AND the residual rows just requested are in flight: every tile
// began its epilogue by waiting for all of them (HBM latency), the prefetch window of the residuals defeated for its first
// sub-tiles.  The staging pieces are never a DMA target while the epilogue runs (the just-consumed ring slot / a strip of
// the wave's own), so no wait is needed; LDS operations of one wave execute in order, the reads carry their own lgkmcnt
// wait.  -DCTRLV_PP_ASM_STAGING=0: plain C++ accesses (A/B handle for tools/ab_build.py).
#ifndef CTRLV_PP_ASM_STAGING
#define CTRLV_PP_ASM_STAGING 1
#endif
__device__ __forceinline__ void stg_write16(char* p, float a, float b, float c, float d) {
#if CTRLV_PP_ASM_STAGING
  const f32x4 v = {a, b, c, d};
  asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(unsigned long)LDS_PTR(p)), "v"(v) : "memory");
#else
  *(float4*)p = make_float4(a, b, c, d);
#endif
}
// the four 16-byte reads of a sub-tile's two passes (issued together, one wait)
__device__ __forceinline__ void stg_read4x16(const char* a0, const char* a1, const char* b0, const char* b1, float4 (&img)[2][2]) {
#if CTRLV_PP_ASM_STAGING
  f32x4 r0, r1, r2, r3;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r0) : "v"((unsigned)(unsigned long)LDS_PTR(a0)) : "memory");
  asm volatile("ds_read_b128 %0, %1" : "=v"(r1) : "v"((unsigned)(unsigned long)LDS_PTR(a1)) : "memory");
  asm volatile("ds_read_b128 %0, %1" : "=v"(r2) : "v"((unsigned)(unsigned long)LDS_PTR(b0)) : "memory");
  asm volatile("ds_read_b128 %0, %1" : "=v"(r3) : "v"((unsigned)(unsigned long)LDS_PTR(b1)) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)::"memory");
  img[0][0] = make_float4(r0.x, r0.y, r0.z, r0.w); img[0][1] = make_float4(r1.x, r1.y, r1.z, r1.w);
  img[1][0] = make_float4(r2.x, r2.y, r2.z, r2.w); img[1][1] = make_float4(r3.x, r3.y, r3.z, r3.w);
#else
  img[0][0] = *(const float4*)a0; img[0][1] = *(const float4*)a1;
  img[1][0] = *(const float4*)b0; img[1][1] = *(const float4*)b1;
#endif
}

// Epilogue through a per-wave LDS transpose.  In the MFMA result layout a lane owns one output ROW of a 32x32
// sub-tile, so direct stores are 8 bytes per lane at a row stride: every store / residual-load instruction touches
// 32 cache lines and the epilogue of a short-K tile cost more than its main loop (tools/gemm_stamp.py).  Here each
// 32x32 fp32 sub-tile goes through 4 KiB of wave-private LDS (the four 1-KiB ring pieces of the just-consumed slot that
// only THIS wave's next DMA refills) and comes back with 8 consecutive columns per lane: bias / residual / row-vector
// reads and the bf16 store are 16-32 B per lane and row-contiguous.
//
// The whole epilogue is STRAIGHT-LINE code: every global access is a buffer_load / buffer_store whose per-lane offset
// is forced out of range for rows >= M and columns >= n_store (the hardware range check drops the store / returns
// zeros), so there is no divergent control flow.  That matters because on gfx9 `vmcnt` counts loads AND stores in
// order: with exec-masked branches around the stores the compiler could only wait `vmcnt(0)` for a prefetched
// operand, i.e. for the write acknowledgement of the previous sub-tile's stores (~1300 cycles per sub-tile in the
// stamps).  Branch-free, its counted waits let stores drain in the background while operands for sub-tile s+PF are
// loaded PF sub-tiles ahead.  LDS accesses of one wave execute in order, so the staging write -> read -> next write
// sequence needs no explicit waits either.
// EPI: compile-time operand set -- bit 0: row-vector table V, bit 1: residual R1, bit 2: residual R2 (bias and s_acc
// are always honoured).  SiLU / fp32 output are served by the 128x128 kernel of gemm.hip only.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// ROW-HALO staging of the 3x3 gather (round 4; north_star: "convs with coalesced HBM reads and LDS halo staging").
// The three horizontal taps (dx = -1, 0, +1) of one (dy, 32-channel block) read the SAME input pixels shifted by one:
// for stride 1 without upsampling and a row width that divides the 256-row tile, the tile's R = 256 / W image rows are
// staged ONCE per (dy, channel block) as R rows of W + 2 pixels (a zero column on either side, written by out-of-range
// DMA lanes) and the three half-steps read their fragments at pixel offsets 0 / 1 / 2 -- 17-18 KiB through the texture
// path per three half-steps instead of 48, one A piece per wave and half-step instead of two.  Why it matters
// (profiles/r04_conv3x3_phase_stamps.txt): the CU's vector-memory path takes 64 B per clock, i.e. 576 cycles for the
// 36 KiB of a 256 x 320 half-step against 640 cycles of MFMA work -- LDS-DMA issue was 400 of the load phase's 780
// cycles.  The K traversal of such a layer is (dy, channel block, dx) in EVERY kernel that serves it (conv_halo_geometry
// is a function of the layer, never of M; gemm.hip follows the same order), so the bits do not depend on the kernel.
__host__ __device__ inline bool conv_halo_geometry(const ctrlv_gemm_desc& d) {
  // (rows narrower than 32 pixels: a wave's 32 lanes straddle slot rows, the shifted fragment reads collide in the LDS
  //  banks and the M = 7200 layers of the 9 x 16 level lost 5 % inside the model: per-tap gather there)
  return d.mode == 1 && d.taps == 9 && d.stride == 1 && d.up == 0 && d.Wd >= 32 && d.Wd <= 256 && (256 % d.Wd) == 0 &&
         d.A2 == nullptr;
}

// rows of the row-vector table V that rows [0, M) can address (vmod may be "no modulus" = 1 << 30)
__host__ __device__ inline long pp_vtable_rows(const ctrlv_gemm_desc& d) {
  const long groups = (long)(d.M - 1) / d.vdiv + 1;
  const long reach = d.vmode == 1 ? groups : groups * d.vS;
  return reach < d.vmod ? reach : (long)d.vmod;
}

// Bias lives in a wave-private LDS strip (WTN floats: the columns of this wave's tile), filled before the first tile
// and re-filled at the end of every epilogue with the NEXT tile's columns (loaded into 4 VGPRs at the start of the
// epilogue, so its latency hides behind the whole epilogue).  The K loop reads it once per tile as the C operand of
// the tile's first MFMAs (bias_c in the kernel), so the accumulators already contain acc + bias when the epilogue runs.
template <int WTN>
__device__ __forceinline__ u32x4_t pp_bias_load(const ctrlv_gemm_desc& d, int wbase_n, int lane, bool zero = false) {
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(d.bias ? (const void*)d.bias : d.W), 0, d.bias ? d.N * 4 : 0, 0x00020000);   // no bias: reads 0
  // (zero: K slices after the first start from 0 -- the bias belongs to slice 0)
  return __builtin_amdgcn_raw_buffer_load_b128(rsB, (lane < WTN / 4 && !zero) ? (unsigned)((wbase_n + lane * 4) * 4) : 0xFFFFFFFFu,
                                               0, 0);
}
// STORE-DATA HAZARD (root cause of two "intermittent wrong dword" defects).  A buffer_store_dwordx4
// reads its four data VGPRs over several cycles AFTER it has issued; a VALU instruction that overwrites one of them within
// the next two issue slots replaces the bytes that leave the CU, for some lanes (the ones whose quad the store has not
// read yet) -- seen as ZERO dwords (a zero-initialisation scheduled right behind the store: the EPI = 3 / TN = 10
// instantiation of round 3) and as raw fp32 bit patterns in an fp16 output (a v_pk_mul_f32 of the next row pass right
// behind it: <256,2,4,0,false,0> of the fp16 build).  LLVM's hazard recognizer knows the hazard (2 wait states on gfx940+)
// but EXEMPTS stores whose soffset is an SGPR -- which is exactly the form this epilogue uses since its addressing moved the
// sub-tile displacement into the scalar offset (round 3); gfx950 has the hazard for that form too.  So the two wait states
// are written out: an `s_nop 1` that takes the stored registers as operands -- the register allocator cannot hand them to
// another value before it, the scheduler cannot move a redefinition above it.
__device__ __forceinline__ void store_data_hazard_guard(const u32x4_t& pv) {
  asm volatile("s_nop 1" ::"v"(pv));
}
__device__ __forceinline__ void pp_store_out(const u32x4_t& pv, __amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(pv, rs, voff, soff, 0);
  store_data_hazard_guard(pv);
}
// the eight lo bytes of a split output (one byte per element: byte offsets are the element offsets)
__device__ __forceinline__ void pp_store_out_lo(const u32x2_t& pv, __amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b64(pv, rs, voff, soff, 0);
  asm volatile("s_nop 1" ::"v"(pv));
}

template <int WTN>
__device__ __forceinline__ void pp_bias_store(char* bias_lds, const u32x4_t& v, int lane) {
  // (asm for the reason given at stg_write16: the refill behind an epilogue runs with the next tile's LDS-DMA in flight)
#if CTRLV_PP_ASM_STAGING
  if (lane < WTN / 4) asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(unsigned long)LDS_PTR(bias_lds + lane * 16)), "v"(v) : "memory");
#else
  if (lane < WTN / 4) *(u32x4_t*)(bias_lds + lane * 16) = v;
#endif
}

// GNS (producer-side GroupNorm statistics, round 4): the launch also writes, per 64-row wave tile and group of N / 32
// output channels, the (mean, M2) of the values it stores -- the chunk partials gn_finalize_kernel combines (norm.hip),
// in place of a gn_stats pass that re-reads the tensor.  Per lane: sums and squares of its 8 channels over the wave
// tile's rows of column block j (sub-tiles are walked column block by column block); per column block one round trip
// through the wave's staging pieces turns the 64 lanes' sums into 32 column sums + 32 column square sums; after the last
// block the wave's WTN column sums go through the staging pieces once more and WTN / cpg lanes fold their group's columns.
// Needs WTN % cpg == 0 and a wave tile inside one image (S % 64 == 0): ctrlv_gemm_gn_partials_serves.
// LO (round 5): SPLIT residual-trunk planes (include/ctrlv_hip.h R1_lo / R2_lo / out_lo): R1 / R2 are read as hi + lo (one
// fma per plane, hi first -- gemm_epilogue.h's sequence) and the output goes out as hi = rne(v) plus lo = rne(v - hi).
template <int TM, int TN, bool GEGLU, int EPI, bool RAW = false, bool GNS = false, bool LO = false>
__device__ __forceinline__ void gemm_epilogue_lds(const ctrlv_gemm_desc& d, f32x16 (&acc)[TM][TN], int bm, int bn,
                                                  int wr, int wc, int WTM, int WTN, int lane, char* p0, char* p1,
                                                  char* p2, char* p3, const char* bias_lds, const char* gelu_tab,
                                                  char* gns_strip = nullptr) {
  constexpr unsigned kOOB = 0xFFFFFFFFu;
  constexpr int kFlags = 0x00020000;
  const int r32 = lane & 31, hsel = lane >> 5, l4 = lane & 3;
  // staging image: row r (0..31) lives in piece r>>3 at (r&7)*128 B; 16-B chunk c of a row is stored at c ^ (r&7)
  char* const wpiece = (r32 < 8) ? p0 : (r32 < 16) ? p1 : (r32 < 24) ? p2 : p3;
  char* const wrow = wpiece + (r32 & 7) * 128;
  const int row_a = lane >> 2, row_b = 16 + row_a;   // rows read back in pass 0 / pass 1 (same r&7)
  const char* const rp_a = (row_a < 8 ? p0 : p1) + (row_a & 7) * 128;
  const char* const rp_b = (row_b < 24 ? p2 : p3) + (row_b & 7) * 128;
  const int rx0 = ((l4 * 2) ^ (row_a & 7)) * 16, rx1 = ((l4 * 2 + 1) ^ (row_a & 7)) * 16;
  const int wbase_n = bn + wc * WTN;                 // first column of this wave's tile (weight-row order)
  const int m0 = bm + wr * WTM + row_a;              // + i*32 + pass*16
  constexpr int NSUB = TM * TN;

  const __amdgpu_buffer_rsrc_t rsO =
      __builtin_amdgcn_make_buffer_rsrc(d.out, 0, (int)((long)d.M * d.ldo * 2), kFlags);
  const __amdgpu_buffer_rsrc_t rsR1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((EPI & 2) ? d.R1 : d.W), 0, (EPI & 2) ? (int)((long)d.M * d.ldr1 * 2) : 0, kFlags);
  const __amdgpu_buffer_rsrc_t rsR2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((EPI & 4) ? d.R2 : d.W), 0, (EPI & 4) ? (int)((long)d.M * d.ldr2 * 2) : 0, kFlags);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((EPI & 1) ? (const void*)d.V : d.W), 0, (EPI & 1) ? (int)(pp_vtable_rows(d) * d.ldv * 4) : 0, kFlags);
  // (LO: a missing lo plane is an empty descriptor -- its loads return zeros, hi + 0.  lo planes hold ONE BYTE per element
  //  (common.h lo_t): their byte offsets are the hi plane's halved)
  const __amdgpu_buffer_rsrc_t rsOL = __builtin_amdgcn_make_buffer_rsrc(
      (LO && d.out_lo) ? d.out_lo : (void*)d.W, 0, (LO && d.out_lo) ? (int)((long)d.M * d.ldo) : 0, kFlags);
  const __amdgpu_buffer_rsrc_t rsR1L = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((LO && (EPI & 2) && d.R1_lo) ? d.R1_lo : d.W), 0, (LO && (EPI & 2) && d.R1_lo) ? (int)((long)d.M * d.ldr1) : 0, kFlags);
  const __amdgpu_buffer_rsrc_t rsR2L = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((LO && (EPI & 4) && d.R2_lo) ? d.R2_lo : d.W), 0, (LO && (EPI & 4) && d.R2_lo) ? (int)((long)d.M * d.ldr2) : 0, kFlags);

  // byte offset of the row-vector table row of (i, pass): two integer divisions per row, hoisted out of the sub-tiles
  // (GNS: the launcher guarantees vmode 1 with vdiv a multiple of the 64-row wave tile: one table row per wave tile, a scalar)
  unsigned v_row[GNS ? 1 : TM][2];
  unsigned v_row_s = 0;
  if constexpr (GNS && (EPI & 1)) {
    const int mw = bm + wr * WTM;
    v_row_s = __builtin_amdgcn_readfirstlane((unsigned)(((mw < d.M ? mw : 0) / d.vdiv) % d.vmod) * (unsigned)(d.ldv * 4));
  } else if (EPI & 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int m = m0 + i * 32 + pass * 16;
        const int mc = m < d.M ? m : 0;
        const unsigned vi = d.vmode == 1 ? (unsigned)((mc / d.vdiv) % d.vmod)
                                         : (unsigned)((((long)(mc / d.vdiv) * d.vS + (mc % d.vS)) % d.vmod));
        v_row[i][pass] = vi * (unsigned)(d.ldv * 4);
      }
  }

  if constexpr (!GEGLU) {
    // Residual operands are streamed from HBM (~1 us): they are prefetched through a window that GROWS as the
    // epilogue retires accumulators -- every finished sub-tile frees 16 accumulator registers, enough for the R1 rows of
    // two more sub-tiles (one with R2) -- so a 10-sub-tile epilogue pays the memory latency about once, not ten times.
    struct Res { u32x4_t r1[2], r2[2]; u32x2_t r1l[LO ? 2 : 1], r2l[LO ? 2 : 1]; };
    constexpr bool HAS_RES = (EPI & 6) != 0;
    // sub-tiles in flight before the first one is processed: bounded by what the 320-wide tile (252+ VGPRs) can hold
    // (LO: every residual row is two planes -- one sub-tile ahead, one more per processed one)
    constexpr int P0 = (GNS || LO) ? 1 : (TN > 2 ? (EPI == 2 ? 2 : 1) : ((EPI & 4) ? 2 : 3));
    constexpr int GROW = ((EPI & 4) || GNS || LO) ? 1 : (TN > 2 ? 2 : 2);   // additional sub-tiles issued per processed one
    Res q[HAS_RES ? NSUB : 1];
    const int ocol0 = wbase_n + l4 * 8;
    // Addresses: ONE per-lane byte offset per operand (row m0, column ocol0) -- or out of range -- and the displacement
    // of sub-tile (i, j) / pass as the SCALAR offset of the buffer access (only the per-lane part is range-checked):
    // no per-(sub-tile, pass) lane offsets to compute or to keep in registers while all accumulators are still live.
    const unsigned r1_base = (unsigned)m0 * (unsigned)(d.ldr1 * 2) + (unsigned)(ocol0 * 2);
    const unsigned r2_base = (unsigned)m0 * (unsigned)(d.ldr2 * 2) + (unsigned)(ocol0 * 2);
    const unsigned o_base = (unsigned)m0 * (unsigned)(d.ldo * 2) + (unsigned)(ocol0 * 2);
    auto load_res = [&](int s) {
      const int i = GNS ? s % TM : s / TN, j = GNS ? s / TM : s % TN;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int dr = i * 32 + pass * 16;
        const bool ok = m0 + dr < d.M && ocol0 + j * 32 < d.n_store;
        if (EPI & 2)
          q[s].r1[pass] = __builtin_amdgcn_raw_buffer_load_b128(rsR1, ok ? r1_base : kOOB, (dr * d.ldr1 + j * 32) * 2, 0);
        if (LO && (EPI & 2))
          q[s].r1l[pass] = __builtin_amdgcn_raw_buffer_load_b64(rsR1L, ok ? (r1_base >> 1) : kOOB, dr * d.ldr1 + j * 32, 0);
        if (EPI & 4)
          q[s].r2[pass] = __builtin_amdgcn_raw_buffer_load_b128(rsR2, ok ? r2_base : kOOB, (dr * d.ldr2 + j * 32) * 2, 0);
        if (LO && (EPI & 4))
          q[s].r2l[pass] = __builtin_amdgcn_raw_buffer_load_b64(rsR2L, ok ? (r2_base >> 1) : kOOB, dr * d.ldr2 + j * 32, 0);
      }
    };
    if (HAS_RES) {
#pragma unroll
      for (int s = 0; s < P0 && s < NSUB; ++s) load_res(s);
    }
    // GNS: this column block's sums (pairs of channels: v_pk_add_f32 / v_pk_fma_f32 -- the epilogue's VALU instructions
    // issue at a fraction of their rate while the other wave group owns the matrix pipe); column sums per block
    // Sums are taken about a per-COLUMN pilot (round 6): the value of the wave tile's first row, broadcast from
    // lanes 0-3 (row_a == 0) to the lanes that hold the same eight columns -- eight lane exchanges per column block.  With
    // raw sums, sum x^2 - (sum x)^2 / n loses the variance in fp32 once |mean| >> std; gn_stats_kernel has always shifted.
    f32x2_t gcs[GNS ? 4 : 1], gcq[GNS ? 4 : 1], gpil[GNS ? 4 : 1];
    // (column sums of the finished blocks: in the wave's strip of LDS where the kernel has one -- gns_strip, [2][TN * 32]
    //  floats -- else in registers until the staging pieces are free)
    float gcol[GNS ? TN : 1];
    u32x4_t gvv[2] = {};                                             // GNS + row vector: this column block's 8 table entries
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const int i = GNS ? s % TM : s / TN, j = GNS ? s / TM : s % TN;
      // (GNS + LO: the window advances BEHIND this sub-tile's staging writes -- its 16 accumulators are dead by then and the
      //  next sub-tile's four residual rows take their registers; ahead of them the kernel went 25 registers over budget)
      constexpr bool LATE_WINDOW = GNS && LO;
      if (HAS_RES && !LATE_WINDOW) {
#pragma unroll
        for (int k = P0 + GROW * s; k < P0 + GROW * (s + 1); ++k)
          if (k < NSUB) load_res(k);
      }
      if (GNS && i == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { gcs[e] = f32x2_t{0.f, 0.f}; gcq[e] = f32x2_t{0.f, 0.f}; }
      }
      const int ocol = ocol0 + j * 32;
      // scale of this 32-column block (wave-uniform): the q block of a fused q|k|v projection takes s_acc2
      const float sc = (wbase_n + j * 32 < d.n_scale2) ? d.s_acc2 : d.s_acc;
      // row-vector operands (L2-resident tables): issued ahead of this sub-tile's LDS round trip
      u32x4_t vv[2][2] = {};
      if constexpr (GNS && (EPI & 1)) {
        // one table row for the whole wave tile: the 8 columns of this lane, loaded once per column block
        if (i == 0) {
          const bool okv = bm + wr * WTM < d.M && ocol < d.n_store;
          gvv[0] = __builtin_amdgcn_raw_buffer_load_b128(rsV, okv ? (unsigned)(ocol * 4) : kOOB, v_row_s, 0);
          gvv[1] = __builtin_amdgcn_raw_buffer_load_b128(rsV, okv ? (unsigned)(ocol * 4 + 16) : kOOB, v_row_s, 0);
        }
        vv[0][0] = vv[1][0] = gvv[0];
        vv[0][1] = vv[1][1] = gvv[1];
      } else if (EPI & 1) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const bool okv = m0 + i * 32 + pass * 16 < d.M && ocol < d.n_store;
          vv[pass][0] = __builtin_amdgcn_raw_buffer_load_b128(rsV, okv ? v_row[i][pass] + (unsigned)(ocol * 4) : kOOB, 0, 0);
          vv[pass][1] = __builtin_amdgcn_raw_buffer_load_b128(rsV, okv ? v_row[i][pass] + (unsigned)(ocol * 4 + 16) : kOOB, 0, 0);
        }
      }
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int c = (2 * qd + hsel) ^ (r32 & 7);
        stg_write16(wrow + c * 16, acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]);
      }
      __builtin_amdgcn_wave_barrier();     // compiler-only: the image is exchanged between lanes of this wave
      if (HAS_RES && LATE_WINDOW) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = P0 + GROW * s; k < P0 + GROW * (s + 1); ++k)
          if (k < NSUB) load_res(k);
        __builtin_amdgcn_sched_barrier(0);
      }
      float4 img[2][2];
      stg_read4x16(rp_a + rx0, rp_a + rx1, rp_b + rx0, rp_b + rx1, img);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const float4 v0 = img[pass][0], v1 = img[pass][1];
        float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const int m = m0 + i * 32 + pass * 16;
        const bool ok = m < d.M && ocol < d.n_store;
        // Fixed operation sequence -- t = acc * s_acc (rounded), then ONE fma per residual -- shared with the 2-stage
        // kernel (gemm_epilogue.h).  Left to the contraction heuristics, the SLP vectoriser paired fma(s_acc, acc, s1*R)
        // in one half of a v_pk_fma_f32 with fma(s1, R, s_acc*acc) in the other: legal, but a layer then differed in an
        // occasional last bit depending on which kernel (i.e. which batch size) served it.
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = o[e] * sc;              // (the bias is already in the accumulator)
        }
        if (EPI & 2) {
          float f[8];
          const u32x4_t r = q[HAS_RES ? s : 0].r1[pass];
          unpack_elx8(make_uint4(r.x, r.y, r.z, r.w), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(d.s1, f[e], o[e]);
          if (LO) {
            const u32x2_t rl = q[HAS_RES ? s : 0].r1l[pass];
            unpack_lo8(make_uint2(rl.x, rl.y), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(d.s1, f[e], o[e]);
          }
        }
        if (EPI & 4) {
          float f[8];
          const u32x4_t r = q[HAS_RES ? s : 0].r2[pass];
          unpack_elx8(make_uint4(r.x, r.y, r.z, r.w), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(d.s2, f[e], o[e]);
          if (LO) {
            const u32x2_t rl = q[HAS_RES ? s : 0].r2l[pass];
            unpack_lo8(make_uint2(rl.x, rl.y), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(d.s2, f[e], o[e]);
          }
        }
        if (EPI & 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] += __uint_as_float(vv[pass][0][e]);
            o[4 + e] += __uint_as_float(vv[pass][1][e]);
          }
        }
        if (GNS) {      // (rows past M: the whole wave tile is, M being a multiple of 64 -- its partials are not written)
          if (i == 0 && pass == 0) {
            const int src = (lane & 3) * 4;                      // byte index of the lane that holds row 0 of these columns
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c0 = (e >> 1) * 4 + (e & 1);
              gpil[e] = f32x2_t{__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o[c0]))),
                                __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o[c0 + 2])))};
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // channel pairs (0,2) (1,3) (4,6) (5,7): the pairing the compiler's own v_pk_mul / v_pk_add of the epilogue
            // arithmetic uses (x,z / y,w of the staged float4) -- any other costs two v_mov per pair
            const int c0 = (e >> 1) * 4 + (e & 1);
            const f32x2_t o2 = f32x2_t{o[c0], o[c0 + 2]} - gpil[e];
            gcs[e] += o2;
            gcq[e] += o2 * o2;
          }
        }
        const uint4 pk = pack_elx8(o);
        const u32x4_t pv = {pk.x, pk.y, pk.z, pk.w};
        pp_store_out(pv, rsO, ok ? o_base : kOOB, ((i * 32 + pass * 16) * d.ldo + j * 32) * 2);
        if (LO) {
          const uint2 pl = split_lo8(o, pk);
          const u32x2_t pvl = {pl.x, pl.y};
          pp_store_out_lo(pvl, rsOL, ok ? (o_base >> 1) : kOOB, (i * 32 + pass * 16) * d.ldo + j * 32);
        }
      }
      if (GNS && i == TM - 1) {
        // 64 lanes x 16 sums -> 64 column sums.  Value v (0..7: shifted sum of channel v, 8..15: its squares) of lane (row_a, l4)
        // goes to piece v >> 2 at float (v & 3) * 64 + l4 * 16 + row_a; lane L then adds the 16 consecutive floats of
        // (v = L >> 2, l4 = L & 3): gcol[j] = column 8 * (L & 3) + ((L >> 2) & 7) of block j, kind L >> 5.
        // (lane constants of this block are rebuilt per column block from an opaque copy of the lane id: kept live across
        //  the epilogue they push K-loop state into scratch -- and a scratch reload in the K loop is a vmcnt(0) there)
        int lj = lane;
        asm volatile("" : "+v"(lj));
        char* const pc[4] = {p0, p1, p2, p3};
        const int wofs = ((lj & 3) * 16 + (lj >> 2)) * 4;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int ch = v & 7, pi = (ch >> 2) * 2 + (ch & 1);
          const f32x2_t pr = v < 8 ? gcs[pi] : gcq[pi];
          *(float*)(pc[v >> 2] + (v & 3) * 256 + wofs) = ((ch >> 1) & 1) ? pr.y : pr.x;
        }
        __builtin_amdgcn_wave_barrier();
        const char* rsrc = ((lj >> 4) == 0 ? p0 : (lj >> 4) == 1 ? p1 : (lj >> 4) == 2 ? p2 : p3) +
                           ((lj >> 2) & 3) * 256 + (lj & 3) * 64;
        f32x2_t t2 = f32x2_t{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 v4 = *(const float4*)(rsrc + k * 16);
          t2 += f32x2_t{v4.x, v4.y} + f32x2_t{v4.z, v4.w};
        }
        // lane L < 32 holds a = sum (x - K) of its column, lane L + 32 b = sum (x - K)^2 of the same column: one half-wave
        // swap brings both to both; the column's pilot is this lane's own gpil entry of channel (L >> 2) & 7 (its l4 is the
        // column's).  The column's (mean, M2) over the 64 rows are well conditioned whatever the mean is.
        float colv;
        {
          const float t = t2.x + t2.y;
          const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
          const float oth = __uint_as_float((lj >> 5) ? sw2[0] : sw2[1]);
          const float a = (lj >> 5) ? oth : t, b = (lj >> 5) ? t : oth;
          const int ch = (lj >> 2) & 7;
          float kp = gpil[0].x;
#pragma unroll
          for (int c = 1; c < 8; ++c) {
            const int pi = (c >> 2) * 2 + (c & 1);
            const float pv_ = ((c >> 1) & 1) ? gpil[pi].y : gpil[pi].x;
            kp = ch == c ? pv_ : kp;
          }
          constexpr float inv_rows = 1.0f / 64.0f;               // (GNS: 64-row wave tiles)
          colv = (lj >> 5) ? b - a * a * inv_rows : kp + a * inv_rows;
        }
        if (gns_strip) *(float*)(gns_strip + ((lj >> 5) * (TN * 32) + j * 32 + 8 * (lj & 3) + ((lj >> 2) & 7)) * 4) = colv;
        else gcol[j] = colv;
        __builtin_amdgcn_wave_barrier();
      }
      // keep the machine scheduler from hoisting the later sub-tiles' loads up here: the prefetch window is sized to
      // the registers that are free at each point, hoisting turns it into hundreds of spills
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (GNS) {
      // the wave's WTN column sums and square sums through the strip [kind][WTN] (pieces p0 | p1 hold 2 * WTN floats only
      // if they are adjacent: they are not, so kind 0 lives in p0 and kind 1 in p1), then one lane per group
      static_assert(TN * 32 * 4 <= 1024, "GNS: a kind's column sums must fit one staging piece");
      const int kind = lane >> 5, cin = 8 * (lane & 3) + ((lane >> 2) & 7);
      if (!gns_strip) {
        char* const strip = kind ? p1 : p0;
#pragma unroll
        for (int j = 0; j < TN; ++j) *(float*)(strip + (j * 32 + cin) * 4) = gcol[j];
      }
      const char* const ks = gns_strip ? gns_strip : p0;
      const char* const kq = gns_strip ? gns_strip + TN * 32 * 4 : p1;
      __builtin_amdgcn_wave_barrier();
      const int cpg = d.N >> 5, ng = (TN * 32) / cpg;          // channels per group; groups of this wave's columns
      const int mw = bm + wr * WTM;                            // first row of the wave tile
      if (lane < ng && mw < d.M) {
        // the group's columns, each (mean_c, M2_c) over WTM rows, merged with Chan's update (gn_stats_kernel's last stage)
        float mean = 0.f, m2 = 0.f;
        for (int c = 0; c < cpg; ++c) {
          const float mc = *(const float*)(ks + (lane * cpg + c) * 4), qc = *(const float*)(kq + (lane * cpg + c) * 4);
          const float dlt = mc - mean, rk = 1.0f / (float)(c + 1);
          m2 += qc + dlt * dlt * ((float)WTM * (float)c * rk);
          mean += dlt * rk;
        }
        const int g = wbase_n / cpg + lane;
        *(float2*)(d.gn_partials + (((long)(mw / WTM) * 32 + g) * 2)) = make_float2(mean, m2);
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    if constexpr (RAW) {
      // training forward: the projection itself (acc already holds acc + bias) also goes out, as bf16 [M][ld_raw] in the
      // packed column order -- the plain path's LDS transpose, one 32 x 32 sub-tile at a time.  The backward then needs
      // no recompute GEMM (ctrlv_geglu_bwd reads this tensor).
      const __amdgpu_buffer_rsrc_t rsRaw =
          __builtin_amdgcn_make_buffer_rsrc(d.raw_out, 0, (int)((long)d.M * d.ld_raw * 2), kFlags);
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        const int i = s / TN, j = s % TN;
        const int ocol = wbase_n + j * 32 + l4 * 8;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int c = (2 * qd + hsel) ^ (r32 & 7);
          stg_write16(wrow + c * 16, acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]);
        }
        __builtin_amdgcn_wave_barrier();
        float4 img[2][2];
        stg_read4x16(rp_a + rx0, rp_a + rx1, rp_b + rx0, rp_b + rx1, img);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const float4 v0 = img[pass][0], v1 = img[pass][1];
          const float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          const int m = m0 + i * 32 + pass * 16;
          const uint4 pk = pack_elx8(o);
          const u32x4_t pv = {pk.x, pk.y, pk.z, pk.w};
          __builtin_amdgcn_raw_buffer_store_b128(
              pv, rsRaw, (m < d.M && ocol < d.N) ? (unsigned)m * (unsigned)(d.ld_raw * 2) + (unsigned)(ocol * 2) : kOOB, 0, 0);
          store_data_hazard_guard(pv);
        }
      }
    }
    // GEGLU: weight rows come in 16-row (value, gate) blocks, so quads 0,1 of a 32x32 sub-tile are 16 values and
    // quads 2,3 their gates, in the same lane: out = (a + ba) * gelu(g + bg) is computed in the MFMA layout (its bias
    // is a per-column broadcast from the LDS strip) and two adjacent sub-tiles (16 outputs each) share one staged
    // 32-column image; a lone last sub-tile fills only the left half.
    // The table reads of block b + 1 (16 gates: one staged image) are issued before block b's image goes through the
    // staging round trip, so neither latency is paid per group of four gates (the staging writes and the table reads are
    // both LDS accesses: the compiler keeps them in source order, so the order below is the order that runs).
    constexpr int NP = (TN + 1) / 2, NB = TM * NP;
    float fr[16];
    float2 tb[16];
    auto lookup = [&](int b) {
      const int i = b / NP, j = (b % NP) * 2;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int js = j + half;
        if (js >= TN) continue;                            // compile time
#pragma unroll
        for (int qd = 0; qd < 2; ++qd)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            gelu_tab_lookup(acc[i][js][4 * (qd + 2) + e], gelu_tab, fr[half * 8 + qd * 4 + e], tb[half * 8 + qd * 4 + e]);
      }
    };
    lookup(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int i = b / NP, jp = b % NP, j = jp * 2;
      float o[16];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int js = j + half;
        if (js >= TN) continue;                            // compile time
#pragma unroll
        for (int qd = 0; qd < 2; ++qd)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int k = half * 8 + qd * 4 + e;           // a * gelu(g), Phi from the LDS table (common.h)
            o[k] = geglu_tab_finish(acc[i][js][4 * qd + e], acc[i][js][4 * (qd + 2) + e], fr[k], tb[k]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (b + 1 < NB) lookup(b + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (j + half >= TN) continue;
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
          const int c = (half * 4 + 2 * qd + hsel) ^ (r32 & 7), k = half * 8 + qd * 4;
          stg_write16(wrow + c * 16, o[k], o[k + 1], o[k + 2], o[k + 3]);
        }
      }
      const int ocol = ((wbase_n + j * 32) >> 1) + l4 * 8;
      // columns of the right half exist only if sub-tile j+1 does (inside the wave tile and inside N)
      const bool col_ok = ocol < d.n_store && (l4 < 2 || (j + 1 < TN && wbase_n + (j + 1) * 32 < d.N));
      __builtin_amdgcn_wave_barrier();   // compiler-only: the image is exchanged between lanes of this wave
      float4 img[2][2];
      stg_read4x16(rp_a + rx0, rp_a + rx1, rp_b + rx0, rp_b + rx1, img);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const float4 v0 = img[pass][0], v1 = img[pass][1];
        const float ov[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const int m = m0 + i * 32 + pass * 16;
        const uint4 pk = pack_elx8(ov);
        const u32x4_t pv = {pk.x, pk.y, pk.z, pk.w};
        pp_store_out(pv, rsO, (m < d.M && col_ok) ? (unsigned)m * (unsigned)(d.ldo * 2) + (unsigned)(ocol * 2) : kOOB, 0);
      }
    }
  }
}

// HAS_A2: the launch has a second A source for channels >= c_split (skip concat).  Only the plain-GEMM / bias-only
// combination exists (the 1x1 shortcut convs of the up blocks; every other consumer of a concat reads the GroupNorm
// output), so all other instantiations carry no source-select instructions in their hot loop (~15 of ~95 per half-step).
template <int BN, int WM, int WN, int MODE, bool GEGLU, int EPI, bool HAS_A2 = false, bool RAW = false, bool HALO = false,
          bool GNS = false, bool LO = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const ctrlv_gemm_desc d, const int cgrp) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses device-only types (__amdgpu_buffer_rsrc_t): keep it out of the host pass
  constexpr int BM = 256, NW = 8, NH = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_SLOT = BM * 64, B_SLOT = BN * 64, SLOT = A_SLOT + B_SLOT;
  constexpr int A_TOT = BM / 16, B_TOT = BN / 16;            // 1-KiB DMA pieces (16 rows x 64 B) per half-step
  static_assert(!HALO || (MODE == 1 && !HAS_A2 && !GEGLU), "row-halo staging: 3x3 gather only");
  // HALO: one A piece per wave and half-step (piece POS * 8 + wid of the row-halo slot of the NEXT (dy, channel block)
  // triple, POS = half-step mod 3); pieces beyond the slot's real ones go to the dummy KiB
  constexpr int A_Q = HALO ? 1 : A_TOT / NW;                 // per wave (2)
  constexpr int B_Q = (B_TOT + NW - 1) / NW;                 // per wave (2 or 3; see the dummy piece below)
  static_assert(WM * WN == NW && A_TOT % NW == 0, "bad wave layout");
  // A wave's pieces of one half-step, in issue order: A_0 .. A_{A_Q-1}, B_0 .. B_{B_Q-1}.  The first NL are issued in
  // the LOAD phase, the others in the four gaps of the MFMA cluster (where an LDS-DMA issue overlaps the matrix pipe).
  // Whichever phase is longer sets the slot time, so NL balances them: A/B on one device (tools/ab_build.py
  // -DCTRLV_PP_NL=n): 3 of 4 for the 256-wide tile (16 MFMAs per phase); for the 320-wide one (20 MFMAs) 2 of 5 won the
  // single-kernel sweeps of round 1, but inside the model 4 of 5 is the faster one (234.5 -> 233.2 ms per step, three
  // alternations on one device; 1 of 5: 243 ms): repeated launches of one kernel flatter whatever leans on the caches.
  constexpr int NPIECE = A_Q + B_Q;
  constexpr int NL = HALO ? NPIECE - 1 : (BN == 256 ? A_Q + 1 : (BN == 320 ? A_Q + 2 : A_Q));
  constexpr int NC = NPIECE - NL;                            // pieces issued in the compute phase
  static_assert(NL >= 0 && NL < NPIECE, "piece placement out of range");
  // When B_TOT is not a multiple of the wave count (320-wide: 20 pieces, 8 waves) the waves without a real last piece
  // issue a DUMMY one -- out-of-range source (zeros, no memory traffic) into a private 1-KiB scratch strip -- so that
  // every wave has exactly NPIECE loads per half-step and the counted vmcnt waits need no per-wave cases.
  constexpr bool UNEVEN = (B_TOT % NW) != 0;
  // HALO: a ring of NA row-halo slots of up to 24 pieces (A_HSLOT) in front of the 4-slot ring of weight tiles
  constexpr int NA = 3, A_HSLOT = 24 * 1024, A_RING = HALO ? NA * A_HSLOT : 0;
  // (one dummy KiB serves all waves: it is only ever written, with zeros)
  constexpr int BIAS_OFF = HALO ? A_RING + NH * B_SLOT : NH * SLOT, DUMMY_OFF = BIAS_OFF + NW * WTN * 4;
  // The epilogue stages through FOUR wave-private 1-KiB pieces: the wave's two A pieces and two B pieces of the slot
  // consumed last.  A 128-wide tile has only one B piece per wave: its fourth piece is a private strip behind the ring.
  constexpr bool OWN_P3 = B_TOT < 2 * NW;
  constexpr int P3_OFF = DUMMY_OFF + ((UNEVEN || HALO) ? 1024 : 0);
  constexpr int TAB_OFF = P3_OFF + (OWN_P3 ? NW * 1024 : 0);      // Phi table of the GEGLU epilogue (common.h)

  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 4 ring slots | 8 bias strips | dummy piece | P3 | Phi
  CTRLV_CLOCK_BEGIN();

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 2;                                  // waves w and w+4 share a SIMD -> different groups
  const int wr = wid / WN, wc = wid % WN;
  const int r32 = lane & 31, hsel = lane >> 5;

  // KS (EPI 8: split contraction, ctrlv_gemm's K slices in ONE launch): the launch runs d.ksplit copies of the tile grid,
  // copy s contracting over channels [s * Cin, (s + 1) * Cin) of every tap (d.Cin = channels per slice, d.w_cin = channels
  // per tap of W) and storing its raw fp32 accumulators to out + s * M * ldo floats.  The slices are extra ROW BLOCKS of
  // the tile order (row block mt -> slice mt / tiles_m_real): only the per-tile lane offsets know about them.
  constexpr bool KS = EPI == 8;
  const int tiles_n = (d.N + BN - 1) / BN;
  const int tiles_m_real = (d.M + BM - 1) / BM;
  const int tiles_m = KS ? tiles_m_real * d.ksplit : tiles_m_real;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  // Tile order.  Tile numbers run over COLUMN GROUPS of `cgrp` column tiles: all row blocks of group 0 (row-major inside the
  // group), then group 1, ...  The 32 CUs of an XCD work on 32 consecutive numbers, i.e. on (32 / cgrp) row blocks x cgrp
  // column tiles: cgrp weight tiles stay in that XCD's 4 MB L2 for the whole sweep over the rows while every activation
  // tile is fetched once per group.  cgrp = tiles_n is the plain row-major order (right when the whole weight matrix fits
  // L2: C = 320); launch_one() picks it from the weight tile's size.  Before: the C = 1280 GEGLU projection (26 MB of
  // weights, 40 column tiles) re-read all of them for each of its 113 row blocks -- 3.2 GB per launch for 100 MB of operands.
  const int grp_full = tiles_n / cgrp, grp_tiles = tiles_m * cgrp;
  auto tile_mn = [&](int t, int& mt, int& nt) {
    const int gi = t / grp_tiles;
    if (gi < grp_full) {
      const int rem = t - gi * grp_tiles;
      mt = rem / cgrp; nt = gi * cgrp + (rem - mt * cgrp);
    } else {
      // (also reached by the issue stream's look-ahead past the last tile: any row block >= tiles_m reads zeros)
      const int rem = t - grp_full * grp_tiles, w = tiles_n - grp_full * cgrp;
      if (w > 0) { mt = rem / w; nt = grp_full * cgrp + (rem - mt * w); }
      else { mt = tiles_m; nt = 0; }
    }
  };
  // tile of round r for this block: XCD-contiguous inside every window of G tiles
  const int my_first = xcd_remap(blockIdx.x, G);
  const int my_ntiles = (ntiles - my_first + G - 1) / G;       // >= 1 (grid <= ntiles)
  const int J = d.taps * (d.Cin >> 5);                       // half-steps per tile (>= 4: ctrlv_gemm_pp_supports)
  const int wcin = KS ? d.w_cin : d.Cin;                     // channels per tap in W
  const long ktot = (long)d.taps * wcin;

  // ---- DMA addressing.  Sources go through buffer descriptors (buffer_load ... lds): address = base + voffset (per
  // lane) + soffset (scalar), and only the per-lane part is range-checked.  So a lane's offset is computed ONCE per
  // tile -- the byte offset of its row (tap centre), or 0xFFFFFFFF for rows past M / tile overhang, which the hardware
  // turns into zeros -- and a half-step only supplies a scalar: the channel offset plus, for the 3x3 / temporal gathers,
  // the tap displacement.  Tap displacements can be negative, so the descriptor base is moved DOWN by the largest one
  // (Wd+1 rows / S rows) and every scalar offset is >= 0.  The hot loop carries no per-lane address arithmetic beyond a
  // tap-validity select: a single wave issues only one VALU instruction per ~10 cycles while its SIMD partner owns the
  // matrix pipe, so every bookkeeping instruction there is on the critical path (SQ counters: LDS and TA are < 20 %
  // busy, MFMA 40 % -- the loop was issue-bound, not bandwidth-bound).
  const int prow = lane >> 2, pslot = lane & 3;
  const unsigned coff = (pslot ^ ((prow >> 2) & 3)) * 16;    // logical chunk this lane fetches (bytes); the piece
                                                             // base row is a multiple of 16, so (row>>2)&3 == (prow>>2)&3
  const unsigned kOOB = 0xFFFFFFFFu;
  const long a_rows = MODE == 1 ? (long)(d.M / (d.Ho * d.Wo)) * d.H * d.Wd : (long)d.M;
  const int bias_rows = MODE == 1 ? d.Wd + 1 : (MODE == 2 ? d.S : 0);
  const long bias_a = (long)bias_rows * d.lda * 2, bias_a2 = (long)bias_rows * d.lda2 * 2;
  // (num_records covers the bias too: the upsample gather adds its per-lane displacement, bias included, to voffset)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)d.A - bias_a), 0, (int)(a_rows * d.lda * 2 + bias_a), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)(d.A2 ? d.A2 : d.A) - bias_a2), 0, (int)(d.A2 ? a_rows * d.lda2 * 2 + bias_a2 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.W, 0, (int)((long)d.N * ktot * 2), 0x00020000);
  unsigned a_voff[A_Q], a_voff2[A_Q];                        // per-lane row offsets into A / A2 (kOOB if the row is invalid)
  unsigned a_mask[A_Q];                                      // modes 1/2, bits 0..8: tap validity; bits 16,17: y/x parity (upsample)
  unsigned b_voff[B_Q];                                      // byte offset of the weight row (+chunk), kOOB if out of range
  // HALO: the row-halo slot of a tile holds R = 256 / W image rows of W + 2 pixels (64 B each): slot pixel sp = jr * (W + 2)
  // + x + 1.  A wave fills pieces POS * 8 + wid (POS = 0..2, 16 slot pixels each): per lane and POS the byte offset of
  // its pixel's CENTRE row (dy = 0; a (dy, channel block) adds a scalar) -- out of range for the zero columns, for rows
  // past M and for pieces past the slot -- and its three dy-validity bits (image top / bottom).
  unsigned ah_voff[HALO ? 3 : 1], ah_mask[HALO ? 3 : 1];
  const int hw2 = d.Wd + 2, npa = HALO ? (256 + 2 * (256 / (d.Wd > 0 ? d.Wd : 1)) + 15) / 16 : 0;
  auto setup_a_halo = [&](int tile) {
    int mt_, nt_;
    tile_mn(tile, mt_, nt_);
    const int bm = mt_ * BM;
#pragma unroll
    for (int pos = 0; pos < 3; ++pos) {
      const int piece = pos * NW + wid;
      const int sp = piece * 16 + prow;
      const int jr = sp / hw2, x = sp - jr * hw2 - 1;
      const int m = bm + jr * d.Wd + x;
      const bool ok = piece < npa && jr * d.Wd < BM && (unsigned)x < (unsigned)d.Wd && m < d.M;
      const int y = ok ? (m / d.Wd) % d.H : 0;
      ah_voff[pos] = ok ? (unsigned)m * (unsigned)(d.lda * 2) + coff : kOOB;
      ah_mask[pos] = ok ? ((y > 0 ? 1u : 0u) | 2u | (y < d.H - 1 ? 4u : 0u)) : 0u;
    }
  };
  auto setup = [&](int tile) {
    int mt_, nt_;
    tile_mn(tile, mt_, nt_);
    const int ksl = KS ? mt_ / tiles_m_real : 0;               // K slice (>= ksplit past the last tile: all rows invalid)
    const int bm = (KS ? mt_ - ksl * tiles_m_real : mt_) * BM, bn = nt_ * BN;
    const unsigned ks_off = KS ? (unsigned)(ksl * d.Cin * 2) : 0u;   // byte offset of the slice's first channel
    (void)bm;
#pragma unroll
    for (int q = 0; q < (HALO ? 0 : A_Q); ++q) {
      const int m = bm + (q * NW + wid) * 16 + prow;
      const bool ok = m < d.M && (!KS || ksl < d.ksplit);
      int row = m;
      unsigned mask = 0;
      if (MODE == 1) {
        const int hw = d.Ho * d.Wo;
        const int n_img = m / hw, rem = m - n_img * hw;
        const int yo = rem / d.Wo, xo = rem - yo * d.Wo;
        int cy, cx;
        if (d.up) { cy = yo >> 1; cx = xo >> 1; mask = ((unsigned)(yo & 1) << 16) | ((unsigned)(xo & 1) << 17); }
        else { cy = yo * d.stride; cx = xo * d.stride; }
        row = (n_img * d.H + cy) * d.Wd + cx;
        const int hl = d.H << d.up, wl = d.Wd << d.up;
        const int y0 = d.up ? yo : cy, x0 = d.up ? xo : cx;   // coordinates in the (upsampled) conv input grid
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int yi = y0 + t / 3 - 1, xi = x0 + t % 3 - 1;
          if (ok && (unsigned)yi < (unsigned)hl && (unsigned)xi < (unsigned)wl) mask |= 1u << t;
        }
      } else if (MODE == 2) {
        const int f = (m / d.S) % d.F;
        mask = ok ? ((f > 0 ? 1u : 0u) | 2u | (f < d.F - 1 ? 4u : 0u)) : 0u;
      }
      a_voff[q] = ok ? (unsigned)row * (unsigned)(d.lda * 2) + coff + ks_off : kOOB;
      a_voff2[q] = ok ? (unsigned)row * (unsigned)(d.lda2 * 2) + coff : kOOB;
      a_mask[q] = mask;
    }
#pragma unroll
    for (int q = 0; q < B_Q; ++q) {
      const int ib = q * NW + wid;
      const int n = bn + ib * 16 + prow;
      b_voff[q] = (ib < B_TOT && n < d.N) ? (unsigned)n * (unsigned)(ktot * 2) + coff + ks_off : kOOB;
    }
  };

  // issue-stream state (scalars): tile-local half-step about to be issued = (is_tap, is_cc); the tile itself is switched
  // by the K loop below (three half-steps before the consuming stream gets there)
  int is_tap = 0, is_cc = 0;
  // HALO: the weight stream walks (dy, channel block, dx); the row-halo stream is ONE TRIPLE AHEAD of it: the unit issued
  // for half-step h carries the weight pieces of h and piece h % 3 of the slot of the triple AFTER h's (so a slot is
  // complete, and retired by the counted waits, a whole triple before its first read).  ia_* = the triple being filled.
  int is_dy = -1, is_dx = -1;
  int ia_dy = -1, ia_cc = 0, ia_t = 0;
  char* is_sa = nullptr;
  char* is_sb = nullptr;
  bool is_second = false;
  unsigned is_so_a = 0, is_so_w = 0;
  int is_dyo = 0, is_dxo = 0, is_ld2 = 0;
  auto issue_begin = [&](int g) {
    if constexpr (HALO) {
      is_sa = smem + (ia_t % NA) * A_HSLOT;
      is_sb = smem + A_RING + (g & (NH - 1)) * B_SLOT;
      is_ld2 = d.lda * 2;
      is_so_a = __builtin_amdgcn_readfirstlane((unsigned)(((ia_dy + 1) * d.Wd + 1) * is_ld2) + (unsigned)(ia_cc * 2));
      is_so_w = __builtin_amdgcn_readfirstlane((unsigned)(((((is_dy + 1) * 3 + is_dx + 1) * d.Cin) + is_cc) * 2));
      // pin the two offsets in SGPRs HERE: left alone, the compiler carries the pre-readfirstlane VGPR values to the issue
      // points, and under the register pressure of the GNS epilogues one of them went to scratch -- reloaded in the K loop
      // behind an s_waitcnt vmcnt(0), which drains the LDS-DMA pipeline every half-step
      asm volatile("" : "+s"(is_so_a), "+s"(is_so_w));
      return;
    }
    is_sa = smem + (g & (NH - 1)) * SLOT;
    is_sb = is_sa + A_SLOT;
    is_second = HAS_A2 && is_cc >= d.c_split;
    is_ld2 = (is_second ? d.lda2 : d.lda) * 2;               // row pitch in bytes of the active source
    int roff = 0;                                            // tap displacement in rows, biased to be >= 0
    if (MODE == 1) {
      is_dyo = is_tap / 3 - 1; is_dxo = is_tap % 3 - 1;
      roff = d.up ? 0 : (is_dyo + 1) * d.Wd + is_dxo + 1;    // upsample: per-lane (parity), see issue_a
    } else if (MODE == 2) {
      roff = is_tap * d.S;
    }
    // (readfirstlane: these are wave-uniform by construction; it keeps the compiler from wrapping the loads in a
    // waterfall loop when it has routed the arithmetic through vector registers)
    is_so_a = __builtin_amdgcn_readfirstlane((unsigned)(roff * is_ld2) + (unsigned)((is_second ? is_cc - d.c_split : is_cc) * 2));
    is_so_w = __builtin_amdgcn_readfirstlane((unsigned)(((MODE == 0 ? 0 : is_tap * wcin) + is_cc) * 2));
  };
  auto issue_a = [&](int q) {
    unsigned voff = (HAS_A2 && is_second) ? a_voff2[q] : a_voff[q];
    if (MODE != 0) {
      if (MODE == 1 && d.up) {   // nearest x2: source = ((yo + dy - 1) >> 1, (xo + dx - 1) >> 1), relative to the centre
        const int oy = ((int)((a_mask[q] >> 16) & 1) + is_dyo) >> 1, ox = ((int)((a_mask[q] >> 17) & 1) + is_dxo) >> 1;
        voff += (unsigned)((oy * d.Wd + ox + d.Wd + 1) * is_ld2);
      }
      if (!((a_mask[q] >> is_tap) & 1u)) voff = kOOB;
    }
    if (HAS_A2 && is_second)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, LDS_PTR(is_sa + (q * NW + wid) * 1024), 16, voff, is_so_a, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(is_sa + (q * NW + wid) * 1024), 16, voff, is_so_a, 0, 0);
  };
  auto issue_b = [&](int q) {
    if (!UNEVEN || q < B_Q - 1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(is_sb + (q * NW + wid) * 1024), 16, b_voff[q], is_so_w, 0, 0);
    } else {
      char* dst = (q * NW + wid < B_TOT) ? is_sb + (q * NW + wid) * 1024 : smem + DUMMY_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dst), 16, b_voff[q], is_so_w, 0, 0);
    }
  };
  // HALO: piece `pos` (= half-step mod 3) of the slot being filled
  auto issue_a_halo = [&](int pos) {
    unsigned voff = pos == 0 ? ah_voff[0] : (pos == 1 ? ah_voff[HALO ? 1 : 0] : ah_voff[HALO ? 2 : 0]);
    const unsigned msk = pos == 0 ? ah_mask[0] : (pos == 1 ? ah_mask[HALO ? 1 : 0] : ah_mask[HALO ? 2 : 0]);
    if (!((msk >> (ia_dy + 1)) & 1u)) voff = kOOB;
    const int piece = pos * NW + wid;
    char* dst = piece < npa ? is_sa + piece * 1024 : smem + DUMMY_OFF;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst), 16, voff, is_so_a, 0, 0);
  };
  int is_pos = 0;                                            // HALO: position of the unit being issued inside its triple
  auto issue_piece = [&](int pc) {
    if constexpr (HALO) {
      if (pc < A_Q) issue_a_halo(is_pos); else issue_b(pc - A_Q);
    } else {
      if (pc < A_Q) issue_a(pc); else issue_b(pc - A_Q);
    }
  };
  // K traversal of the 3x3 / temporal gathers: tap outermost, the channels inside it (the 2-stage kernel's order too --
  // gemm.hip -- so the bits do not depend on the kernel).  Every tap then re-reads its A rows from the fabric (the re-use
  // distance is a whole sweep over the channels, more than an XCD's L2 holds: 2.5 GB of L2 <-> fabric reads per launch for
  // the 295 MB input of the 320 -> 320 conv of the 72 x 128 level, 8.3 GB for its 960 -> 320 one; TCC_EA0_RDREQ,
  // tools/pmc_fetch.sh).  The alternative -- 64-channel block outermost, taps inside, -DCTRLV_CONV_BLOCK_MAJOR -- fetches
  // every line once (0.47 / 1.3 GB, L2 hit rate 0.77 -> 0.95) and is 7-9 % faster on the C = 640 / 1280 convs when the same
  // launch is repeated, but 1.5-2.5 % SLOWER on every conv shape inside the model (tools/shape_table.py, both orders in
  // one session; 63.6 vs 64.8 ms of conv kernels per step): the reads it saves come out of the Infinity Cache, which is
  // not what these kernels wait for, and its weight reads jump by Cin between consecutive half-steps.  Not the default.
  auto issue_end = [&]() {
    if constexpr (HALO) {
      if (++is_dx > 1) {                                     // weights: (dy, channel block, dx)
        is_dx = -1;
        is_cc += 32;
        if (is_cc == d.Cin) { is_cc = 0; ++is_dy; }
      }
      if (++is_pos == 3) {                                   // row-halo slot complete: next (dy, channel block)
        is_pos = 0;
        ++ia_t;
        ia_cc += 32;
        if (ia_cc == d.Cin) { ia_cc = 0; ++ia_dy; }
      }
      return;
    }
    is_cc += 32;
    if (MODE != 0 && is_cc == d.Cin) { is_cc = 0; ++is_tap; }   // (plain GEMM: one tap, next_tile() rewinds)
  };
  auto issue = [&](int g) {                                  // whole half-step at once (prologue)
    issue_begin(g);
#pragma unroll
    for (int pc = 0; pc < NPIECE; ++pc) issue_piece(pc);
    issue_end();
  };
  auto next_tile = [&](int tile) {                           // the issue stream moves to `tile` (may be past the last
    setup(tile);                                             // one: its rows are all invalid, the pieces read zeros)
    is_tap = 0;
    is_cc = 0;
    is_dy = -1; is_dx = -1;
  };
  auto next_tile_a = [&](int tile) {                         // HALO: the row-halo stream moves to `tile` (one triple early)
    setup_a_halo(tile);
    ia_dy = -1; ia_cc = 0;
  };

  const int sw = (r32 >> 2) & 3;
  const int a_frag = (wr * WTM + r32) * 64;
  const int b_frag = (HALO ? 0 : A_SLOT) + (wc * WTN + r32) * 64;
  // HALO: byte offset of this lane's pixel (row block i) inside a row-halo slot for dx = -1 / 0 / +1, with the chunk
  // swizzle of THAT slot row folded in for the first k-slice (the second one is the same address ^ 32)
  unsigned ah_frag[HALO ? 3 : 1][HALO ? TM : 1];
  if constexpr (HALO) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wr * WTM + i * 32 + r32;
      const int jr = r / d.Wd, x = r - jr * d.Wd;
#pragma unroll
      for (int pos = 0; pos < 3; ++pos) {
        const int lr = jr * hw2 + x + pos;
        ah_frag[pos][i] = (unsigned)(lr * 64 + ((hsel ^ ((lr >> 2) & 3)) * 16));
      }
    }
  }

  // bias strip of this wave for the first tile (see pp_bias_load); its load is retired before any DMA is issued
  char* const bias_lds = smem + BIAS_OFF + wid * (WTN * 4);
  {
    int mt0, nt0;
    tile_mn(my_first, mt0, nt0);
    const int bn0 = nt0 * BN;
    const u32x4_t b = pp_bias_load<WTN>(d, bn0 + wc * WTN, lane, KS && mt0 >= tiles_m_real);
    pp_bias_store<WTN>(bias_lds, b, lane);
  }
  // The first MFMAs of a tile start from the BIAS instead of zero: in the result layout a lane's 16 accumulators of a
  // 32-column sub-tile are columns 8q + 4 hsel + r (q, r = 0..3), so the C operand is four ds_read_b128 from the strip.
  // The epilogue then has no bias reads or adds at all (it was 12-17 % of its instructions, and the epilogue is what
  // limits the K = 320 layers).
  auto bias_c = [&](int n) {
    f32x16 c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *(const float4*)(bias_lds + (n * 32 + 8 * q + 4 * hsel) * 4);
      c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
    }
    return c;
  };
  // Phi table: written here, read in the first epilogue -- every wave passes an `lgkmcnt(0)` + barrier of the K loop between
  if constexpr (GEGLU) gelu_table_fill(smem + TAB_OFF, threadIdx.x, NW * 64);
  // ---- prologue: 3 half-steps in flight
  int is_tile = my_first;
  next_tile(is_tile);
  if constexpr (HALO) {
    // slot of the first triple: all three pieces at once (the loop fills every later slot one piece per half-step)
    next_tile_a(is_tile);
    issue_begin(0);
#pragma unroll
    for (int pos = 0; pos < 3; ++pos) issue_a_halo(pos);
    ++ia_t;
    ia_cc += 32;
    if (ia_cc == d.Cin) { ia_cc = 0; ++ia_dy; }
  }
  issue(0);
  issue(1);
  issue(2);
  wait_vmcnt<2 * NPIECE>();
  raw_barrier();

  f32x16 acc[TM][TN];
  int g = 0;
  constexpr int HM = TM / 2;
  // buffer stores a wave issues in one epilogue (straight-line code: out-of-range ones are issued and counted too)
  constexpr int NSTORE = EPI == 8 ? TM * TN * 4 : (GEGLU ? ((TN + 1) / 2) * TM * 2 + (RAW ? TM * TN * 2 : 0) : TM * TN * (LO ? 4 : 2));
  static_assert(!LO || (!GEGLU && !RAW && EPI != 8), "split planes: plain epilogues only (with or without GroupNorm partials)");
  static_assert(NPIECE + NPIECE + NSTORE <= 63, "vmcnt is a 6-bit counter");
  bool after_epi = false;                                    // this workgroup has run an epilogue (tile > first)
#ifdef CTRLV_PP_STAMP
  unsigned long long c_lread = 0, c_lissue = 0, c_lwait = 0, c_lbar = 0, c_mfma = 0, c_cbar = 0, c_epi = 0;
  STAMP(t_begin);
#endif
  // ================= PING-PONG schedule (see the header of this file)
  if (grp == 1) raw_barrier();                               // stagger: group 1 runs one slot behind
  auto half_step = [&](int j, bool /*last_of_tile*/, auto first_tag, auto pos_tag) {
    constexpr bool MAY_BE_FIRST = decltype(first_tag)::value;
    constexpr int POS = decltype(pos_tag)::value;            // HALO: dx + 1 of this half-step (its position in the triple)
    // ---------------- L phase: fragments of half-step g -> registers; DMA for g+3; retire own DMA(g+1)
    const char* st = HALO ? smem + A_RING + (g & (NH - 1)) * B_SLOT : smem + (g & (NH - 1)) * SLOT;
    elx8 af[TM][2], wf[TN][2];
    STAMP(t0);
    if constexpr (HALO) {
      const char* sa = smem + ((g / 3) % NA) * A_HSLOT;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff_ = ((ks * 2 + hsel) ^ sw) * 16;
#pragma unroll
        for (int n = 0; n < TN; ++n) wf[n][ks] = *(const elx8*)(st + b_frag + n * 32 * 64 + coff_);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][ks] = *(const elx8*)(sa + (ah_frag[POS][i] ^ (ks ? 32u : 0u)));
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff_ = ((ks * 2 + hsel) ^ sw) * 16;
#pragma unroll
        for (int n = 0; n < TN; ++n) wf[n][ks] = *(const elx8*)(st + b_frag + n * 32 * 64 + coff_);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][ks] = *(const elx8*)(st + a_frag + i * 32 * 64 + coff_);
      }
    }
    STAMP(t1);
    // DMA of half-step g+3: the first NL pieces are issued here (load phase), the rest in the gaps of the MFMA
    // cluster below.  Then retire own DMA(g+1): the pieces of g+2 and the NL pieces just issued may stay in flight.
    issue_begin(g + 3);
#pragma unroll
    for (int pc = 0; pc < NL; ++pc) issue_piece(pc);
    STAMP(t1b);
    // `vmcnt` counts loads AND stores, in order.  The pieces retired here in the first two half-steps after an epilogue
    // (half-steps g+1 of the new tile) were issued BEFORE that epilogue: its NSTORE stores are newer and may stay in
    // flight -- with the steady-state count the wave would sit here until the tile's output has been acknowledged by
    // HBM (stamps: 300-550 cycles per half-step on the K = 320 layers, averaged over the tile's ten).
    if (MAY_BE_FIRST && after_epi && j < 2) wait_vmcnt<NPIECE + NL + NSTORE>();
    else wait_vmcnt<NPIECE + NL>();
    STAMP(t2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(t2b);
    lds_done_barrier();
    STAMP(t3);
#if defined(CTRLV_PP_STAMP) && CTRLV_PP_STAMP == 2     // by position in the tile: whole L phase of half-step 0 / 1 / 2 / >= 3
    if (j == 0) { STAMP_ADD(c_lread, t0, t3); } else if (j == 1) { STAMP_ADD(c_lissue, t0, t3); }
    else if (j == 2) { STAMP_ADD(c_lwait, t0, t3); } else { STAMP_ADD(c_lbar, t0, t3); }
#else
    STAMP_ADD(c_lread, t0, t1);
    STAMP_ADD(c_lissue, t1, t1b);
    STAMP_ADD(c_lwait, t1b, t2b);
    STAMP_ADD(c_lbar, t2b, t3);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- C phase: 4 MFMA groups from registers with the DMA pieces of half-step g+3 in the gaps
    // (first half-step of a tile starts from a literal-zero C operand instead of zeroing 128-160 registers)
#pragma unroll
    for (int grpi = 0; grpi < 4; ++grpi) {
      const int ks = grpi >> 1, i0 = (grpi & 1) * HM;
      if (MAY_BE_FIRST && j == 0 && ks == 0) {
#if CTRLV_PP_ASM_STAGING
        // (the strip reads as asm statements: see stg_write16 -- the compiler put a vmcnt(0) in front of them, i.e. every
        //  tile opened by waiting for all of its LDS-DMA in flight, the three half-steps of look-ahead included)
        f32x4 bq[TN][4];
        bias_read_blocks<TN>(bq, (unsigned)(unsigned long)LDS_PTR(bias_lds + 16 * hsel));
        bias_wait_blocks<TN>(bq);
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          f32x16 bc;
#pragma unroll
          for (int q = 0; q < 4; ++q) { bc[4 * q] = bq[n][q].x; bc[4 * q + 1] = bq[n][q].y; bc[4 * q + 2] = bq[n][q].z; bc[4 * q + 3] = bq[n][q].w; }
#pragma unroll
          for (int i = i0; i < i0 + HM; ++i)
            acc[i][n] = mfma_32x32x16(wf[n][0], af[i][0], bc);
        }
#else
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const f32x16 bc = bias_c(n);
#pragma unroll
          for (int i = i0; i < i0 + HM; ++i)
            acc[i][n] = mfma_32x32x16(wf[n][0], af[i][0], bc);
        }
#endif
      } else {
#pragma unroll
        for (int i = i0; i < i0 + HM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[i][n] = mfma_32x32x16(wf[n][ks], af[i][ks], acc[i][n]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (k * 4 / NC == grpi) issue_piece(NL + k);
      if (grpi == 3) issue_end();
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(t4);
    // post-C barrier (pairs with the other group's post-L barrier)
    raw_barrier();
    STAMP(t5);
#if defined(CTRLV_PP_STAMP) && CTRLV_PP_STAMP == 2     // whole C phase of half-step 0 / 1-2 / >= 3 (c_epi: see the tile loop)
    if (j == 0) { STAMP_ADD(c_mfma, t3, t5); } else if (j < 3) { STAMP_ADD(c_cbar, t3, t5); } else { STAMP_ADD(c_epi, t3, t5); }
#else
    STAMP_ADD(c_mfma, t3, t4);
    STAMP_ADD(c_cbar, t4, t5);
#endif
  };

  for (int tr = 0; tr < my_ntiles; ++tr) {
    const int tile = my_first + tr * G;
    int mt_, nt_;
    tile_mn(tile, mt_, nt_);
    const int ksl = KS ? mt_ / tiles_m_real : 0;
    const int bm = (KS ? mt_ - ksl * tiles_m_real : mt_) * BM, bn = nt_ * BN;
    // The issue stream runs three half-steps ahead of the consuming one: it stays in this tile for J-3 half-steps and
    // then moves to the block's next tile (two K loops, so that the per-tile lane state is loop-invariant in each --
    // one loop with a conditional switch costs a dozen register copies per half-step).
    // (the three-iteration tail is kept rolled: hipcc unrolls it fully, renames the accumulators between the copies and
    // the 320-wide tile then needs 246-256 VGPRs instead of 233-240; the main loop keeps the compiler's own partial
    // unrolling, which makes the ring-slot offsets constants: rolled it measured 3 % slower on the long-K convs)
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    if constexpr (HALO) {
      // triples of half-steps (dx = -1, 0, +1): J = 9 Cin / 32 is a multiple of 3.  The row-halo stream is one triple
      // ahead of the weight stream, which is three half-steps ahead of this loop: it changes tile at J - 6, the weights
      // at J - 3.
      for (int j = 0; j < J - 6; j += 3) {
        half_step(j, false, std::true_type{}, P0{}); ++g;
        half_step(j + 1, false, std::true_type{}, P1{}); ++g;
        half_step(j + 2, false, std::true_type{}, P2{}); ++g;
      }
      next_tile_a(is_tile + G);
      half_step(J - 6, false, std::false_type{}, P0{}); ++g;
      half_step(J - 5, false, std::false_type{}, P1{}); ++g;
      half_step(J - 4, false, std::false_type{}, P2{}); ++g;
      is_tile += G;
      next_tile(is_tile);
      half_step(J - 3, false, std::false_type{}, P0{}); ++g;
      half_step(J - 2, false, std::false_type{}, P1{}); ++g;
      half_step(J - 1, true, std::false_type{}, P2{}); ++g;
    } else {
      for (int j = 0; j < J - 3; ++j, ++g) half_step(j, false, std::true_type{}, P0{});
      is_tile += G;
      next_tile(is_tile);
#pragma clang loop unroll(disable)
      for (int j = J - 3; j < J; ++j, ++g) half_step(j, j == J - 1, std::false_type{}, P0{});
    }
    // Tile boundary.  Group 0 takes one EXTRA barrier before its epilogue (it pairs with group 1's last post-C
    // barrier) and group 1 one after its epilogue (pairing with group 0's first post-L barrier of the next tile), so
    // that the two epilogues run CONCURRENTLY instead of each group stalling at a barrier for the whole epilogue of
    // the other (stamps: "C:barrier" was 1.3-2.5x the epilogue itself on the K = 320 layers).  The extra barrier also
    // orders group 0's staging writes after group 1's last load phase, which still reads those ring pieces.
    if (grp == 0) raw_barrier();
    STAMP(t6);
    {
      // wave-private staging: this wave's own four DMA pieces of the slot consumed last (every wave retired its reads
      // of that slot before the last barrier; only this wave's own DMA, issued after this epilogue, refills them)
      char* s0 = HALO ? smem + A_RING + ((g - 1) & (NH - 1)) * B_SLOT - A_SLOT : smem + ((g - 1) & (NH - 1)) * SLOT;
      // (HALO: s0 + A_SLOT = the weight slot consumed last; the two A pieces are the wave's pieces 0 / 1 of the row-halo
      //  slot of the last triple, which is refilled two triples on, i.e. after this epilogue)
      char* sa0 = HALO ? smem + (((g - 1) / 3) % NA) * A_HSLOT : s0;
      // bias columns of the next tile: loaded now, parked in the LDS strip after this epilogue's last bias read
      const bool refill = (tiles_n > 1 || KS) && tr + 1 < my_ntiles;
      u32x4_t nb = {0, 0, 0, 0};
      if (refill) {
        int mtn, ntn;
        tile_mn(tile + G, mtn, ntn);
        nb = pp_bias_load<WTN>(d, ntn * BN + wc * WTN, lane, KS && mtn >= tiles_m_real);
      }
      // The epilogue's lane constants (staging / read-back offsets, output column) do not depend on the tile: hipcc hoists
      // them out of the persistent loop and keeps them live (or spills them) around the K loop, which has no registers to
      // spare.  An opaque copy of the lane id makes them per-tile values: a dozen VALU instructions per tile instead.
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      if constexpr (KS) {
        // raw fp32 accumulators of this slice, straight from the MFMA layout (lane: row r32 of a 32-row block, four quads
        // of 4 consecutive columns): 16-byte stores, no staging
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((float*)d.out + (long)ksl * d.M * d.ldo), 0, (int)((long)d.M * d.ldo * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = bm + wr * WTM + i * 32 + r32;
#pragma unroll
          for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int col = bn + wc * WTN + n * 32 + 8 * q + 4 * hsel;
              const u32x4_t pv = {__float_as_uint(acc[i][n][4 * q]), __float_as_uint(acc[i][n][4 * q + 1]),
                                  __float_as_uint(acc[i][n][4 * q + 2]), __float_as_uint(acc[i][n][4 * q + 3])};
              __builtin_amdgcn_raw_buffer_store_b128(pv, rsP, (m < d.M && col < d.N) ? (unsigned)((m * d.ldo + col) * 4) : kOOB, 0, 0);
              store_data_hazard_guard(pv);
            }
        }
      } else
      gemm_epilogue_lds<TM, TN, GEGLU, EPI, RAW, GNS, LO>(d, acc, bm, bn, wr, wc, WTM, WTN, lane_e, sa0 + wid * 1024, sa0 + (NW + wid) * 1024,
                                s0 + A_SLOT + wid * 1024,
                                OWN_P3 ? smem + P3_OFF + wid * 1024 : s0 + A_SLOT + (NW + wid) * 1024, bias_lds,
                                smem + TAB_OFF,
                                // GNS + HALO: the last 6 KiB of a row-halo slot are never written (a slot has <= 18 real
                                // pieces: conv_halo_geometry; dummies go to the dummy KiB): four waves' strips per slot
                                (GNS && HALO) ? smem + (wid >> 2) * A_HSLOT + 18 * 1024 + (wid & 3) * (2 * WTN * 4) : nullptr);
      if (refill) pp_bias_store<WTN>(bias_lds, nb, lane);
      // The accumulators are dead here -- the next tile's first MFMAs overwrite them from a literal-zero C operand --
      // but that redefinition sits behind a `j == 0` test inside the K loop, so the compiler would keep all 128-160
      // registers live through the epilogue.  An empty asm that "defines" them ends the old live ranges at their last
      // staging write: the epilogue's prefetch window and addresses then live in retired accumulator registers.
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n) asm volatile("" : "=v"(acc[i][n]));
    }
    after_epi = true;
    STAMP(t7);
#if !(defined(CTRLV_PP_STAMP) && CTRLV_PP_STAMP == 2)
    STAMP_ADD(c_epi, t6, t7);
#endif
    if (grp == 1 && tr + 1 < my_ntiles) raw_barrier();
  }
  // the issue stream ran three half-steps past the end (zero-filled pieces): nothing may be in flight when the
  // workgroup's LDS is released
  wait_vmcnt<0>();
  CTRLV_CLOCK_END();
#ifdef CTRLV_PP_STAMP
  STAMP(t_end);
  if (lane == 0 && d.V != nullptr && d.vmode == 0) {
    unsigned long long* o = (unsigned long long*)d.V + ((long)blockIdx.x * 8 + wid) * 10;
    o[0] = t_end - t_begin; o[1] = c_lread; o[2] = c_lissue; o[3] = c_lwait; o[4] = c_lbar; o[5] = c_mfma;
    o[6] = c_cbar; o[7] = c_epi; o[8] = (unsigned long long)my_ntiles * J; o[9] = (unsigned long long)my_ntiles;
  }
#endif
#endif
}

template <int BN, int WM, int WN, int MODE, bool GEGLU, int EPI, bool HAS_A2 = false, bool RAW = false, bool HALO = false,
          bool GNS = false, bool LO = false>
int launch_one(const ctrlv_gemm_desc& d, bool persistent, hipStream_t stream) {
  // DMA ring + one bias strip (BN / WN floats) per wave + one dummy piece (ragged B piece count / row-halo staging) +
  // private fourth staging pieces (128-wide tile only) + the Phi table (GEGLU only); HALO: three 24-KiB row-halo slots in
  // place of the four A tiles
  constexpr int smem = (HALO ? 3 * 24 * 1024 + 4 * BN * 64 : 4 * (256 + BN) * 64) + BN * WM * 4 +
                       (((BN / 16) % 8 || HALO) ? 1024 : 0) + ((BN / 16) < 16 ? 8 * 1024 : 0) + (GEGLU ? kGeluTabBytes : 0);
  static_assert(smem <= 160 * 1024, "ping-pong tile does not fit the LDS");
  // per-device caches (a process may drive several GPUs; the dynamic-LDS attribute is per device code object)
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  auto kfn = gemm_pp_kernel<BN, WM, WN, MODE, GEGLU, EPI, HAS_A2, RAW, HALO, GNS, LO>;
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[dev] = true;
  }
  const int num_cu = ctrlv_num_cu(dev);
  const int tiles = ((d.M + 255) / 256) * ((d.N + BN - 1) / BN) * (EPI == 8 ? d.ksplit : 1);
  // persistent: one 512-thread workgroup per CU, every workgroup the same number of tiles.  1800 tiles (the N = 320 layers of
  // the 72 x 128 level) are 8 rounds on 256 CUs with 8 workgroups in the last one: 225 workgroups finish at the same time
  // and leave 31 CUs to the other stream's kernels (and their power to the clock) for the whole launch, not for its tail.
  int grid = tiles;
  if (persistent && tiles > num_cu) {
    const int rounds = (tiles + num_cu - 1) / num_cu;
    grid = ctrlv_debug().pp_balanced ? (tiles + rounds - 1) / rounds : num_cu;
  }
  // Column-group width of the tile order (see the kernel), from a traffic model of the L2 <-> fabric reads (checked against
  // TCC_EA0_RDREQ, tools/pmc_fetch.sh): an XCD's 32 CUs work on 32 consecutive tile numbers.
  //   row-major (cgrp = tiles_n): A once; the weights once per XCD if they fit its L2, else once per 32-tile window
  //   groups of c column tiles (c weight tiles resident, <= 3 MB): A once per group, the weights once per XCD
  // The 3x3 / temporal gathers keep the row-major order (their weight tiles are streamed along K by CUs that run in step).
  const int tiles_n = (d.N + BN - 1) / BN;
  int cgrp = tiles_n;
  {
    const int forced = ctrlv_debug().pp_cgrp;        // -1 row-major, 0 model (default), n = fixed width
    const double w_tile = (double)BN * d.taps * d.Cin * 2, w_all = (double)d.N * d.taps * d.Cin * 2;
    const double a_all = (double)d.M * d.Cin * 2, budget = 3.0 * 1048576.0;
    if (forced > 0) cgrp = forced < tiles_n ? forced : tiles_n;
    else if (forced == 0 && MODE == 0 && tiles_n > 1) {
      const double windows = (double)tiles / 32.0;
      const double w_per_window = w_tile * (tiles_n < 32 ? tiles_n : 32);
      const double cost_row = a_all + (w_all <= 3.5 * 1048576.0 ? 8.0 * w_all : windows * w_per_window);
      const int cmax = (int)(budget / w_tile);
      if (cmax >= 1 && cmax < tiles_n) {
        const double cost_grp = a_all * ((tiles_n + cmax - 1) / cmax) + 8.0 * w_all;
        if (cost_grp < 0.8 * cost_row) cgrp = cmax;
        // measured inside the model (tools/shape_table.py, CTRLV_PP_CGRP = 1..8 in one session): the C = 640 GEGLU projection
        // on the 320-wide tile is fastest in PAIRS of column tiles (15.3 ms per step against 16.3 row-major, 15.9-16.4 at
        // widths 3..8) -- a pair's output rows are five whole 128-B lines, a single tile's 2.5; the 256-wide tile of the
        // C = 1280 projections gains up to the widest group that fits (13.5 against 14.5 ms)
        if (GEGLU && BN == 320 && cgrp != tiles_n) cgrp = 2;
      }
    }
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), smem, stream, d, cgrp);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

// epilogue operand set of a descriptor (see EPI above); -1 if the ping-pong kernels do not serve it
inline int pp_epi_of(const ctrlv_gemm_desc& d) {
  if (d.ksplit > 0) return (d.act || d.R1 || d.R2 || d.vmode || d.geglu || d.A2 || d.w_cin <= 0) ? -1 : 8;   // K slices: raw fp32
  if (d.act || (d.out_f32 & 1)) return -1;
  const int e = (d.vmode ? 1 : 0) | (d.R1 ? 2 : 0) | (d.R2 ? 4 : 0);
  if (e == 0 || e == 1 || e == 2 || e == 3 || e == 6) return e;
  return -1;
}

// does this descriptor carry SPLIT trunk planes?
inline bool pp_split_io(const ctrlv_gemm_desc& d) { return d.R1_lo || d.R2_lo || d.out_lo; }

// SPLIT residual-trunk launches (R1_lo / R2_lo / out_lo): the 256x320 tile of the fp16 element library only -- the trunk's
// widths are multiples of 320 and its writers are {bias} (proj_in, shortcuts incl. the skip-concat form, the resampling
// convs), {R1} (conv2, temporal conv2 = AlphaBlender, feed-forward / proj_out), {R1, V} (attention out-projections with the
// one-key cross-attention vector, ff_in with the frame embedding) and {R1, R2} (the temporal block's folded AlphaBlender).
template <int BN, int WM, int WN, int MODE>
int launch_epi_split(const ctrlv_gemm_desc& d, bool persistent, hipStream_t stream) {
#ifdef CTRLV_ELEM_F16
  if constexpr (BN == 320) {
    const int e = pp_epi_of(d);
    if constexpr (MODE == 0) {
      switch (e) {
        case 0:
          if (d.A2) return launch_one<BN, WM, WN, MODE, false, 0, true, false, false, false, true>(d, persistent, stream);
          return launch_one<BN, WM, WN, MODE, false, 0, false, false, false, false, true>(d, persistent, stream);
        case 2: return launch_one<BN, WM, WN, MODE, false, 2, false, false, false, false, true>(d, persistent, stream);
        case 3: return launch_one<BN, WM, WN, MODE, false, 3, false, false, false, false, true>(d, persistent, stream);
        case 6: return launch_one<BN, WM, WN, MODE, false, 6, false, false, false, false, true>(d, persistent, stream);
        default: break;
      }
    } else if constexpr (MODE == 1) {
      if (ctrlv_conv_halo_order(d)) {
        // (round 6) a trunk writer that also feeds a GroupNorm -- conv2 of a res block -- writes the norm's chunk partials
        // from the same fp32 values it splits into hi + lo: the split mode no longer costs those norms a statistics pass
        if (e == 2 && d.gn_partials) return launch_one<BN, WM, WN, MODE, false, 2, false, false, true, true, true>(d, persistent, stream);
        if (e == 2) return launch_one<BN, WM, WN, MODE, false, 2, false, false, true, false, true>(d, persistent, stream);
        if (e == 0) return launch_one<BN, WM, WN, MODE, false, 0, false, false, true, false, true>(d, persistent, stream);
      } else {
        if (e == 0) return launch_one<BN, WM, WN, MODE, false, 0, false, false, false, false, true>(d, persistent, stream);
        if (e == 2) return launch_one<BN, WM, WN, MODE, false, 2, false, false, false, false, true>(d, persistent, stream);
      }
    } else {
      // (temporal conv2 = the res block's AlphaBlender epilogue, in front of a transformer's opening GroupNorm)
      if (e == 2 && d.gn_partials) return launch_one<BN, WM, WN, MODE, false, 2, false, false, false, true, true>(d, persistent, stream);
      if (e == 2) return launch_one<BN, WM, WN, MODE, false, 2, false, false, false, false, true>(d, persistent, stream);
    }
  }
#endif
  (void)persistent; (void)stream;
  ctrlv_set_error("ctrlv_gemm: split trunk planes are not served by this ping-pong tile / epilogue (tile 6 of the fp16 "
                  "element library; epilogues {bias}, {R1}, {R1,V}, {R1,R2})");
  return CTRLV_E_BAD_ARG;
}

template <int BN, int WM, int WN, int MODE>
int launch_epi(const ctrlv_gemm_desc& d, bool persistent, hipStream_t stream) {
  if (pp_split_io(d)) return launch_epi_split<BN, WM, WN, MODE>(d, persistent, stream);
  if constexpr (BN == 320 && MODE != 0) {
    // producer-side GroupNorm statistics (gemm_epilogue_lds, GNS): the conv1 / conv2 / temporal conv1 launches of a res
    // block, whose outputs go straight into a GroupNorm, and its temporal conv2 (AlphaBlender epilogue, {R1}) when the
    // block is followed by a transformer (whose first op is a GroupNorm).  ctrlv_gemm has checked
    // ctrlv_gemm_gn_partials_serves(d).
    if (d.gn_partials) {
      const int e = pp_epi_of(d);
      if constexpr (MODE == 1) {
        if (e == 1) return launch_one<BN, WM, WN, MODE, false, 1, false, false, true, true>(d, persistent, stream);
        if (e == 2) return launch_one<BN, WM, WN, MODE, false, 2, false, false, true, true>(d, persistent, stream);
      } else {
        if (e == 1) return launch_one<BN, WM, WN, MODE, false, 1, false, false, false, true>(d, persistent, stream);
        if (e == 2) return launch_one<BN, WM, WN, MODE, false, 2, false, false, false, true>(d, persistent, stream);
      }
      ctrlv_set_error("ctrlv_gemm: gn_partials not served for this launch");
      return CTRLV_E_BAD_ARG;
    }
  }
  if constexpr (BN >= 256) {
    if (d.ksplit > 0) {
      if (pp_epi_of(d) == 8) return launch_one<BN, WM, WN, MODE, false, 8>(d, persistent, stream);
      ctrlv_set_error("ctrlv_gemm: K-slice launch with epilogue operands");
      return CTRLV_E_BAD_ARG;
    }
  }
  if constexpr (MODE == 1) {
    // stride-1 3x3 convs whose row width divides the tile: row-halo staging (the kernel AND the K order: see
    // conv_halo_geometry / ctrlv_conv_halo_order).  A/B handle: CTRLV_CONV_HALO=0 = the per-tap gather in tap-major order.
    if (ctrlv_conv_halo_order(d)) {
      switch (pp_epi_of(d)) {
        case 0: return launch_one<BN, WM, WN, MODE, false, 0, false, false, true>(d, persistent, stream);
        case 1: return launch_one<BN, WM, WN, MODE, false, 1, false, false, true>(d, persistent, stream);
        case 2: return launch_one<BN, WM, WN, MODE, false, 2, false, false, true>(d, persistent, stream);
        default: break;
      }
    }
  }
  switch (pp_epi_of(d)) {
    case 0:
      if constexpr (MODE == 0) {
        if (d.A2) return launch_one<BN, WM, WN, MODE, false, 0, true>(d, persistent, stream);
      }
      return launch_one<BN, WM, WN, MODE, false, 0>(d, persistent, stream);
    case 1: return launch_one<BN, WM, WN, MODE, false, 1>(d, persistent, stream);
    case 2: return launch_one<BN, WM, WN, MODE, false, 2>(d, persistent, stream);
    case 3:
      if constexpr (MODE == 0) return launch_one<BN, WM, WN, MODE, false, 3>(d, persistent, stream);
      break;
    case 6:
      if constexpr (MODE == 0) return launch_one<BN, WM, WN, MODE, false, 6>(d, persistent, stream);
      break;
    default: break;
  }
  ctrlv_set_error("ctrlv_gemm: epilogue operand combination not served by the ping-pong kernels");
  return CTRLV_E_BAD_ARG;
}

}  // namespace
