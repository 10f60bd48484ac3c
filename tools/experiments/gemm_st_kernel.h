// Streamed gather-GEMM for short-K layers: 256 x 160 tile, FOUR waves, two independent workgroups per CU.
//
// Why a second schedule.  In the ping-pong kernel (gemm_pp_kernel.h) the eight waves of a CU belong to one workgroup
// and reach the epilogue of a tile together.  On the K = 320 / 640 layers the epilogue is 15-50 % of a tile
// (tools/gemm_stamp.py: GEGLU's polynomial is ~1000 VALU instructions per wave and both waves of a SIMD issue them at
// the same time; residual epilogues wait on HBM), and the matrix pipe idles through all of it.  Here the CU holds TWO
// 4-wave workgroups (one wave per SIMD each, 256 registers, <= 80 KB LDS): they share nothing and drift apart, so one
// workgroup's epilogue (VALU / memory) runs beside the other's K loop (MFMA) on every SIMD -- the hardware's own
// two-waves-per-SIMD interleave, with no barrier tying the two together.
//
// Inside a workgroup the schedule is the "streamed" one (one barrier per half-step of K = 32, fragments double-buffered
// at k16 granularity, LDS-DMA pieces in the gaps of the MFMA groups, counted vmcnt), with a ring of THREE slots
// (3 x (256 + 160) x 64 B = 78 KB): the DMA of half-step g+2 is issued during half-step g into the slot whose last
// reads every wave retired before the barrier of half-step g-1.  Waves are stacked 4 x 1: each owns 64 rows and all
// 160 columns (wave tile 64 x 160 = 2 x 5 MFMA sub-tiles, 160 accumulators -- the same as the 256 x 320 ping-pong
// tile), so one bias strip serves the workgroup.  The epilogue is gemm_epilogue_lds (shared with the ping-pong kernel:
// identical arithmetic, identical bits).  Mode 0 (nn.Linear / 1x1 conv) only.
#pragma once
#include "gemm_pp_kernel.h"

namespace {

#ifdef CTRLV_ST_PACKED_GELU
constexpr bool ST_PACKED_GELU = true;      // A/B: the ping-pong kernel's packed polynomial
#else
constexpr bool ST_PACKED_GELU = false;
#endif

template <int BN, bool GEGLU, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_st_kernel(const ctrlv_gemm_desc d) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, NW = 4, NH = 3, LEAD = NH - 1;
  constexpr int WTM = BM / NW, WTN = BN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_SLOT = BM * 64, B_SLOT = BN * 64, SLOT = A_SLOT + B_SLOT;
  constexpr int A_TOT = BM / 16, B_TOT = BN / 16;            // 1-KiB DMA pieces (16 rows x 64 B) per half-step
  constexpr int A_Q = A_TOT / NW;                            // 4 per wave
  constexpr int B_Q = (B_TOT + NW - 1) / NW;                 // 3 per wave (10 pieces: waves 2, 3 issue a dummy)
  constexpr int NPIECE = A_Q + B_Q;
  constexpr bool UNEVEN = (B_TOT % NW) != 0;
  constexpr int BIAS_OFF = NH * SLOT, DUMMY_OFF = BIAS_OFF + WTN * 4;
  constexpr int N1 = (NPIECE + 1) / 2;                       // pieces issued in the first half of a half-step
  static_assert(TM == 2 && A_TOT % NW == 0 && A_Q == 4, "wave layout");

  extern __shared__ __attribute__((aligned(1024))) char smem[];   // 3 ring slots | bias strip | dummy piece

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5;

  const int tiles_n = (d.N + BN - 1) / BN;
  const int tiles_m = (d.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int my_first = xcd_remap(blockIdx.x, G);
  const int my_ntiles = (ntiles - my_first + G - 1) / G;       // >= 1 (grid <= ntiles)
  const int J = d.Cin >> 5;                                  // half-steps per tile (> LEAD: ctrlv_gemm_st_supports)

  // ---- DMA addressing (see gemm_pp_kernel.h): per-lane row offset once per tile, a half-step adds a scalar
  const int prow = lane >> 2, pslot = lane & 3;
  const unsigned coff = (pslot ^ ((prow >> 2) & 3)) * 16;
  const unsigned kOOB = 0xFFFFFFFFu;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, (int)((long)d.M * d.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.W, 0, (int)((long)d.N * d.Cin * 2), 0x00020000);
  unsigned a_voff[A_Q], b_voff[B_Q];
  auto setup = [&](int tile) {
    const int bm = (tile / tiles_n) * BM, bn = (tile % tiles_n) * BN;
#pragma unroll
    for (int q = 0; q < A_Q; ++q) {
      const int m = bm + (q * NW + wid) * 16 + prow;
      a_voff[q] = m < d.M ? (unsigned)m * (unsigned)(d.lda * 2) + coff : kOOB;
    }
#pragma unroll
    for (int q = 0; q < B_Q; ++q) {
      const int ib = q * NW + wid;
      const int n = bn + ib * 16 + prow;
      b_voff[q] = (ib < B_TOT && n < d.N) ? (unsigned)n * (unsigned)(d.Cin * 2) + coff : kOOB;
    }
  };
  int is_cc = 0;
  char* is_sa = nullptr;
  unsigned is_so = 0;
  int is_slot = 0;                                           // ring slot the issue stream fills next
  auto issue_begin = [&]() {
    is_sa = smem + is_slot * SLOT;
    is_so = __builtin_amdgcn_readfirstlane((unsigned)(is_cc * 2));
  };
  auto issue_piece = [&](int pc) {
    if (pc < A_Q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(is_sa + (pc * NW + wid) * 1024), 16, a_voff[pc], is_so, 0, 0);
    } else {
      const int q = pc - A_Q;
      char* dst = is_sa + A_SLOT + (q * NW + wid) * 1024;
      if (UNEVEN && q == B_Q - 1 && q * NW + wid >= B_TOT) dst = smem + DUMMY_OFF;   // out-of-range source: zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(dst), 16, b_voff[q], is_so, 0, 0);
    }
  };
  auto issue_end = [&]() {
    is_cc += 32;
    is_slot = is_slot == NH - 1 ? 0 : is_slot + 1;
  };
  auto issue = [&]() {
    issue_begin();
#pragma unroll
    for (int pc = 0; pc < NPIECE; ++pc) issue_piece(pc);
    issue_end();
  };
  auto next_tile = [&](int tile) {
    setup(tile);
    is_cc = 0;
  };

  const int sw = (r32 >> 2) & 3;
  const int a_frag = (wid * WTM + r32) * 64;
  const int b_frag = A_SLOT + r32 * 64;

  // one bias strip for the workgroup: every wave stores the same 160 floats (a wave's own store precedes its own
  // reads in LDS order; the other waves' stores write identical values)
  char* const bias_lds = smem + BIAS_OFF;
  {
    const int bn0 = (my_first % tiles_n) * BN;
    const u32x4_t b = pp_bias_load<WTN>(d, bn0, lane);
    pp_bias_store<WTN>(bias_lds, b, lane);
  }
  auto bias_c = [&](int n) {
    f32x16 c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *(const float4*)(bias_lds + (n * 32 + 8 * q + 4 * hsel) * 4);
      c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
    }
    return c;
  };
  auto read_frags = [&](const char* st, int ks, bf16x8 (&af)[TM], bf16x8 (&wf)[TN]) {
    const int coff_ = ((ks * 2 + hsel) ^ sw) * 16;
#pragma unroll
    for (int n = 0; n < TN; ++n) wf[n] = *(const bf16x8*)(st + b_frag + n * 32 * 64 + coff_);
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(st + a_frag + i * 32 * 64 + coff_);
  };

  // ---- prologue: LEAD half-steps in flight, the first one landed
  int is_tile = my_first;
  next_tile(is_tile);
  issue();
  issue();
  wait_vmcnt<NPIECE>();
  raw_barrier();

  f32x16 acc[TM][TN];
  bf16x8 af0[TM], wf0[TN], af1[TM], wf1[TN];
  int rd_slot = 0;                                           // ring slot of the half-step being consumed
  read_frags(smem, 0, af0, wf0);

#ifdef CTRLV_PP_STAMP
  unsigned long long c_h1 = 0, c_wait = 0, c_bar = 0, c_h2 = 0, c_lg = 0, c_epi = 0, c_hs = 0;
  STAMP(t_begin);
#endif
  auto half_step = [&](int j, bool last_of_tile, auto first_tag) {
    constexpr bool MAY_BE_FIRST = decltype(first_tag)::value;
    const char* st = smem + rd_slot * SLOT;
    STAMP(t0);
    read_frags(st, 1, af1, wf1);
    issue_begin();
    __builtin_amdgcn_sched_barrier(0);
    // first half: k16 #0 from F0, the first N1 DMA pieces of half-step g+2 in the two gaps
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (MAY_BE_FIRST && j == 0) {
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[h][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0[n], af0[h], bias_c(n), 0, 0, 0);
      } else {
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[h][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0[n], af0[h], acc[h][n], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < N1; ++k)
        if (k * 2 / N1 == h) issue_piece(k);
      __builtin_amdgcn_sched_barrier(0);
    }
    // middle: own DMA(g+1) retired (the N1 pieces just issued may stay in flight), F1 landed; after the barrier slot
    // g+1 is complete and nobody reads slot g any more
    STAMP(t1);
    wait_vmcnt<N1>();
    STAMP(t2);
    lds_done_barrier();
    STAMP(t3);
    __builtin_amdgcn_sched_barrier(0);
    rd_slot = rd_slot == NH - 1 ? 0 : rd_slot + 1;
    if (!last_of_tile) read_frags(smem + rd_slot * SLOT, 0, af0, wf0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int n = 0; n < TN; ++n) acc[h][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1[n], af1[h], acc[h][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NPIECE - N1; ++k)
        if (k * 2 / (NPIECE - N1) == h) issue_piece(N1 + k);
      if (h == 1) issue_end();
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(t4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(t5);
    STAMP_ADD(c_h1, t0, t1); STAMP_ADD(c_wait, t1, t2); STAMP_ADD(c_bar, t2, t3); STAMP_ADD(c_h2, t3, t4);
    STAMP_ADD(c_lg, t4, t5);
#ifdef CTRLV_PP_STAMP
    ++c_hs;
#endif
  };

  for (int tr = 0; tr < my_ntiles; ++tr) {
    const int tile = my_first + tr * G;
    const int bm = (tile / tiles_n) * BM, bn = (tile % tiles_n) * BN;
    for (int j = 0; j < J - LEAD; ++j) half_step(j, false, std::true_type{});
    is_tile += G;
    next_tile(is_tile);                                      // (may be past the last tile: all rows invalid, zeros)
    for (int j = J - LEAD; j < J; ++j) half_step(j, j == J - 1, std::false_type{});
    STAMP(t6);
    {
      // wave-private staging: this wave's own four A pieces of the slot consumed last (all reads of that slot retired
      // before the last barrier; only this wave's own DMA, issued after this epilogue, refills them)
      const int done_slot = rd_slot == 0 ? NH - 1 : rd_slot - 1;
      char* s0 = smem + done_slot * SLOT;
      const bool refill = tiles_n > 1 && tr + 1 < my_ntiles;
      u32x4_t nb = {0, 0, 0, 0};
      if (refill) nb = pp_bias_load<WTN>(d, ((tile + G) % tiles_n) * BN, lane);
      gemm_epilogue_lds<TM, TN, GEGLU, EPI, ST_PACKED_GELU>(d, acc, bm, bn, wid, 0, WTM, WTN, lane, s0 + wid * 1024, s0 + (NW + wid) * 1024,
                                            s0 + (2 * NW + wid) * 1024, s0 + (3 * NW + wid) * 1024, bias_lds);
      if (refill) pp_bias_store<WTN>(bias_lds, nb, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n) asm volatile("" : "=v"(acc[i][n]));
    }
    STAMP(t7);
    STAMP_ADD(c_epi, t6, t7);
    if (tr + 1 < my_ntiles) {     // first fragments of the next tile (its slot was completed by the last barrier)
      read_frags(smem + rd_slot * SLOT, 0, af0, wf0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  wait_vmcnt<0>();                // the issue stream ran LEAD half-steps past the end (zero-filled pieces)
#ifdef CTRLV_PP_STAMP
  STAMP(t_end);
  if (lane == 0 && d.V != nullptr && d.vmode == 0) {
    unsigned long long* o = (unsigned long long*)d.V + ((long)blockIdx.x * 4 + wid) * 10;
    o[0] = t_end - t_begin; o[1] = c_h1; o[2] = c_wait; o[3] = c_bar; o[4] = c_h2; o[5] = c_lg; o[6] = 0;
    o[7] = c_epi; o[8] = c_hs; o[9] = (unsigned long long)my_ntiles;
  }
#endif
#endif
}

template <int BN, bool GEGLU, int EPI>
int launch_st_one(const ctrlv_gemm_desc& d, hipStream_t stream) {
  constexpr int smem = 3 * (256 + BN) * 64 + BN * 4 + (((BN / 16) % 4) ? 1024 : 0);
  static_assert(smem <= 80 * 1024, "two workgroups per CU need <= 80 KB each");
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  auto kfn = gemm_st_kernel<BN, GEGLU, EPI>;
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[dev] = true;
  }
  const int slots = 2 * ctrlv_num_cu(dev);
  const int tiles = ((d.M + 255) / 256) * ((d.N + BN - 1) / BN);
  const int grid = tiles > slots ? slots : tiles;            // persistent: two 256-thread workgroups per CU
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), smem, stream, d);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

}  // namespace
