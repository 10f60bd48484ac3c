// Gather-GEMM for gfx950 at FOUR waves per SIMD on v_mfma_f32_16x16x32 (round 5): 256 x BN tile, 16 waves (1024 threads),
// wave tile 64 x (BN / 4), <= 128 VGPRs per wave.
//
// Why a second core.  The ping-pong kernel (gemm_pp_kernel.h: v_mfma_f32_32x32x16, 8 waves, two per SIMD)
// spends 83 % of its K loop on the matrix pipe and the chip answers that load with a 1.7 GHz clock; the bare loop of
// tools/micro/mfma_shape.hip says the 16x16x32 shape holds 1.9 GHz for the same FLOPs (MI355X_MICROARCH "DVFS give-back"
// item 7) -- but one wave alone issues a 16x16x32 only every 26-33 cycles (its 16-cycle pipe slot is shorter than the
// wave's own issue cadence), so the shape pays only with TWO waves of a SIMD in their MFMA phase at the same time.  Hence:
//   * 16 waves in two GROUPS of eight (waves w and w + 8 ... every SIMD hosts two waves of each group); group 1 runs one
//     barrier slot behind group 0: while the two group-0 waves of a SIMD interleave their 2 x 20 MFMAs from registers
//     (640 pipe cycles), its two group-1 waves read their fragments, issue their LDS-DMA pieces and wait -- the schedule of
//     the ping-pong kernel with every "wave" replaced by a pair that halves its tile.
//   * the same LDS ring: 4 slots of (256 + BN) rows x 64 B (one K half-step of 32), three half-steps of LDS-DMA in flight
//     across raw s_barriers, counted vmcnt.  36 one-KiB pieces per half-step on 16 waves: wave w issues A piece w, weight
//     piece w and -- BN = 320 only, w < 4 -- weight piece 16 + w; the counted waits differ by that wave-uniform case.
//   * fragment layout of 16x16x32: lane l holds row (l & 15), k-chunk (l >> 4) -- one ds_read_b128 per 16-row block; the
//     64-B rows are chunk-swizzled on the DMA source address with the permutation {0,2,3,1}[(row >> 2) & 3], which makes
//     these reads bank-conflict-free (tools/micro/mfma_shape.hip).
//   * the WEIGHT tile is the A operand (as in the ping-pong kernel), so a lane owns output row (l & 15) and registers
//     0..3 of an accumulator are FOUR CONSECUTIVE A-operand rows 4 (l >> 4) + r.  Which weight row is fragment row i is
//     free -- it is only an LDS address -- so two neighbouring 16-column blocks are read INTERLEAVED in groups of four
//     (fragment row i of the even block = column 8 (i >> 2) + (i & 3), of the odd block = the same + 4): a lane then holds
//     EIGHT consecutive output columns of its row in the two accumulators -- the 16-byte residual reads and stores the
//     ping-pong epilogue gets out of a round trip through LDS come straight out of the registers here.  GEGLU: the
//     (16 value | 16 gate) row blocks of the packed weight are read as (value even, value odd, gate even, gate odd) of a
//     32-output-column range: value and gate of a column meet in one lane, eight outputs per lane.
// Arithmetic: one v_mfma_f32_16x16x32 sums a whole K half-step; its internal order is not the 32x32x16 kernels', so a
// layer is given to this core BY ITS SHAPE ONLY (ctrlv_gemm_w16_serves) and then runs here at EVERY row count -- a clip's
// bits must not depend on the batch it is computed in.
#pragma once
#include "gemm_pp_kernel.h"      // wait_vmcnt / barriers / store guard / pp_epi_of / pp_vtable_rows / conv geometry helpers

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4v;

__device__ __forceinline__ f32x4v mfma_16x16x32(const elx8& a, const elx8& b, const f32x4v& c) {
#ifdef CTRLV_ELEM_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// EPI: bit 0 row-vector table V, bit 1 residual R1, bit 2 residual R2 (bias, s_acc, n_scale2 always honoured).
// (A single-stream schedule -- every wave reads, issues, computes and retires in one stream, one barrier per half-step, the
//  four waves of a SIMD covering each other's fragment-read latency -- was built and measured: within +-4 % of this one on
//  every shape, profiles/r05_w16_sweeps.txt; tools/experiments/gemm_w16_kernel_r05_single_stream_schedule.h.)
template <int BN, int MODE, bool GEGLU, int EPI>
__global__ __launch_bounds__(1024) void gemm_w16_kernel(const ctrlv_gemm_desc d, const int cgrp) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, NW = 16, NH = 4, WM = 4, WN = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;                 // 64 x 80 (BN 320) or 64 x 64 (BN 256)
  constexpr int TM = WTM / 16, TN = WTN / 16;                 // 4 x 5 / 4 x 4 accumulators of 16 x 16
  constexpr int NPAIR = TN / 2, ODD = TN & 1;                 // column blocks read interleaved in pairs (+ a plain last one)
  static_assert(!GEGLU || (TN == 4), "GEGLU: a wave's weight rows are two whole (value | gate) blocks");
  constexpr int A_SLOT = BM * 64, B_SLOT = BN * 64, SLOT = A_SLOT + B_SLOT;
  constexpr int B_TOT = BN / 16;                              // weight pieces per half-step (16 or 20)
  constexpr bool UNEVEN = B_TOT > NW;                         // BN 320: waves 0..3 carry a third piece
  constexpr int TAB_OFF = NH * SLOT;                          // Phi table of the GEGLU epilogue (common.h)

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  CTRLV_CLOCK_BEGIN();

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 3;                                   // waves w, w+4, w+8, w+12 share a SIMD: two of each group
  const int wr = (wid >> 2) & 3, wc = wid & 3;                // (group = the upper two row blocks of waves)
  const int r16 = lane & 15, q4 = lane >> 4;
  const bool third = UNEVEN && wid < B_TOT - NW;              // this wave issues a third piece per half-step

  const int tiles_n = (d.N + BN - 1) / BN;
  const int tiles_m = (d.M + BM - 1) / BM;
  const int ntiles = tiles_m * tiles_n;
  const int G = gridDim.x;
  const int grp_full = tiles_n / cgrp, grp_tiles = tiles_m * cgrp;
  auto tile_mn = [&](int t, int& mt, int& nt) {             // column-group tile order (gemm_pp_kernel.h)
    const int gi = t / grp_tiles;
    if (gi < grp_full) {
      const int rem = t - gi * grp_tiles;
      mt = rem / cgrp; nt = gi * cgrp + (rem - mt * cgrp);
    } else {
      const int rem = t - grp_full * grp_tiles, w = tiles_n - grp_full * cgrp;
      if (w > 0) { mt = rem / w; nt = grp_full * cgrp + (rem - mt * w); }
      else { mt = tiles_m; nt = 0; }
    }
  };
  const int my_first = xcd_remap(blockIdx.x, G);
  const int my_ntiles = (ntiles - my_first + G - 1) / G;
  const int J = d.taps * (d.Cin >> 5);                       // half-steps per tile (>= 4)
  const long ktot = (long)d.taps * d.Cin;

  // ---- LDS-DMA addressing (the ping-pong kernel's: per-lane row offset once per tile, scalar tap / channel offset per
  // half-step, out-of-range offsets read zeros).  Chunk swizzle of the 64-B rows: physical chunk c of row r holds logical
  // chunk c ^ kPerm[(r >> 2) & 3].
  const int prow = lane >> 2, pslot = lane & 3;
  const unsigned coff = (unsigned)((pslot ^ ((0x1320 >> (((prow >> 2) & 3) * 4)) & 3)) * 16);
  const unsigned kOOB = 0xFFFFFFFFu;
  const long a_rows = MODE == 1 ? (long)(d.M / (d.Ho * d.Wo)) * d.H * d.Wd : (long)d.M;
  const int bias_rows = MODE == 1 ? d.Wd + 1 : (MODE == 2 ? d.S : 0);
  const long bias_a = (long)bias_rows * d.lda * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((const char*)d.A - bias_a), 0, (int)(a_rows * d.lda * 2 + bias_a), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.W, 0, (int)((long)d.N * ktot * 2), 0x00020000);
  unsigned a_voff = kOOB, a_mask = 0, b_voff0 = kOOB, b_voff1 = kOOB;
  auto setup = [&](int tile) {
    int mt_, nt_;
    tile_mn(tile, mt_, nt_);
    const int bm = mt_ * BM, bn = nt_ * BN;
    {
      const int m = bm + wid * 16 + prow;
      const bool ok = m < d.M;
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
        const int y0 = d.up ? yo : cy, x0 = d.up ? xo : cx;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int yi = y0 + t / 3 - 1, xi = x0 + t % 3 - 1;
          if (ok && (unsigned)yi < (unsigned)hl && (unsigned)xi < (unsigned)wl) mask |= 1u << t;
        }
      } else if (MODE == 2) {
        const int f = (m / d.S) % d.F;
        mask = ok ? ((f > 0 ? 1u : 0u) | 2u | (f < d.F - 1 ? 4u : 0u)) : 0u;
      }
      a_voff = ok ? (unsigned)row * (unsigned)(d.lda * 2) + coff : kOOB;
      a_mask = mask;
    }
    {
      const int n0 = bn + wid * 16 + prow;
      b_voff0 = n0 < d.N ? (unsigned)n0 * (unsigned)(ktot * 2) + coff : kOOB;
      const int n1 = bn + (NW + wid) * 16 + prow;
      b_voff1 = (third && n1 < d.N) ? (unsigned)n1 * (unsigned)(ktot * 2) + coff : kOOB;
    }
  };

  int is_tap = 0, is_cc = 0;
  auto issue = [&](int g) {                                   // all pieces of one half-step (this wave's two or three)
    char* sa = smem + (g & (NH - 1)) * SLOT;
    char* sb = sa + A_SLOT;
    int roff = 0, dyo = 0, dxo = 0;
    if (MODE == 1) {
      dyo = is_tap / 3 - 1; dxo = is_tap % 3 - 1;
      roff = d.up ? 0 : (dyo + 1) * d.Wd + dxo + 1;
    } else if (MODE == 2) {
      roff = is_tap * d.S;
    }
    const unsigned so_a = __builtin_amdgcn_readfirstlane((unsigned)(roff * d.lda * 2) + (unsigned)(is_cc * 2));
    const unsigned so_w = __builtin_amdgcn_readfirstlane((unsigned)(((MODE == 0 ? 0 : is_tap * d.Cin) + is_cc) * 2));
    unsigned voff = a_voff;
    if (MODE != 0) {
      if (MODE == 1 && d.up) {
        const int oy = ((int)((a_mask >> 16) & 1) + dyo) >> 1, ox = ((int)((a_mask >> 17) & 1) + dxo) >> 1;
        voff += (unsigned)((oy * d.Wd + ox + d.Wd + 1) * d.lda * 2);
      }
      if (!((a_mask >> is_tap) & 1u)) voff = kOOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(sa + wid * 1024), 16, voff, so_a, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(sb + wid * 1024), 16, b_voff0, so_w, 0, 0);
    if (third) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_PTR(sb + (NW + wid) * 1024), 16, b_voff1, so_w, 0, 0);
    is_cc += 32;
    if (MODE != 0 && is_cc == d.Cin) { is_cc = 0; ++is_tap; }
  };
  auto next_tile = [&](int tile) {
    setup(tile);
    is_tap = 0;
    is_cc = 0;
  };
  // own pieces of the NEXT half-step landed; the two half-steps behind it may stay in flight (counted, per-wave piece count).
  // `vmcnt` counts loads AND stores in order: in the first two half-steps after an epilogue the pieces retired here were
  // issued BEFORE that epilogue's NSTORE stores, which may stay in flight too (gemm_pp_kernel.h: waiting for them is waiting
  // for the tile's write acknowledgement).
  constexpr int NSTORE = GEGLU ? TM : TM * (NPAIR + ODD);
  auto wait_next = [&](bool behind_epilogue) {
    if (behind_epilogue) { if (third) wait_vmcnt<6 + NSTORE>(); else wait_vmcnt<4 + NSTORE>(); }
    else { if (third) wait_vmcnt<6>(); else wait_vmcnt<4>(); }
  };

  // ---- fragment addresses.  Activation rows (MFMA B operand): row wr*64 + i*16 + r16, logical chunk q4.
  const int perm_r = (0x1320 >> (((r16 >> 2) & 3) * 4)) & 3;           // swizzle of a row whose (row >> 2) & 3 == r16 >> 2
  const int a_frag = (wr * WTM + r16) * 64 + ((q4 ^ perm_r) * 16);
  // Weight rows (MFMA A operand), interleaved pairs: fragment row i of the EVEN block of pair p is tile row 32p + 8(i>>2) +
  // (i&3), of the ODD block the same + 4; (row >> 2) & 3 = (2 (i>>2) [+ 1]) & 3.  GEGLU: value rows {0,8,32,40}[i>>2] +
  // (i&3) [+ 4], gate rows = value rows + 16 -- the same swizzle classes.
  const int j4 = r16 >> 2, i4 = r16 & 3;
  const int rowE = GEGLU ? (32 * (j4 >> 1) + 8 * (j4 & 1) + i4) : (8 * j4 + i4);
  const int swE = (0x1320 >> ((((rowE >> 2)) & 3) * 4)) & 3, swO = (0x1320 >> ((((rowE + 4) >> 2) & 3) * 4)) & 3;
  const int wE_frag = A_SLOT + (wc * WTN + rowE) * 64 + ((q4 ^ swE) * 16);
  const int wO_frag = A_SLOT + (wc * WTN + rowE + 4) * 64 + ((q4 ^ swO) * 16);
  const int wL_frag = A_SLOT + (wc * WTN + NPAIR * 32 + r16) * 64 + ((q4 ^ perm_r) * 16);   // plain last block (BN 320)

  if constexpr (GEGLU) gelu_table_fill(smem + TAB_OFF, threadIdx.x, NW * 64);
  // ---- prologue: three half-steps in flight
  int is_tile = my_first;
  next_tile(is_tile);
  issue(0);
  issue(1);
  issue(2);
  if (third) wait_vmcnt<6>(); else wait_vmcnt<4>();
  raw_barrier();

  f32x4v acc[TM][TN];
  int g = 0;
#ifdef CTRLV_PP_STAMP
  unsigned long long c_lread = 0, c_lissue = 0, c_lwait = 0, c_lbar = 0, c_mfma = 0, c_cbar = 0, c_epi = 0;
  STAMP(t_begin);
#endif
  if (grp == 1) raw_barrier();                                // stagger: group 1 runs one barrier slot behind

  bool after_epi = false;
  auto half_step = [&](bool first, bool early) {
    const char* st = smem + (g & (NH - 1)) * SLOT;
    elx8 af[TM], wf[TN];
    STAMP(t0);
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      if constexpr (GEGLU) {                                  // (value even, value odd, gate even, gate odd)
        wf[0] = *(const elx8*)(st + wE_frag);
        wf[1] = *(const elx8*)(st + wO_frag);
        wf[2] = *(const elx8*)(st + wE_frag + 16 * 64);
        wf[3] = *(const elx8*)(st + wO_frag + 16 * 64);
      } else {
        wf[2 * p] = *(const elx8*)(st + wE_frag + p * 32 * 64);
        wf[2 * p + 1] = *(const elx8*)(st + wO_frag + p * 32 * 64);
      }
    }
    if constexpr (ODD) wf[TN - 1] = *(const elx8*)(st + wL_frag);
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *(const elx8*)(st + a_frag + i * 16 * 64);
    STAMP(t1);
    issue(g + 3);
    STAMP(t1b);
    wait_next(early && after_epi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(t2b);
    lds_done_barrier();
    STAMP(t3);
    STAMP_ADD(c_lwait, t1b, t2b);
    STAMP_ADD(c_lbar, t2b, t3);
    STAMP_ADD(c_lread, t0, t1);
    STAMP_ADD(c_lissue, t1, t1b);
    __builtin_amdgcn_sched_barrier(0);
    if (first) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[i][n] = mfma_16x16x32(wf[n], af[i], f32x4v{0.f, 0.f, 0.f, 0.f});
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[i][n] = mfma_16x16x32(wf[n], af[i], acc[i][n]);
    }
    __builtin_amdgcn_sched_barrier(0);
    STAMP(t4);
    raw_barrier();
    STAMP(t5);
    STAMP_ADD(c_mfma, t3, t4);
    STAMP_ADD(c_cbar, t4, t5);
  };

  for (int tr = 0; tr < my_ntiles; ++tr) {
    const int tile = my_first + tr * G;
    int mt_, nt_;
    tile_mn(tile, mt_, nt_);
    const int bm = mt_ * BM, bn = nt_ * BN;
    // (J >= 4: the first two half-steps of a tile -- the ones whose retired pieces predate the last epilogue -- are always in
    //  front of the issue stream's move to the next tile at J - 3, except j = 1 at J = 4)
    half_step(true, true); ++g;
    if (J > 4) { half_step(false, true); ++g; }
    for (int j = 2; j < J - 3; ++j, ++g) half_step(false, false);
    is_tile += G;
    next_tile(is_tile);
#pragma clang loop unroll(disable)
    for (int j = J - 3; j < J; ++j, ++g) half_step(false, J == 4 && j == 1);
    // tile boundary: the two groups' epilogues run concurrently (gemm_pp_kernel.h)
    if (grp == 0) raw_barrier();
    STAMP(t6);
    {
      constexpr int kFlags = 0x00020000;
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int er16 = lane_e & 15, eq4 = lane_e >> 4;
      const int m0 = bm + wr * WTM + er16;                     // + i * 16
      const int wbase_n = bn + wc * WTN;
      const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, (int)((long)d.M * d.ldo * 2), kFlags);
      const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(d.bias ? (const void*)d.bias : d.W), 0, d.bias ? d.N * 4 : 0, kFlags);       // no bias: reads 0
      if constexpr (GEGLU) {
        // 32 output columns per wave: lane (row, eq4) -> columns 8 eq4 .. + 7 of them; value column c of the range is weight
        // row 32 (c >> 4) + (c & 15), its gate 16 rows further
        const char* tab = smem + TAB_OFF;
        const int oc = ((bn + wc * WTN) >> 1) + 8 * eq4;      // first output column of this lane
        const int c0 = 8 * eq4;                               // ... inside the wave's 32-column range
        const int wrow0 = wbase_n + 32 * (c0 >> 4) + (c0 & 15);                 // weight row of its first value column
        const u32x4_t bv0 = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(wrow0 * 4), 0, 0);
        const u32x4_t bv1 = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)((wrow0 + 4) * 4), 0, 0);
        const u32x4_t bg0 = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)((wrow0 + 16) * 4), 0, 0);
        const u32x4_t bg1 = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)((wrow0 + 20) * 4), 0, 0);
        const bool col_ok = oc < d.n_store && wbase_n + 32 * (c0 >> 4) < d.N;
        // training forward (raw_out, wave-uniform): the projection itself also goes out, elements [M][ld_raw] in the packed
        // column order (= weight-row order) -- the lane's eight value rows wrow0 .. + 7 and its eight gate rows 16 further are
        // two 16-byte stores.  The ROUTING of a layer must not depend on raw_out (a checkpointed recompute has to
        // run on the core its forward ran on), so this core writes it like the ping-pong tiles do.
        const bool raw = d.raw_out != nullptr;
        const __amdgpu_buffer_rsrc_t rsRaw = __builtin_amdgcn_make_buffer_rsrc(
            raw ? d.raw_out : (void*)d.W, 0, raw ? (int)((long)d.M * d.ld_raw * 2) : 0, kFlags);
        const bool raw_ok = wbase_n + 32 * (c0 >> 4) < d.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float o[8];
          const int m = m0 + i * 16;
          if (raw) {
            float pv_[8], pg_[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              pv_[r] = acc[i][0][r] + __uint_as_float(bv0[r]); pv_[4 + r] = acc[i][1][r] + __uint_as_float(bv1[r]);
              pg_[r] = acc[i][2][r] + __uint_as_float(bg0[r]); pg_[4 + r] = acc[i][3][r] + __uint_as_float(bg1[r]);
            }
            const uint4 kv = pack_elx8(pv_), kg = pack_elx8(pg_);
            const u32x4_t sv = {kv.x, kv.y, kv.z, kv.w}, sg = {kg.x, kg.y, kg.z, kg.w};
            const unsigned roff = (m < d.M && raw_ok) ? (unsigned)m * (unsigned)(d.ld_raw * 2) + (unsigned)(wrow0 * 2) : kOOB;
            pp_store_out(sv, rsRaw, roff, 0);
            pp_store_out(sg, rsRaw, roff, 32);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = geglu_tab(acc[i][0][r] + __uint_as_float(bv0[r]), acc[i][2][r] + __uint_as_float(bg0[r]), tab);
            o[4 + r] = geglu_tab(acc[i][1][r] + __uint_as_float(bv1[r]), acc[i][3][r] + __uint_as_float(bg1[r]), tab);
          }
          const uint4 pk = pack_elx8(o);
          const u32x4_t pv = {pk.x, pk.y, pk.z, pk.w};
          pp_store_out(pv, rsO, (m < d.M && col_ok) ? (unsigned)m * (unsigned)(d.ldo * 2) + (unsigned)(oc * 2) : kOOB, 0);
        }
      } else {
        const __amdgpu_buffer_rsrc_t rsR1 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((EPI & 2) ? d.R1 : d.W), 0, (EPI & 2) ? (int)((long)d.M * d.ldr1 * 2) : 0, kFlags);
        const __amdgpu_buffer_rsrc_t rsR2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((EPI & 4) ? d.R2 : d.W), 0, (EPI & 4) ? (int)((long)d.M * d.ldr2 * 2) : 0, kFlags);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((EPI & 1) ? (const void*)d.V : d.W), 0, (EPI & 1) ? (int)(pp_vtable_rows(d) * d.ldv * 4) : 0, kFlags);
        unsigned v_row[TM];
        if (EPI & 1) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int m = m0 + i * 16;
            const int mc = m < d.M ? m : 0;
            const unsigned vi = d.vmode == 1 ? (unsigned)((mc / d.vdiv) % d.vmod)
                                             : (unsigned)((((long)(mc / d.vdiv) * d.vS + (mc % d.vS)) % d.vmod));
            v_row[i] = vi * (unsigned)(d.ldv * 4);
          }
        }
        // column blocks: NPAIR interleaved pairs (8 consecutive columns per lane) + the plain last block (4 per lane)
#pragma unroll
        for (int p = 0; p < NPAIR + ODD; ++p) {
          const bool pair = p < NPAIR;
          const int ocol = wbase_n + (pair ? p * 32 + 8 * eq4 : NPAIR * 32 + 4 * eq4);
          const float sc = (wbase_n + p * 32 < d.n_scale2) ? d.s_acc2 : d.s_acc;
          const bool col_ok = ocol < d.n_store && ocol < d.N;
          const u32x4_t b0 = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)(ocol * 4), 0, 0);
          u32x4_t b1 = {0, 0, 0, 0};
          if (pair) b1 = __builtin_amdgcn_raw_buffer_load_b128(rsB, (unsigned)((ocol + 4) * 4), 0, 0);
          // residual rows of the four row blocks: issued together, consumed in order
          u32x4_t r1[TM], r2[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int m = m0 + i * 16;
            const bool ok = m < d.M && col_ok;
            if (EPI & 2) {
              const unsigned off = ok ? (unsigned)m * (unsigned)(d.ldr1 * 2) + (unsigned)(ocol * 2) : kOOB;
              if (pair) r1[i] = __builtin_amdgcn_raw_buffer_load_b128(rsR1, off, 0, 0);
              else { const auto t = __builtin_amdgcn_raw_buffer_load_b64(rsR1, off, 0, 0); r1[i] = u32x4_t{t[0], t[1], 0, 0}; }
            }
            if (EPI & 4) {
              const unsigned off = ok ? (unsigned)m * (unsigned)(d.ldr2 * 2) + (unsigned)(ocol * 2) : kOOB;
              if (pair) r2[i] = __builtin_amdgcn_raw_buffer_load_b128(rsR2, off, 0, 0);
              else { const auto t = __builtin_amdgcn_raw_buffer_load_b64(rsR2, off, 0, 0); r2[i] = u32x4_t{t[0], t[1], 0, 0}; }
            }
          }
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int m = m0 + i * 16;
            const bool ok = m < d.M && col_ok;
            float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o[r] = acc[i][pair ? 2 * p : TN - 1][r] + __uint_as_float(b0[r]);
              if (pair) o[4 + r] = acc[i][pair ? 2 * p + 1 : TN - 1][r] + __uint_as_float(b1[r]);
            }
            {
#pragma clang fp contract(off)
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = o[e] * sc;
            }
            if (EPI & 2) {
              float f[8];
              unpack_elx8(make_uint4(r1[i][0], r1[i][1], r1[i][2], r1[i][3]), f);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(d.s1, f[e], o[e]);
            }
            if (EPI & 4) {
              float f[8];
              unpack_elx8(make_uint4(r2[i][0], r2[i][1], r2[i][2], r2[i][3]), f);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(d.s2, f[e], o[e]);
            }
            if (EPI & 1) {
              const u32x4_t v0 = __builtin_amdgcn_raw_buffer_load_b128(rsV, ok ? v_row[i] + (unsigned)(ocol * 4) : kOOB, 0, 0);
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] += __uint_as_float(v0[e]);
              if (pair) {
                const u32x4_t v1 = __builtin_amdgcn_raw_buffer_load_b128(rsV, ok ? v_row[i] + (unsigned)(ocol * 4 + 16) : kOOB, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[4 + e] += __uint_as_float(v1[e]);
              }
            }
            const uint4 pk = pack_elx8(o);
            const unsigned ooff = ok ? (unsigned)m * (unsigned)(d.ldo * 2) + (unsigned)(ocol * 2) : kOOB;
            if (pair) {
              const u32x4_t pv = {pk.x, pk.y, pk.z, pk.w};
              pp_store_out(pv, rsO, ooff, 0);
            } else {
              typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
              const u32x2_t pv = {pk.x, pk.y};
              __builtin_amdgcn_raw_buffer_store_b64(pv, rsO, ooff, 0, 0);
              asm volatile("s_nop 1" ::"v"(pv));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    after_epi = true;
    STAMP(t7);
    STAMP_ADD(c_epi, t6, t7);
    if (grp == 1 && tr + 1 < my_ntiles) raw_barrier();
  }
  wait_vmcnt<0>();
  CTRLV_CLOCK_END();
#ifdef CTRLV_PP_STAMP     // (diagnostic build, tools/gemm_stamp.py: the ping-pong kernel's record format, 16 waves per workgroup)
  STAMP(t_end);
  if (lane == 0 && d.V != nullptr && d.vmode == 0) {
    unsigned long long* o = (unsigned long long*)d.V + ((long)blockIdx.x * NW + wid) * 10;
    o[0] = t_end - t_begin; o[1] = c_lread; o[2] = c_lissue; o[3] = c_lwait; o[4] = c_lbar; o[5] = c_mfma;
    o[6] = c_cbar; o[7] = c_epi; o[8] = (unsigned long long)my_ntiles * J; o[9] = (unsigned long long)my_ntiles;
  }
#endif
#endif
}

template <int BN, int MODE, bool GEGLU, int EPI>
int launch_w16(const ctrlv_gemm_desc& d, hipStream_t stream) {
  constexpr int smem = 4 * (256 + BN) * 64 + (GEGLU ? kGeluTabBytes : 0);
  static_assert(smem <= 160 * 1024, "w16 tile does not fit the LDS");
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  auto kfn = gemm_w16_kernel<BN, MODE, GEGLU, EPI>;
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[dev] = true;
  }
  const int num_cu = ctrlv_num_cu(dev);
  const int tiles_n = (d.N + BN - 1) / BN;
  const int tiles = ((d.M + 255) / 256) * tiles_n;
  int grid = tiles;
  if (tiles > num_cu) {                  // persistent, every workgroup the same number of tiles (gemm_pp_kernel.h launch_one)
    const int rounds = (tiles + num_cu - 1) / num_cu;
    grid = (tiles + rounds - 1) / rounds;
  }
  // column-group tile order: the ping-pong kernel's traffic model (launch_one)
  int cgrp = tiles_n;
  if (MODE == 0 && tiles_n > 1) {
    const double w_tile = (double)BN * d.taps * d.Cin * 2, w_all = (double)d.N * d.taps * d.Cin * 2;
    const double a_all = (double)d.M * d.Cin * 2, budget = 3.0 * 1048576.0;
    const double windows = (double)tiles / 32.0;
    const double w_per_window = w_tile * (tiles_n < 32 ? tiles_n : 32);
    const double cost_row = a_all + (w_all <= 3.5 * 1048576.0 ? 8.0 * w_all : windows * w_per_window);
    const int cmax = (int)(budget / w_tile);
    if (cmax >= 1 && cmax < tiles_n) {
      const double cost_grp = a_all * ((tiles_n + cmax - 1) / cmax) + 8.0 * w_all;
      if (cost_grp < 0.8 * cost_row) cgrp = cmax;
    }
    const int forced = ctrlv_debug().pp_cgrp;        // (the ping-pong kernel's A/B handle)
    if (forced > 0) cgrp = forced < tiles_n ? forced : tiles_n;
    else if (forced < 0) cgrp = tiles_n;
  }
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(1024), smem, stream, d, cgrp);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

}  // namespace
