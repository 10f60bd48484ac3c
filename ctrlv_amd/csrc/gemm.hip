// Gather-GEMM for gfx950: one kernel family for nn.Linear, 1x1 / 3x3 / stride-2 / upsample-fused Conv2d and the
// (3,1,1) temporal Conv3d of the SVD UNet / ControlNet, on channels-last bf16 rows.
//
//   out[m, n] = epilogue( sum_{tap, c} A[src_row(m, tap), c] * W[n, tap*Cin + c] )
//
// Structure (MI355X-first, see DESIGN.md "gather-GEMM"):
//   * BM x BN x 64 tiles; both operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): the LDS image is
//     lane-linear (8 rows x 128 B per wave-instruction) and the bank-conflict swizzle is applied on the per-lane
//     SOURCE address (guide rule 21); conv halos / padding rows / tile overhang read a 256-B zero page, so the
//     3x3 and temporal taps are pure address arithmetic -- no im2col, no halo copies in HBM.
//   * two LDS stages, the DMA of K-step t+1 is in flight while the MFMAs of step t run.
//   * v_mfma_f32_32x32x16_bf16 with the WEIGHT tile as the A operand, so every lane ends up owning one output ROW
//     (m) and 4 consecutive columns per accumulator quad: the epilogue reads residuals / writes bf16 as 8-byte
//     vectors and the per-row broadcast vectors (temb, frame pos-emb, 1-key cross-attention) cost one index per lane.
//   * XCD-aware block remap so the N-tiles that share an A row-panel (and the M-tiles that share conv halo rows)
//     sit in one XCD's L2.
#include <stdarg.h>

#include "common.h"
#include "gemm_epilogue.h"

// ping-pong schedule (gemm_pp_kernel.h, gemm_pp_m*.hip): tile 5 = 256x256, tile 6 = 256x320 (7 / 8: non-persistent)
int ctrlv_gemm_launch_pp(const ctrlv_gemm_desc& d, int tile, hipStream_t stream);
bool ctrlv_gemm_pp_supports(const ctrlv_gemm_desc& d);
bool ctrlv_conv_halo_order(const ctrlv_gemm_desc& d);      // gemm_pp_m0.hip: K order (dy, 32-channel block, dx) for this conv?
// four-waves-per-SIMD 16x16x32 core (gemm_w16_kernel.h, gemm_w16.hip): tile 12 = 256x256, tile 13 = 256x320
int ctrlv_gemm_launch_w16(const ctrlv_gemm_desc& d, int tile, hipStream_t stream);
bool ctrlv_gemm_w16_supports(const ctrlv_gemm_desc& d, int tile);
constexpr int kHaloOrderFlag = 0x100;                      // set in the kernel's copy of d.tile (host-side field otherwise)

namespace {

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const ctrlv_gemm_desc d) {
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(A_INSTR >= 1 && B_INSTR >= 1, "tile too small for the wave count");

  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid / WN, wc = wid % WN;
  // Phi table of the GEGLU epilogue (common.h), behind the two stages; the K loop's barriers order it before its use
  if (d.geglu) gelu_table_fill(smem + 2 * STAGE, threadIdx.x, NW * 64);

  const int tiles_n = (d.N + BN - 1) / BN;
  const int tiles_m = (d.M + BM - 1) / BM;
  const int t_id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int bm = (t_id / tiles_n) * BM, bn = (t_id % tiles_n) * BN;

  const int kpt = d.Cin >> 6;             // K-steps per tap
  const int nk = d.taps * kpt;
  const long ktot = (long)d.taps * d.Cin;

  // ---- per-lane gather state for the A rows this lane stages (fixed over the K loop)
  const int prow = lane >> 3, pslot = lane & 7;
  int a_y0[A_INSTR], a_x0[A_INSTR];       // mode 1: top-left input coord (pre-upsample grid is y>>up); mode 2: frame
  long a_base[A_INSTR];                   // mode 0/2: row index m; mode 1: first pixel row of the image
  int a_coff[A_INSTR];                    // logical 16-B chunk (x8 channels) this lane fetches
  bool a_ok[A_INSTR];
#pragma unroll
  for (int q = 0; q < A_INSTR; ++q) {
    const int rt = (q * NW + wid) * 8 + prow;
    const int m = bm + rt;
    a_coff[q] = (pslot ^ ((rt >> 1) & 7)) * 8;
    a_ok[q] = m < d.M;
    a_y0[q] = 0; a_x0[q] = 0; a_base[q] = m;
    if (d.mode == 1) {
      const int hw = d.Ho * d.Wo;
      const int n_img = m / hw, rem = m - n_img * hw;
      const int yo = rem / d.Wo, xo = rem - yo * d.Wo;
      a_y0[q] = yo * d.stride - 1;
      a_x0[q] = xo * d.stride - 1;
      a_base[q] = (long)n_img * d.H * d.Wd;
    } else if (d.mode == 2) {
      a_y0[q] = (m / d.S) % d.F;
    }
  }
  int b_coff[B_INSTR];
  long b_row[B_INSTR];
  bool b_ok[B_INSTR];
#pragma unroll
  for (int q = 0; q < B_INSTR; ++q) {
    const int rt = (q * NW + wid) * 8 + prow;
    const int n = bn + rt;
    b_coff[q] = (pslot ^ ((rt >> 1) & 7)) * 8;
    b_ok[q] = n < d.N;
    b_row[q] = (long)n * ktot;
  }
  const char* zsrc = (const char*)g_ctrlv_zeros + pslot * 16;
  const int hlim = d.H << d.up, wlim = d.Wd << d.up;

  // Stride-1 3x3 convs whose row width divides the ping-pong tile are summed in the order (dy, 32-channel block, dx) by
  // every kernel (gemm_pp_kernel.h conv_halo_geometry: the row-halo kernels stage one slot per (dy, block) for the three
  // dx).  Here a 64-wide K step is two consecutive 32-channel UNITS of that order: a lane's 16-B chunk belongs to unit
  // 2 kt + (chunk >> 2), which fixes its tap and channel offset -- the same sequence of products per output element.
  const bool halo_order = (d.tile & kHaloOrderFlag) != 0;
  const int nb32 = d.Cin >> 5;
  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    if (halo_order) {
#pragma unroll
      for (int q = 0; q < A_INSTR; ++q) {
        const int lc = a_coff[q] >> 3;
        const int u = 2 * kt + (lc >> 2), rest = u / 3, dxi = u - rest * 3, dyi = rest / nb32, cb = rest - dyi * nb32;
        const int yi = a_y0[q] + dyi, xi = a_x0[q] + dxi;
        const bool ok = a_ok[q] && (unsigned)yi < (unsigned)d.H && (unsigned)xi < (unsigned)d.Wd;
        const long row = a_base[q] + (long)yi * d.Wd + xi;
        const char* p = ok ? (const char*)((const el_t*)d.A + row * d.lda + cb * 32 + (lc & 3) * 8) : zsrc;
        __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(sa + (q * NW + wid) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < B_INSTR; ++q) {
        const int lc = b_coff[q] >> 3;
        const int u = 2 * kt + (lc >> 2), rest = u / 3, dxi = u - rest * 3, dyi = rest / nb32, cb = rest - dyi * nb32;
        const long wcol = (long)(dyi * 3 + dxi) * d.Cin + cb * 32 + (lc & 3) * 8;
        const char* p = b_ok[q] ? (const char*)((const el_t*)d.W + b_row[q] + wcol) : zsrc;
        __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(sb + (q * NW + wid) * 1024), 16, 0, 0);
      }
      return;
    }
    // K order: tap outermost (gemm_pp_kernel.h issue_end: the same order, so the same bits; the block-major alternative
    // behind the same macro)
    const int tap = kt / kpt;
    int cc = (kt - tap * kpt) << 6;
    const int wcol = tap * d.Cin + cc;
    const el_t* src = (const el_t*)d.A;
    int ld = d.lda;
    if (d.A2 != nullptr && cc >= d.c_split) {
      src = (const el_t*)d.A2; ld = d.lda2; cc -= d.c_split;
    }
    const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
    for (int q = 0; q < A_INSTR; ++q) {
      bool ok = a_ok[q];
      long row = a_base[q];
      if (d.mode == 1) {
        const int yi = a_y0[q] + dy, xi = a_x0[q] + dx;
        ok = ok && (unsigned)yi < (unsigned)hlim && (unsigned)xi < (unsigned)wlim;
        row += (long)(yi >> d.up) * d.Wd + (xi >> d.up);
      } else if (d.mode == 2) {
        const int ff = a_y0[q] + tap - 1;
        ok = ok && (unsigned)ff < (unsigned)d.F;
        row += (long)(tap - 1) * d.S;
      }
      const char* p = ok ? (const char*)(src + row * ld + cc + a_coff[q]) : zsrc;
      __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(sa + (q * NW + wid) * 1024), 16, 0, 0);
    }
    const el_t* wsrc = (const el_t*)d.W + wcol;
#pragma unroll
    for (int q = 0; q < B_INSTR; ++q) {
      const char* p = b_ok[q] ? (const char*)(wsrc + b_row[q] + b_coff[q]) : zsrc;
      __builtin_amdgcn_global_load_lds(GLB_PTR(p), LDS_PTR(sb + (q * NW + wid) * 1024), 16, 0, 0);
    }
  };

  // fragment read offsets: row (lane&31) of the 32-row sub-tile, 16-B chunk (2*ks + lane>>5) ^ swizzle
  const int r32 = lane & 31, hsel = lane >> 5, sw = (lane >> 1) & 7;
  // accumulators start from the bias (result layout: accumulator e of a 32-column sub-tile is column 8*(e>>2) + 4*hsel
  // + (e&3)) -- the same summation order as the ping-pong kernels, so a layer gives bit-identical results whichever
  // tile serves it (tests: a 2-clip batch and its two 1-clip halves pick different tiles)
  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int col = bn + wc * WTN + j * 32 + 8 * (e >> 2) + 4 * hsel + (e & 3);
      const float bv = (d.bias != nullptr && col < d.N) ? d.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][j][e] = bv;
    }

  const int a_frag_base = (wr * WTM + r32) * 128;
  const int b_frag_base = A_BYTES + (wc * WTN + r32) * 128;

  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const char* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((ks * 2 + hsel) ^ sw) * 16;
      elx8 af[TM], wf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const elx8*)(st + a_frag_base + i * 32 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = *(const elx8*)(st + b_frag_base + j * 32 * 128 + coff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = mfma_32x32x16(wf[j], af[i], acc[i][j]);
    }
  }

  gemm_epilogue<TM, TN>(d, acc, bm, bn, wr, wc, WTM, WTN, r32, hsel, smem + 2 * STAGE);
}

// ---- per-clip rows (the GEMMs of the conditioning path: time / added-id embedding MLPs, the fused time_emb_proj, the
// one-key cross-attention vectors -- ~25 launches per model and step with M = B clips).  On the 128 x 128 MFMA tile such a
// launch is N / 128 workgroups walking K in 64-element steps behind a barrier each: ~30 us of latency for 3 MB of weights.
// Here a wave owns one output column: the weight row streams through its lanes in 16-byte pieces, the activation rows come
// out of L2, one wave reduction per row.  Same epilogue semantics as gemm_epilogue.  Rows are computed independently, in
// chunks of 8 (blockIdx.y): a row's bits are the same at ANY batch size (round 4 chose this kernel by M <= 8, so a clip's
// conditioning vectors -- and with them its whole output -- changed in bits when a ninth clip joined the batch).
__global__ __launch_bounds__(256) void gemv_small_kernel(const ctrlv_gemm_desc d0) {
  const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= d0.N) return;                                 // (whole waves: no barrier below)
  // this workgroup's chunk of rows: re-base the row-indexed operands (row-vector indices keep the absolute row: m0 + m)
  ctrlv_gemm_desc d = d0;
  const int m0 = blockIdx.y * 8;
  d.M = d0.M - m0 < 8 ? d0.M - m0 : 8;
  const el_t* w = (const el_t*)d.W + (long)n * d.Cin;
  const el_t* a = (const el_t*)d.A + (long)m0 * d.lda;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int k8 = lane; k8 < (d.Cin >> 3); k8 += 64) {
    float wf[8];
    unpack_elx8(*(const uint4*)(w + k8 * 8), wf);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < d.M) {
        float af[8];
        unpack_elx8(*(const uint4*)(a + (long)m * d.lda + k8 * 8), af);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[m] = __builtin_fmaf(af[e], wf[e], acc[m]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    if (m >= d.M) break;
    float v = acc[m];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane != 0 || n >= d.n_store) continue;
    v += d.bias ? d.bias[n] : 0.f;
    {
#pragma clang fp contract(off)
      v = v * d.s_acc;
    }
    const int mg = m0 + m;                               // row of the launch
    if (d.R1) v = __builtin_fmaf(d.s1, el_to_f32(((const el_t*)d.R1)[(long)mg * d.ldr1 + n]), v);
    if (d.R2) v = __builtin_fmaf(d.s2, el_to_f32(((const el_t*)d.R2)[(long)mg * d.ldr2 + n]), v);
    if (d.vmode == 1) v += d.V[(long)((mg / d.vdiv) % d.vmod) * d.ldv + n];
    else if (d.vmode == 2) v += d.V[(long)(((long)(mg / d.vdiv) * d.vS + (mg % d.vS)) % d.vmod) * d.ldv + n];
    if (d.act == 1) v = silu_f(v);
    if (d.out_f32 & 1) ((float*)d.out)[(long)mg * d.ldo + n] = v;
    else ((el_t*)d.out)[(long)mg * d.ldo + n] = f32_to_el(v);
  }
}

// ---- split contraction (small images, long K): sum of the K slices' raw accumulators + the launch's epilogue.
// One thread per (row, 4 columns): slices in order, then exactly the operation sequence of gemm_epilogue above.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ctrlv_gemm_desc d, const float* __restrict__ part, int slices) {
  const int n4 = d.n_store >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)d.M * n4) return;
  const int m = (int)(idx / n4), ncol = (int)(idx - (long)m * n4) * 4;
  const long pstride = (long)d.M * d.N;
  const float* p = part + (long)m * d.N + ncol;
  float4 a = *(const float4*)p;
  for (int s = 1; s < slices; ++s) {
    const float4 b = *(const float4*)(p + s * pstride);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  float o[4] = {a.x, a.y, a.z, a.w};
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = o[e] * d.s_acc;
  }
  if (d.R1) {
    const uint2 rv = *(const uint2*)((const el_t*)d.R1 + (long)m * d.ldr1 + ncol);
    o[0] = __builtin_fmaf(d.s1, el_lo_f32(rv.x), o[0]);
    o[1] = __builtin_fmaf(d.s1, el_hi_f32(rv.x), o[1]);
    o[2] = __builtin_fmaf(d.s1, el_lo_f32(rv.y), o[2]);
    o[3] = __builtin_fmaf(d.s1, el_hi_f32(rv.y), o[3]);
    if (d.R1_lo) {
      float rl[4];                  // (lo planes: one byte per element)
      unpack_lo4(*(const unsigned*)((const lo_t*)d.R1_lo + (long)m * d.ldr1 + ncol), rl);
      o[0] = __builtin_fmaf(d.s1, rl[0], o[0]);
      o[1] = __builtin_fmaf(d.s1, rl[1], o[1]);
      o[2] = __builtin_fmaf(d.s1, rl[2], o[2]);
      o[3] = __builtin_fmaf(d.s1, rl[3], o[3]);
    }
  }
  if (d.R2) {
    const uint2 rv = *(const uint2*)((const el_t*)d.R2 + (long)m * d.ldr2 + ncol);
    o[0] = __builtin_fmaf(d.s2, el_lo_f32(rv.x), o[0]);
    o[1] = __builtin_fmaf(d.s2, el_hi_f32(rv.x), o[1]);
    o[2] = __builtin_fmaf(d.s2, el_lo_f32(rv.y), o[2]);
    o[3] = __builtin_fmaf(d.s2, el_hi_f32(rv.y), o[3]);
    if (d.R2_lo) {
      float rl[4];                  // (lo planes: one byte per element)
      unpack_lo4(*(const unsigned*)((const lo_t*)d.R2_lo + (long)m * d.ldr2 + ncol), rl);
      o[0] = __builtin_fmaf(d.s2, rl[0], o[0]);
      o[1] = __builtin_fmaf(d.s2, rl[1], o[1]);
      o[2] = __builtin_fmaf(d.s2, rl[2], o[2]);
      o[3] = __builtin_fmaf(d.s2, rl[3], o[3]);
    }
  }
  if (d.vmode) {
    const long vi = d.vmode == 1 ? (long)((m / d.vdiv) % d.vmod) : (((long)(m / d.vdiv) * d.vS + (m % d.vS)) % d.vmod);
    const float4 vv = *(const float4*)(d.V + vi * d.ldv + ncol);
    o[0] += vv.x; o[1] += vv.y; o[2] += vv.z; o[3] += vv.w;
  }
  const uint2 pk = make_uint2(pack_elx2(o[0], o[1]), pack_elx2(o[2], o[3]));
  *(uint2*)((el_t*)d.out + (long)m * d.ldo + ncol) = pk;
  if (d.out_lo) {
#pragma clang fp contract(off)
    const float l0 = o[0] - el_lo_f32(pk.x), l1 = o[1] - el_hi_f32(pk.x), l2 = o[2] - el_lo_f32(pk.y), l3 = o[3] - el_hi_f32(pk.y);
    *(unsigned*)((lo_t*)d.out_lo + (long)m * d.ldo + ncol) = pack_lo4(l0, l1, l2, l3);
  }
}

// Split plan of a launch: number of K slices (1 = none) and the ping-pong tile that runs them.  The 3x3 / temporal convs
// of the SMALL levels (<= 256 pixels per image: 9 x 16 at the benchmark's size; 10 x 16 and 5 x 8 at the reference's
// default 320 x 512) have few output tiles -- 40-160 for a 50-image batch on 256 CUs -- and a contraction of 3840-23040:
// the chip ran them at 0.10-0.38 of its rate.  K is cut into channel ranges (every tap of a range), all slices run in ONE
// launch (gemm_pp_kernel.h, KS) and a streaming kernel adds them and applies the epilogue.  A function of the LAYER's shape
// only (pixels per image, N, Cin, taps) -- priced for the 50 frame-images of a CFG'd 25-frame clip, never the actual batch:
// a clip takes the same path, with the same summation order, alone and in any batch.
static int splitk_plan(const ctrlv_gemm_desc& d, int* tile_out) {
  if (!ctrlv_debug().splitk || (d.mode == 0 && d.S <= 0) || d.geglu || d.A2 || d.act || d.out_f32 || d.raw_out || d.gn_partials || d.n_scale2 || d.tile) return 1;
  if (d.N % 32 || d.N < 256 || d.n_store != d.N || d.ldo % 8 || (d.R1 && d.ldr1 % 8) || (d.R2 && d.ldr2 % 8) ||
      (d.vmode && d.ldv % 8) || d.Cin % 64)
    return 1;
  const long S = d.mode == 1 ? (long)d.Ho * d.Wo : (long)d.S;
  if (S <= 0 || S > 256 || d.M % S != 0) return 1;
  if (d.mode == 1 && ctrlv_conv_halo_order(d)) return 1;
  const int nb = d.Cin / 64;
  const long tiles_m = (50 * S + 255) / 256;
  // one half-step of a 256-wide tile (measured: 222 us for the 360 half-steps of the 7200 x 1280 x 11520 conv, 0.74 us per
  // half-step on the two-round 28800-row one); streaming rate of the partial sums (written once, read once)
  const double us_unit = 0.65, bytes_per_us = 4.0e6;
  auto cost = [&](int bn, int s) {
    const long items = tiles_m * ((d.N + bn - 1) / bn) * s;
    const long rounds = (items + 255) / 256;
    const double gemm_us = (double)rounds * (d.taps * (d.Cin / s) / 32) * (bn / 256.0) * us_unit;
    return gemm_us + (s > 1 ? (double)s * 50.0 * S * d.N * 8.0 / bytes_per_us : 0.0);
  };
  const double base = cost(d.N % 320 == 0 && 50 * S >= 16384 ? 320 : 256, 1);
  double best = base * 0.85;             // a split has to be worth its second launch
  int best_s = 1, best_tile = 0;
  for (int bn = 256; bn <= 320; bn += 64) {
    if (bn == 320 && d.N % 320 != 0) continue;
    for (int sl = 2; sl <= nb && sl <= 16; ++sl) {
      if (nb % sl != 0 || d.taps * (d.Cin / sl) / 32 < 24) continue;
      const double c = cost(bn, sl);
      if (c < best) { best = c; best_s = sl; best_tile = bn == 320 ? 6 : 5; }
    }
  }
  *tile_out = best_tile;
  return best_s;
}

// Which LAYERS run on the 16x16x32 core (0 = none): a function of the layer's shape only -- never of M -- because the core
// sums a K half-step in one instruction whose internal order is not the 32x32x16 kernels': a layer given to it runs there at
// EVERY row count, so a clip's bits do not depend on the batch (tests/test_fullsize_gpu.py clip independence).
static int w16_tile_of(const ctrlv_gemm_desc& d) {
  if (!ctrlv_debug().w16 || d.mode != 0) return 0;
  int tile = 0;
  // in-model A/B (tools/shape_table.py, two alternations on one device, profiles/r05_w16_in_model_ab.txt): the C = 1280 GEGLU
  // projections 14.2 -> 13.35 ms per step (-5.7 %), the C = 640 one 15.97 -> 16.48 ms (+3.2 %: stays on the ping-pong tile)
  if (d.geglu) tile = (d.Cin >= 1280 && d.N % 256 == 0) ? 12 : 0;
  if (tile && !ctrlv_gemm_w16_supports(d, tile)) tile = 0;           // (raw_out, odd pitches ...: shape-level conditions too)
  return tile;
}

template <int BM, int BN, int WM, int WN>
int launch(const ctrlv_gemm_desc& d, hipStream_t stream) {
  constexpr int smem = 2 * (BM + BN) * 128 + kGeluTabBytes;   // staging ring | Phi table (GEGLU launches)
  static bool attr_set[CTRLV_MAX_DEVICES] = {};      // per device: the attribute belongs to the device's code object
  auto kfn = gemm_kernel<BM, BN, WM, WN>;
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[dev] = true;
  }
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  hipLaunchKernelGGL(kfn, dim3(tiles), dim3(WM * WN * 64), smem, stream, d);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

}  // namespace

extern "C" size_t ctrlv_gemm_splitk_ws_bytes(const ctrlv_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0) return 0;
  int tile = 0;
  const int s = splitk_plan(*d, &tile);
  return s > 1 ? (size_t)s * d->M * d->N * sizeof(float) : 0;
}

extern "C" int ctrlv_gemm(const ctrlv_gemm_desc* dp, ctrlv_stream_t stream_) {
  CTRLV_CHECK_ARG(dp != nullptr, "ctrlv_gemm: null descriptor");
  ctrlv_gemm_desc d = *dp;
  hipStream_t stream = (hipStream_t)stream_;
  CTRLV_CHECK_ARG(d.A && d.W && d.out, "ctrlv_gemm: A, W and out must be non-null");
  CTRLV_CHECK_SHAPE(d.M > 0 && d.N > 0 && d.Cin > 0, "ctrlv_gemm: M, N, Cin must be positive (M=%d N=%d Cin=%d)", d.M,
                    d.N, d.Cin);
  CTRLV_CHECK_SHAPE(d.Cin % 64 == 0, "ctrlv_gemm: Cin=%d must be a multiple of 64 (pad the weight / im2col)", d.Cin);
  CTRLV_CHECK_SHAPE(d.N % 32 == 0, "ctrlv_gemm: N=%d must be a multiple of 32 (pad the weight rows)", d.N);
  CTRLV_CHECK_SHAPE(d.lda % 8 == 0 && (d.A2 == nullptr || (d.lda2 % 8 == 0 && d.c_split % 64 == 0)),
                    "ctrlv_gemm: lda/lda2 must be multiples of 8 and c_split a multiple of 64");
  CTRLV_CHECK_SHAPE(d.ldo % 4 == 0 && d.n_store % 4 == 0, "ctrlv_gemm: ldo and n_store must be multiples of 4");
  if (d.mode == 0) {
    CTRLV_CHECK_SHAPE(d.taps == 1, "ctrlv_gemm: mode 0 needs taps == 1");
  } else if (d.mode == 1) {
    CTRLV_CHECK_SHAPE(d.taps == 9 && d.H > 0 && d.Wd > 0 && d.Ho > 0 && d.Wo > 0 && (d.stride == 1 || d.stride == 2) &&
                          (d.up == 0 || d.up == 1),
                      "ctrlv_gemm: bad conv2d geometry");
    CTRLV_CHECK_SHAPE(d.M % (d.Ho * d.Wo) == 0, "ctrlv_gemm: M must be a multiple of Ho*Wo in conv2d mode");
    CTRLV_CHECK_SHAPE(((d.H << d.up) + 2 - 3) / d.stride + 1 == d.Ho && ((d.Wd << d.up) + 2 - 3) / d.stride + 1 == d.Wo,
                      "ctrlv_gemm: conv2d output size %dx%d inconsistent with input %dx%d stride %d up %d", d.Ho, d.Wo,
                      d.H, d.Wd, d.stride, d.up);
  } else if (d.mode == 2) {
    CTRLV_CHECK_SHAPE(d.taps == 3 && d.F > 0 && d.S > 0 && d.M % (d.F * d.S) == 0, "ctrlv_gemm: bad temporal geometry");
  } else {
    CTRLV_CHECK_ARG(false, "ctrlv_gemm: unknown mode %d", d.mode);
  }
  CTRLV_CHECK_ARG(d.vmode >= 0 && d.vmode <= 2, "ctrlv_gemm: bad vmode");
  if (d.vmode) CTRLV_CHECK_ARG(d.V && d.vdiv > 0 && d.vmod > 0 && d.ldv % 4 == 0 && (d.vmode == 1 || d.vS > 0), "ctrlv_gemm: bad row-vector table");
  CTRLV_CHECK_SHAPE(d.n_scale2 >= 0 && d.n_scale2 % 32 == 0 && (d.n_scale2 == 0 || !d.geglu),
                    "ctrlv_gemm: n_scale2=%d must be a non-negative multiple of 32 (and 0 with GEGLU)", d.n_scale2);
  if (d.geglu) {
    CTRLV_CHECK_SHAPE(d.N % 32 == 0, "ctrlv_gemm: GEGLU needs N %% 32 == 0 (16 value + 16 gate columns per sub-tile)");
    CTRLV_CHECK_ARG(!d.R1 && !d.R2 && !d.vmode && !d.act && !(d.out_f32 & 1) && d.mode == 0,
                    "ctrlv_gemm: GEGLU epilogue takes bias only (mode 0)");
  }
  const bool split_io = d.R1_lo || d.R2_lo || d.out_lo;
  if (split_io) {         // SPLIT residual-trunk planes (include/ctrlv_hip.h)
    CTRLV_CHECK_ARG(CTRLV_ELEM_DTYPE == 1, "ctrlv_gemm: R1_lo / R2_lo / out_lo (split trunk) are served by the fp16 element "
                                           "library only");
    CTRLV_CHECK_ARG((!d.R1_lo || d.R1) && (!d.R2_lo || d.R2), "ctrlv_gemm: R1_lo / R2_lo need R1 / R2");
    CTRLV_CHECK_ARG(!d.geglu && !d.act && !d.out_f32 && !d.raw_out && !d.n_scale2,
                    "ctrlv_gemm: split planes do not combine with GEGLU / SiLU / fp32 output / raw_out / n_scale2");
  }
  if (d.splitk_ws) {      // split contraction where the layer's shape calls for it (splitk_plan)
    int tile_s = 0;
    const int slices = splitk_plan(d, &tile_s);
    if (slices > 1) {
      ctrlv_gemm_desc dd = d;
      dd.Cin = d.Cin / slices; dd.w_cin = d.Cin; dd.ksplit = slices;
      dd.out = d.splitk_ws; dd.ldo = d.N; dd.n_store = d.N;
      dd.R1 = dd.R2 = nullptr; dd.V = nullptr; dd.vmode = 0; dd.s_acc = 1.0f; dd.splitk_ws = nullptr; dd.tile = 0;
      dd.R1_lo = dd.R2_lo = nullptr; dd.out_lo = nullptr;
      if (ctrlv_gemm_pp_supports(dd) && (long)d.M * d.N * 4 < 0x7FFFFFF0L) {
        int rc = ctrlv_gemm_launch_pp(dd, tile_s, stream);
        if (rc != CTRLV_OK) return rc;
        const long n_thr = (long)d.M * (d.n_store >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, stream, d,
                           (const float*)d.splitk_ws, slices);
        CTRLV_LAUNCH_CHECK();
        return CTRLV_OK;
      }
    }
  }
  // Per-clip rows: tile 11 = "this is a per-clip GEMM".  The plans say so for every GEMM of the conditioning path: the choice
  // is the LAYER's, never the row count's (no launch is routed here by its M: a layer's bits must not change with the batch).
  if (d.tile == 11) {
    CTRLV_CHECK_ARG(d.mode == 0 && !d.geglu && !d.A2 && !d.raw_out && d.n_scale2 == 0 && (d.out_f32 & ~1) == 0 && !split_io,
                    "ctrlv_gemm: tile 11 (per-clip rows) serves plain mode-0 launches only");
    hipLaunchKernelGGL(gemv_small_kernel, dim3((unsigned)((d.N + 3) / 4), (unsigned)((d.M + 7) / 8)), dim3(256), 0, stream, d);
    CTRLV_LAUNCH_CHECK();
    return CTRLV_OK;
  }
  if (d.tile == 12 || d.tile == 13) return ctrlv_gemm_launch_w16(d, d.tile, stream);
  if (d.tile == 0) {
    const int w16 = w16_tile_of(d);
    if (w16) return ctrlv_gemm_launch_w16(d, w16, stream);
  }
  int tile = d.tile;
  if (tile == 0) {
    // measured on MI355X (tools/gemm_sweep.py, profiles/r01_gemm_sweep.txt): the persistent ping-pong tiles win on
    // every layer shape of the UNet; 256x320 where N is a multiple of 320 (C = 320 / 640 levels, conv / FF-out at
    // 1280, GEGLU at K < 1280), 256x256 otherwise (N = 3840 qkv, K = 1280 GEGLU: +7 %); tiny-M per-clip GEMMs stay on
    // 128x128.
    const bool big = d.M >= 1024 && d.N >= 128;
    if (!big) tile = 1;
    else if (d.N <= 128 && d.mode != 0) tile = 10;     // 256x128 ping-pong tile (profiles/r02_vae_decode.txt)
    else if (d.N % 320 == 0 && (d.geglu ? d.Cin < 1280 : d.N < 3840) && d.M >= 16384) tile = 6;
    else tile = 5;
    // split planes: the 256x320 ping-pong tile is the one instantiated with the split epilogue (N = 320 / 640 / 1280: the
    // trunk's widths), the 2-stage kernel serves everything else -- the same operation sequence in both
    if (split_io && tile >= 5) tile = d.N % 320 == 0 ? 6 : 1;
    {
      const int force = ctrlv_debug().force_tile;     // (tools/shape_table.py: tile 5 / 6 for every large launch)
      if ((force == 5 || force == 6) && tile >= 5 && tile <= 6) tile = force;
    }
  }
  if (d.gn_partials) {  // producer-side GroupNorm statistics: the 256x320 ping-pong tile, whatever M is
    CTRLV_CHECK_SHAPE(ctrlv_gemm_gn_partials_serves(&d), "ctrlv_gemm: gn_partials is not served for this launch (ask "
                                                         "ctrlv_gemm_gn_partials_serves first)");
    tile = 6;
  }
  if (d.raw_out) {      // second output of the GEGLU projection (training forward): ping-pong tiles only
    CTRLV_CHECK_ARG(d.geglu && d.ld_raw >= d.N, "ctrlv_gemm: raw_out needs geglu = 1 and ld_raw >= N");
    if (tile < 5 || tile > 8) tile = d.N % 320 == 0 ? 6 : 5;   // (tile 10 does not write raw_out)
    CTRLV_CHECK_SHAPE(ctrlv_gemm_pp_supports(d), "ctrlv_gemm: raw_out needs a shape the ping-pong tiles serve (K >= 128, "
                                                 "ld_raw a multiple of 8)");
  }
  if (tile >= 5) {
    // the ping-pong kernels' epilogue moves 8 columns (16 B of bf16) per lane: needs 8-element granularity
    const bool wide_ok = d.n_store % 8 == 0 && d.ldo % 8 == 0 && (!d.R1 || d.ldr1 % 8 == 0) &&
                         (!d.R2 || d.ldr2 % 8 == 0) && (!d.vmode || d.ldv % 8 == 0);
    if (!wide_ok || !ctrlv_gemm_pp_supports(d)) {
      // (the 2-stage kernels do not write raw_out: falling back would leave the caller's tensor uninitialised)
      CTRLV_CHECK_SHAPE(!d.raw_out, "ctrlv_gemm: raw_out needs n_store / ldo to be multiples of 8 (ping-pong tiles only)");
      CTRLV_CHECK_SHAPE(d.tile == 0, "ctrlv_gemm: tiles 5-8 need n_store / ldo / ldr / ldv to be multiples of 8 and an "
                                     "epilogue of {bias, V, R1, R1+V, R1+R2} without SiLU / fp32 output");
      tile = 1;
    }
  }
  d.tile = ctrlv_conv_halo_order(d) ? kHaloOrderFlag : 0;     // (from here on d.tile only carries the K-order flag)
  switch (tile) {
    case 1: return launch<128, 128, 2, 2>(d, stream);
    case 2: return launch<256, 256, 2, 4>(d, stream);
    case 3: return launch<256, 64, 4, 1>(d, stream);
    case 4: return launch<256, 128, 4, 2>(d, stream);
    case 5:
    case 7: return ctrlv_gemm_launch_pp(d, tile, stream);
    case 6:
    case 8: return ctrlv_gemm_launch_pp(d, tile, stream);
    case 10: return ctrlv_gemm_launch_pp(d, tile, stream);
    default: CTRLV_CHECK_ARG(false, "ctrlv_gemm: unknown tile %d", tile);
  }
  return CTRLV_OK;
}
