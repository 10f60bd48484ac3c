// Self-attention cores for gfx950 (head_dim 64, bf16 in/out, fp32 softmax + accumulation).
//
// Both kernels compute S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 (K tile = A operand, Q = B operand) so that each
// lane owns ONE query row and the 16 accumulator registers of a 32-key tile are that row's scores: the online
// softmax is in-register (one v_permlane32_swap to merge the two half-waves) and the exponentiated tile is, after a
// bf16 pack, directly the B operand of O^T += V^T . P (guide section 3 "accumulator tile as the next MFMA's operand").
// V^T fragments come from a row-major V tile in LDS through ds_read_b64_tr_b16 (hardware transpose).
// K/V tiles are staged by LDS-DMA (global_load_lds_dwordx4) with the bank-conflict swizzle on the source address.
//
//   spatial : 128 queries x 64-key tiles per workgroup (4 waves x 32 rows), 2- or 3-slot K/V ring, S up to 9216.
//   temporal: one wave per (clip, pixel, head): 25 frames padded to one 32x32 tile; the (b f) s c <-> (b s) f c
//             permutes of TemporalBasicTransformerBlock are row-stride arithmetic (stride S*3C between frames).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr float kScaleLog2 = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) * log2(e)

__device__ __forceinline__ float half_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// V^T fragment (A operand of O^T += V^T.P) for 16 keys starting at `kbase` (+4 for the upper half-wave, folded
// into voff by the caller) and 32 d-columns: two transposed 4x16 block reads.
__device__ __forceinline__ elx8 vt_frag(const char* vt, int voff_lo, int voff_hi) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vt + voff_lo));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vt + voff_hi));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(elx8, v);
}

// byte offset inside a row-major [keys][64] bf16 V tile (128-B rows) of (key, d) with the tr-read swizzle:
// 16-B chunk index ^= ((key>>1)&1)<<2  (keeps the 4 rows of a transposed block on distinct banks)
__device__ __forceinline__ int v_off(int key, int d) {
  const int chunk = (d >> 3) ^ (((key >> 1) & 1) << 2);
  return key * 128 + chunk * 16 + (d & 7) * 2;
}

// D = A.B + C with D and C in DIFFERENT registers (C stays live).  hipcc selects the accumulate-in-place form of the
// MFMA and copies C into D first (16 v_mov_b64 per 64-key tile when C is the kept -m block of attn_spatial64_kernel),
// so this one instruction is written out.  Hazards: A / B / C are not written inside the statement; the consumer of D is
// the next MFMA of the same accumulation chain (C operand, same registers: no wait states needed).
__device__ __forceinline__ f32x16 mfma_keep_c(const elx8& a, const elx8& b, const f32x16& c) {
  f32x16 d;
  asm(CTRLV_MFMA_32x32x16_ASM " %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ elx8 pack_p(const f32x16& p, int s) {
  elx8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (el_native_t)p[8 * s + j];
  return r;
}

// ---------------------------------------------------------------------------------------------- spatial
// launch_bounds(256, 2): a 256-register budget makes hipcc keep the score / output accumulators in arch VGPRs; with the
// default budget it parks them in AGPRs and spends 159 v_accvgpr_read/write per 64-key tile to feed the softmax VALU.
// PRE: q arrives pre-scaled by (1/8) log2(e) (see attn_spatial64_kernel); here that only changes the constant.
template <int NSLOT, bool PRE>
__global__ __launch_bounds__(256, NSLOT == 2 ? 4 : 3) void attn_spatial_kernel(const el_t* __restrict__ qkv, el_t* __restrict__ out,
                                                           float* __restrict__ lse, int S, int C) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 3 x (K 8 KiB | V 8 KiB) ring
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5, sw = (lane >> 1) & 7;
  const int head = blockIdx.y, img = blockIdx.z;
  const long row0 = (long)img * S;
  const int ld = 3 * C;
  const el_t* qp = qkv + head * 64;
  constexpr float kScale = PRE ? 1.0f : kScaleLog2;

  const int qrow = blockIdx.x * 128 + wid * 32 + r32;
  elx8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qrow < S) v = *(const uint4*)(qp + (row0 + qrow) * ld + 16 * ks + 8 * hsel);
    qf[ks] = __builtin_bit_cast(elx8, v);
  }

  // K/V tiles are gathered by LDS-DMA through a buffer descriptor over THIS image's rows: the per-lane byte offsets
  // are computed once, a tile only adds a scalar offset, and keys >= S fall outside num_records (the hardware range
  // check returns zeros) -- no per-tile address arithmetic or bounds selects on the VALU, which is this kernel's
  // critical resource (softmax: ~2 VALU slots per MFMA cycle at head_dim 64).
  const int prow = lane >> 3, pslot = lane & 7;
  const __amdgpu_buffer_rsrc_t rs_kv =
      __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + row0 * ld), 0, (int)((long)S * ld * 2), 0x00020000);
  // piece q of a wave covers tile rows (q*4 + wid)*8 + prow: q only adds 32 rows (a scalar offset; the swizzle terms
  // depend on (row >> 1) & 7 and are unchanged), so one K and one V offset per lane serve the whole kernel
  const int rt0 = wid * 8 + prow;
  const unsigned koff = (unsigned)(rt0 * ld + C + head * 64 + (pslot ^ ((rt0 >> 1) & 7)) * 8) * 2u;
  const unsigned voff = (unsigned)(rt0 * ld + 2 * C + head * 64 + (pslot ^ (((rt0 >> 1) & 1) << 2)) * 8) * 2u;
  const int tile_bytes = 64 * ld * 2;
  const int full_tiles = S / 64;
  auto issue = [&](int t, int stage) {
    char* ks_ = smem + stage * 16384;
    char* vs_ = ks_ + 8192;
    // only the per-lane offset is range-checked: full tiles pass the tile offset as a scalar, the ragged last tile adds
    // it to the lane offset so that keys >= S fall past num_records (zeros) instead of reading the next image
    const bool ragged = t >= full_tiles;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int so = t * tile_bytes + q * (tile_bytes >> 1);
      const unsigned ko = ragged ? koff + (unsigned)so : koff, vo = ragged ? voff + (unsigned)so : voff;
      const int sso = ragged ? 0 : so;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(ks_ + (q * 4 + wid) * 1024), 16, ko, sso, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(vs_ + (q * 4 + wid) * 1024), 16, vo, sso, 0, 0);
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // per-lane V^T read offsets (relative to a 16-key step base): row (i>>2) (+4 for upper half), col 16*((lane>>4)&1)+4*(i&3)
  const int i16 = lane & 15;
  const int vkey = 4 * hsel + (i16 >> 2);
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);

  // Online softmax without a per-element running-max pass.  A tile is exponentiated against the max KEPT from
  // earlier tiles (packed fma + v_exp_f32) and summed; only if some lane's partial row sum exceeds 2^12 (or is
  // inf / NaN: always on the first tile, m_run = -inf) the tile takes the slow path: scores are recomputed from the
  // K tile still in LDS, the true row max is folded into m_run, O and l are rescaled once, and the tile is
  // exponentiated again.  P stays <= 2^12 (exact in bf16's fp32 exponent range, fp32 accumulation), and the common
  // path drops the 24 v_max3/v_max and the compare of the classic deferred-rescale scheme (T13) from a loop that is
  // VALU-bound: measured on MI355X v_exp_f32 costs two VALU issue slots, the kernel spends ~1.6 slots per MFMA cycle.
  constexpr float kSumLimit = 4096.0f;
  auto scores = [&](const char* kst, f32x16 (&sacc)[2], int t, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const elx8 kf = *(const elx8*)(kst + (kt * 32 + r32) * 128 + (((ks * 2 + hsel) ^ sw) * 16));
        sacc[kt] = mfma_32x32x16(kf, qf[ks], sacc[kt]);
      }
      // keep the four K fragments of the second 32-key half out of flight until the first half is consumed: the
      // kernel has to fit 128 VGPRs (4 waves per SIMD), and hoisting all eight costs 16 registers
      if (kt == 0) asm volatile("" ::: "memory");
    }
    if (MASKED) {  // key masking on the ragged last tile only (separate instantiation: no selects in the main loop)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = t * 64 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hsel;
          if (key >= S) sacc[kt][e] = -INFINITY;
        }
    }
  };
  auto exp_sum = [&](f32x16 (&sacc)[2]) -> float {
    f32x2_t rs2 = {0.f, 0.f};
    const f32x2_t nm = {-m_run, -m_run};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        f32x2_t x = {sacc[kt][e], sacc[kt][e + 1]};
        x = x * kScale + nm;
        f32x2_t pe = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
        sacc[kt][e] = pe.x;
        sacc[kt][e + 1] = pe.y;
        rs2 += pe;
      }
    return rs2.x + rs2.y;
  };
  auto tile = [&](int t, auto masked_tag) {
    const char* kst = smem + (t % NSLOT) * 16384;
    const char* vst = kst + 8192;
    f32x16 sacc[2];
    scores(kst, sacc, t, masked_tag);
    float rs = exp_sum(sacc);
    if (!__all(rs <= kSumLimit)) {
      scores(kst, sacc, t, masked_tag);
      float mx = sacc[0][0];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[kt][e]);
      mx = half_max(mx) * kScale;
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
      rs = exp_sum(sacc);
    }
    l_run += rs;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const elx8 pf = pack_p(sacc[kt], s);
        const int kb = kt * 32 + 16 * s + vkey;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const elx8 vf = vt_frag(vst, v_off(kb, dt * 32 + vcol), v_off(kb + 8, dt * 32 + vcol));
          oacc[dt] = mfma_32x32x16(vf, pf, oacc[dt]);
        }
      }
    }
  };

  // 3-slot K/V ring, two tiles of LDS-DMA in flight: every wave retires its own pieces of tile t with a COUNTED
  // vmcnt (the 4 pieces of tile t+1 may stay outstanding), a raw s_barrier (no implicit vmcnt(0) drain) makes all
  // waves' pieces visible and proves slot (t+2)%3 == (t-1)%3 is no longer being read, then tile t+2 is issued.
  const int nt = (S + 63) / 64;
  const int nt_full = S / 64;
  issue(0, 0);
  if (NSLOT == 3 && nt > 1) issue(1, 1);
  for (int t = 0; t < nt_full; ++t) {
    if (NSLOT == 3 && t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (t + NSLOT - 1 < nt) issue(t + NSLOT - 1, (t + NSLOT - 1) % NSLOT);
    tile(t, std::false_type{});
  }
  if (nt_full < nt) {      // ragged last tile: separate instantiation with key masking
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    tile(nt_full, std::true_type{});
  }

  const float l_tot = half_sum(l_run);
  const float inv = 1.0f / l_tot;
  // training forward: L = m + log2(l) per row (log2 domain of the scaled scores), [img][head][S] fp32 -- the backward
  // kernels rebuild P = exp2(s * scale * log2e - L) from it
  if (lse && qrow < S && hsel == 0) lse[((long)img * gridDim.y + head) * S + qrow] = m_run + __builtin_amdgcn_logf(l_tot);
  if (qrow < S) {
    el_t* op = out + (row0 + qrow) * C + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dcol = dt * 32 + 8 * q + 4 * hsel;
        uint2 pk = make_uint2(pack_elx2(oacc[dt][4 * q] * inv, oacc[dt][4 * q + 1] * inv),
                              pack_elx2(oacc[dt][4 * q + 2] * inv, oacc[dt][4 * q + 3] * inv));
        *(uint2*)(op + dcol) = pk;
      }
  }
}

// ---------------------------------------------------------------------------------------------- spatial, 64 rows / wave
// Same algorithm with TWO 32-row query blocks per wave (256 queries per workgroup): every K / V^T fragment read from
// LDS feeds two MFMAs, the per-tile loop / DMA / barrier overhead is shared by 32 MFMAs instead of 16, and a wave
// carries four independent accumulator chains.  The spatial kernel is bound by instruction issue around the MFMAs
// (SQ counters: MFMA busy 36 %, LDS and VMEM waits negligible; no-traffic diagnostic build only +10 %), so fewer
// instructions per MFMA is what moves it.  2 waves per SIMD.
//
// PRE (round 3): the caller's q columns are PRE-SCALED by (1/sqrt(64)) * log2(e) -- the fused q|k|v projection applies
// that factor to its q column block in its fp32 epilogue, before the one bf16 rounding (ctrlv_gemm_desc.s_acc2), so the
// scaled scores come out of the K.Q^T MFMAs directly -- and the running max is SUBTRACTED BY THE MATRIX PIPE: each row
// block keeps 16 registers holding -m (all equal; a lane owns one query row, so its 16 accumulators of a 32-key block
// share one max) and the first MFMA of a score chain takes them as its C operand.  The exponent's argument is then the
// accumulator itself: the per-score v_fma (64 of the ~240 VALU instructions of a 64-key tile, ~18 % of the loop's issue
// slots) is gone.  m only changes on the slow path (see attn_spatial_kernel), which rewrites the 16 registers.
// The other experiments of round 2 (skewed row blocks, anti-phase start delay, one wave per SIMD) are recorded in
// DESIGN.md section 8 and tools/experiments/.
// K / V ring of the 64-row kernel: kNS64 slots of (K 8 KiB | V 8 KiB), kNS64 - 1 tiles in flight.  Alone (K / V out of
// the Infinity Cache) two slots were enough; inside the model they come from HBM beside the other stream's traffic and
// one tile of 64 keys (~0.7 us of compute) no longer covers the latency.  A/B handle: -DCTRLV_ATTN_SLOTS=2.
#ifndef CTRLV_ATTN_SLOTS
#define CTRLV_ATTN_SLOTS 4
#endif
constexpr int kNS64 = CTRLV_ATTN_SLOTS;
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
static_assert(kNS64 >= 2 && kNS64 <= 4, "2 workgroups per CU: at most 4 x 16 KiB per workgroup");

template <bool PRE>
__global__ __launch_bounds__(256, 2) void attn_spatial64_kernel(const el_t* __restrict__ qkv, el_t* __restrict__ out,
                                                              float* __restrict__ lse, int S, int C) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 2 x (K 8 KiB | V 8 KiB) ring
  CTRLV_CLOCK_BEGIN();
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5, sw = (lane >> 1) & 7;
  // XCD-aware order (common.h xcd_remap): the query blocks of one (image, head) run on ONE XCD, one after the other, so
  // its K / V rows (2.4 MB at S = 9216) are fetched into one L2 instead of all eight (the hardware deals consecutive
  // workgroups out to the XCDs round-robin).  Repeated launches of this kernel alone are 1-4 % slower with it (K / V then
  // come out of the Infinity Cache either way); inside the model, where they come from HBM beside the other stream's
  // traffic, the family is 1.5-2 % faster (42.7 -> 42.0 ms per step, three alternations).  -DCTRLV_ATTN_NO_XCD: old order.
  const int nqb = gridDim.x, nhd = gridDim.y;
  const int wg = xcd_remap((int)(blockIdx.x + nqb * (blockIdx.y + nhd * blockIdx.z)), nqb * nhd * (int)gridDim.z);
  const int qblk = wg % nqb, head = (wg / nqb) % nhd, img = wg / (nqb * nhd);
  const long row0 = (long)img * S;
  const int ld = 3 * C;
  const el_t* qp = qkv + head * 64;
  constexpr float kScale = PRE ? 1.0f : kScaleLog2;      // what is left to apply to a raw score

  int qrow[2];
  elx8 qf[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    qrow[rb] = qblk * 256 + wid * 64 + rb * 32 + r32;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qrow[rb] < S) v = *(const uint4*)(qp + (row0 + qrow[rb]) * ld + 16 * ks + 8 * hsel);
      qf[rb][ks] = __builtin_bit_cast(elx8, v);
    }
  }

  const int prow = lane >> 3, pslot = lane & 7;
  const __amdgpu_buffer_rsrc_t rs_kv =
      __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + row0 * ld), 0, (int)((long)S * ld * 2), 0x00020000);
  const int rt0 = wid * 8 + prow;
  const unsigned koff = (unsigned)(rt0 * ld + C + head * 64 + (pslot ^ ((rt0 >> 1) & 7)) * 8) * 2u;
  const unsigned voff = (unsigned)(rt0 * ld + 2 * C + head * 64 + (pslot ^ (((rt0 >> 1) & 1) << 2)) * 8) * 2u;
  const int tile_bytes = 64 * ld * 2;
  const int full_tiles = S / 64;
  auto issue = [&](int t, int stage) {
    char* ks_ = smem + stage * 16384;
    char* vs_ = ks_ + 8192;
    // only the per-lane offset is range-checked: full tiles pass the tile offset as a scalar, the ragged last tile adds
    // it to the lane offset so that keys >= S fall past num_records (zeros) instead of reading the next image
    const bool ragged = t >= full_tiles;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int so = t * tile_bytes + q * (tile_bytes >> 1);
      const unsigned ko = ragged ? koff + (unsigned)so : koff, vo = ragged ? voff + (unsigned)so : voff;
      const int sso = ragged ? 0 : so;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(ks_ + (q * 4 + wid) * 1024), 16, ko, sso, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(vs_ + (q * 4 + wid) * 1024), 16, vo, sso, 0, 0);
    }
  };

  f32x16 oacc[2][2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[rb][dt][e] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY};
  float l_run[2] = {0.f, 0.f};
  // PRE: -m of each row block as the C operand of its score chains (+inf before the first tile: the first tile's sums
  // are inf and take the slow path, exactly as with m = -inf in the subtracting form)
  f32x16 negm[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int e = 0; e < 16; ++e) negm[rb][e] = INFINITY;

  const int i16 = lane & 15;
  const int vkey = 4 * hsel + (i16 >> 2);
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);

  constexpr float kSumLimit = 4096.0f;
  // scores of one 64-key tile for both row blocks.  SUB (PRE only): start the chains from -m instead of zero.
  auto scores = [&](const char* kst, f32x16 (&sacc)[2][2], int t, auto masked_tag, auto sub_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    constexpr bool SUB = decltype(sub_tag)::value;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (!SUB) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int e = 0; e < 16; ++e) sacc[rb][kt][e] = 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const elx8 kf = *(const elx8*)(kst + (kt * 32 + r32) * 128 + (((ks * 2 + hsel) ^ sw) * 16));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          if (SUB && ks == 0) sacc[rb][kt] = mfma_keep_c(kf, qf[rb][ks], negm[rb]);
          else sacc[rb][kt] = mfma_32x32x16(kf, qf[rb][ks], sacc[rb][kt]);
      }
    }
    if (MASKED) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int key = t * 64 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hsel;
            if (key >= S) sacc[rb][kt][e] = -INFINITY;
          }
    }
  };
  // SCALAR fp32 on purpose (this file is built with -fno-slp-vectorize): beside MFMAs a v_pk_add_f32 / v_pk_fma_f32 costs
  // about four issue slots, two plain v_fma_f32 cost two (MI355X_MICROARCH.md, "price of one filler beside MFMAs").
  // DIRECT: the accumulators already hold scale * s - m (PRE fast path).  The exponentials are packed to bf16 at once
  // (the B operands of O^T += V^T.P), so the 64 score registers of a tile die here and not at the end of the P.V phase.
  auto exp_pack = [&](const f32x16 (&sacc)[2], elx8 (&pf)[2][2], float m, auto direct_tag) -> float {
    constexpr bool DIRECT = decltype(direct_tag)::value;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    const float nm = -m;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        float p0, p1, p2, p3;
        if (DIRECT) {
          p0 = __builtin_amdgcn_exp2f(sacc[kt][e]);
          p1 = __builtin_amdgcn_exp2f(sacc[kt][e + 1]);
          p2 = __builtin_amdgcn_exp2f(sacc[kt][e + 2]);
          p3 = __builtin_amdgcn_exp2f(sacc[kt][e + 3]);
        } else {
          p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e], kScale, nm));
          p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e + 1], kScale, nm));
          p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e + 2], kScale, nm));
          p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e + 3], kScale, nm));
        }
        elx8& d = pf[kt][e >> 3];
        const int o = e & 7;
        d[o] = (el_native_t)p0; d[o + 1] = (el_native_t)p1; d[o + 2] = (el_native_t)p2; d[o + 3] = (el_native_t)p3;
        r0 += p0; r1 += p1; r2 += p2; r3 += p3;
      }
    return (r0 + r1) + (r2 + r3);
  };
  auto tile = [&](int t, auto masked_tag) {
    const char* kst = smem + (t % kNS64) * 16384;
    const char* vst = kst + 8192;
    elx8 pf[2][2][2];       // [row block][32-key half][16-key step]
    float rs[2];
    {
      f32x16 sacc[2][2];
      scores(kst, sacc, t, masked_tag, std::integral_constant<bool, PRE>{});
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) rs[rb] = exp_pack(sacc[rb], pf[rb], m_run[rb], std::integral_constant<bool, PRE>{});
    }
    if (!__all(rs[0] <= kSumLimit && rs[1] <= kSumLimit)) {       // slow path: see attn_spatial_kernel
      f32x16 sacc[2][2];
      scores(kst, sacc, t, masked_tag, std::false_type{});
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float mx = sacc[rb][0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[rb][kt][e]);
        mx = half_max(mx) * kScale;
        const float m_new = fmaxf(m_run[rb], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
        m_run[rb] = m_new;
        l_run[rb] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[rb][dt][e] *= alpha;
        if (PRE) {
#pragma unroll
          for (int e = 0; e < 16; ++e) negm[rb][e] = -m_new;
        }
        rs[rb] = exp_pack(sacc[rb], pf[rb], m_run[rb], std::false_type{});
      }
    }
    l_run[0] += rs[0];
    l_run[1] += rs[1];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kb = kt * 32 + 16 * s + vkey;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const elx8 vf = vt_frag(vst, v_off(kb, dt * 32 + vcol), v_off(kb + 8, dt * 32 + vcol));
          oacc[0][dt] = mfma_32x32x16(vf, pf[0][kt][s], oacc[0][dt]);
          oacc[1][dt] = mfma_32x32x16(vf, pf[1][kt][s], oacc[1][dt]);
        }
      }
    }
  };

  const int nt = (S + 63) / 64;
  const int nt_full = S / 64;
  // kNS64 - 1 tiles in flight.  Every wave issues exactly four pieces per tile -- past the last tile too (all lanes out
  // of range: zeros into a free slot, no memory traffic) -- so the counted wait is the same at every iteration.
#pragma unroll
  for (int p = 0; p < kNS64 - 1; ++p) issue(p, p);
  for (int t = 0; t < nt_full; ++t) {
    wait_vm<4 * (kNS64 - 2)>();
    asm volatile("s_barrier" ::: "memory");
    issue(t + kNS64 - 1, (t + kNS64 - 1) % kNS64);
    tile(t, std::false_type{});
  }
  if (nt_full < nt) {
    wait_vm<4 * (kNS64 - 2)>();
    asm volatile("s_barrier" ::: "memory");
    tile(nt_full, std::true_type{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the look-ahead pieces: nothing in flight at exit

#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const float l_tot = half_sum(l_run[rb]);
    const float m_fin = m_run[rb];
    const float inv = 1.0f / l_tot;
    if (lse && qrow[rb] < S && hsel == 0)
      lse[((long)img * gridDim.y + head) * S + qrow[rb]] = m_fin + __builtin_amdgcn_logf(l_tot);
    if (qrow[rb] < S) {
      el_t* op = out + (row0 + qrow[rb]) * C + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dcol = dt * 32 + 8 * q + 4 * hsel;
          uint2 pk = make_uint2(pack_elx2(oacc[rb][dt][4 * q] * inv, oacc[rb][dt][4 * q + 1] * inv),
                                pack_elx2(oacc[rb][dt][4 * q + 2] * inv, oacc[rb][dt][4 * q + 3] * inv));
          *(uint2*)(op + dcol) = pk;
        }
    }
  }
  CTRLV_CLOCK_END();
}

// ---------------------------------------------------------------------------------------------- temporal
// One wave per (clip, pixel, head): Q, K and V of the 25 frames (32 x 64 tiles, rows >= F zero) all arrive by LDS-DMA
// through a per-clip buffer descriptor -- every load is 8 rows x 128 contiguous bytes, frames >= F are out of range --
// and the 25 x 64 output goes back through the (dead) Q tile so that the stores are 16 B per lane, row-contiguous.
// The first version gathered Q / K fragments straight from global memory (32 B per cache line per instruction) and
// stored 8 B per lane at a 17 MB row stride: 3.07 TB/s on a kernel that moves 4 x M x C x 2 bytes and nothing else.
__global__ __launch_bounds__(256, 2) void attn_temporal_kernel(const el_t* __restrict__ qkv, el_t* __restrict__ out,
                                                            int B, int F, int S, int C) {
  __shared__ __attribute__((aligned(1024))) char smem[4 * 12288];  // per wave: Q | K | V tiles of 32 x 64 bf16
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int heads = C >> 6;
  const long nprob = (long)B * S * heads;
  const long pid = (long)blockIdx.x * 4 + wid;
  if (pid >= nprob) return;  // whole-wave exit (no workgroup barrier is used below)
  const int head = (int)(pid % heads);
  const long bs = pid / heads;
  const int b = (int)(bs / S), s = (int)(bs % S);
  const int ld = 3 * C;
  const int r32 = lane & 31, hsel = lane >> 5, sw = (lane >> 1) & 7;

  char* qst = smem + wid * 12288;
  char* kst = qst + 4096;
  char* vst = qst + 8192;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(qkv + (long)b * F * S * ld), 0, (int)((long)F * S * ld * 2), 0x00020000);
  const int prow = lane >> 3, pslot = lane & 7;
  const unsigned row_off = (unsigned)((prow * S + s) * ld + head * 64) * 2u;      // frame `prow` of this pixel / head
  // swizzles (see the spatial kernel): Q / K chunk ^= (row >> 1) & 7, V chunk ^= ((row >> 1) & 1) << 2; row = 8 q + prow
  const unsigned qk_even = row_off + (unsigned)((pslot ^ (prow >> 1)) * 16);
  const unsigned qk_odd = row_off + (unsigned)((pslot ^ ((prow >> 1) | 4)) * 16);
  const unsigned v_voff = row_off + (unsigned)((pslot ^ (((prow >> 1) & 1) << 2)) * 16);
  const int frame8 = 8 * S * ld * 2;                                              // bytes between row groups of 8 frames
  // (only the per-lane offset is range-checked, so the frame-group term must be part of it: frames >= F are then past
  // num_records and read zeros; the q | k | v column block is a scalar offset)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned qk = ((q & 1) ? qk_odd : qk_even) + (unsigned)(q * frame8);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(qst + q * 1024), 16, qk, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(kst + q * 1024), 16, qk, C * 2, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(vst + q * 1024), 16, v_voff + (unsigned)(q * frame8), C * 4, 0, 0);

  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // this wave's own Q and K pieces have landed (V may still fly)
  __builtin_amdgcn_sched_barrier(0);
  f32x16 sacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int fo = r32 * 128 + (((ks * 2 + hsel) ^ sw) * 16);
    const elx8 kf = *(const elx8*)(kst + fo);
    const elx8 qf = *(const elx8*)(qst + fo);
    sacc = mfma_32x32x16(kf, qf, sacc);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int key = (e & 3) + 8 * (e >> 2) + 4 * hsel;
    if (key >= F) sacc[e] = -INFINITY;
    mx = fmaxf(mx, sacc[e]);
  }
  mx = half_max(mx) * kScaleLog2;
  float rs_ = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float p = __builtin_amdgcn_exp2f(sacc[e] * kScaleLog2 - mx);
    sacc[e] = p;
    rs_ += p;
  }
  const float inv = 1.0f / half_sum(rs_);

  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
  const int i16 = lane & 15;
  const int vkey = 4 * hsel + (i16 >> 2);
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // V landed
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const elx8 pf = pack_p(sacc, st);
    const int kb = 16 * st + vkey;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const elx8 vf = vt_frag(vst, v_off(kb, dt * 32 + vcol), v_off(kb + 8, dt * 32 + vcol));
      oacc[dt] = mfma_32x32x16(vf, pf, oacc[dt]);
    }
  }
  // O^T (lane = query frame r32, 32 d-values) -> row-major [frame][64] bf16 in the Q tile (its fragments are consumed;
  // LDS accesses of one wave execute in order), 16-B chunk index XOR (row & 7) against bank conflicts
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dcol = dt * 32 + 8 * q + 4 * hsel;
      const uint2 pk = make_uint2(pack_elx2(oacc[dt][4 * q] * inv, oacc[dt][4 * q + 1] * inv),
                                  pack_elx2(oacc[dt][4 * q + 2] * inv, oacc[dt][4 * q + 3] * inv));
      *(uint2*)(qst + r32 * 128 + (((dcol >> 3) ^ (r32 & 7)) * 16) + (dcol & 4) * 2) = pk;
    }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int fr = q * 8 + prow;
    const uint4 v = *(const uint4*)(qst + fr * 128 + ((pslot ^ (fr & 7)) * 16));
    if (fr < F) *(uint4*)(out + ((long)(b * F + fr) * S + s) * C + head * 64 + pslot * 8) = v;
  }
}

}  // namespace

static int attention_spatial_launch(const void* qkv, void* out, float* lse, int n_img, int S, int C, bool pre, hipStream_t stream) {
  CTRLV_CHECK_ARG(qkv && out, "attention_spatial: null pointer");
  CTRLV_CHECK_SHAPE(n_img > 0 && S > 0 && C > 0 && C % 64 == 0, "attention_spatial: C=%d must be a multiple of 64 (head_dim 64)", C);
  CTRLV_CHECK_SHAPE(n_img <= 65535 && C / 64 <= 65535, "attention_spatial: grid too large");
  // 64 query rows per wave on the long sequences; ctrlv_debug().attn_rows = 32 | 64 forces one (A/B, tests)
  const int rows = ctrlv_debug().attn_rows;
  const bool use64 = rows == 64 || (rows != 32 && S >= 1024);   // measured: +4-5 % at S = 9216 / 2304, -16 % at S = 576
  if (use64) {
    dim3 grid64((S + 255) / 256, C / 64, n_img);
    if (pre)
      hipLaunchKernelGGL(attn_spatial64_kernel<true>, grid64, dim3(256), kNS64 * 16384, stream, (const el_t*)qkv, (el_t*)out, lse, S, C);
    else
      hipLaunchKernelGGL(attn_spatial64_kernel<false>, grid64, dim3(256), kNS64 * 16384, stream, (const el_t*)qkv, (el_t*)out, lse, S, C);
    CTRLV_LAUNCH_CHECK();
    return CTRLV_OK;
  }
  // 32 rows per wave, 2-slot K/V ring (32 KiB LDS, 4 waves/SIMD): 883 TFLOP/s at S = 9216 against 810 for 3 slots (two
  // tiles of LDS-DMA in flight but 3 waves/SIMD) -- occupancy beats prefetch depth for this VALU-heavy d = 64 kernel.
  dim3 grid((S + 127) / 128, C / 64, n_img);
  if (pre)
    hipLaunchKernelGGL((attn_spatial_kernel<2, true>), grid, dim3(256), 32768, stream, (const el_t*)qkv, (el_t*)out, lse, S, C);
  else
    hipLaunchKernelGGL((attn_spatial_kernel<2, false>), grid, dim3(256), 32768, stream, (const el_t*)qkv, (el_t*)out, lse, S, C);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

CTRLV_CLOCK_READER(attention)

extern "C" int ctrlv_attention_spatial(const void* qkv, void* out, int n_img, int S, int C, ctrlv_stream_t stream) {
  return attention_spatial_launch(qkv, out, nullptr, n_img, S, C, false, (hipStream_t)stream);
}
extern "C" int ctrlv_attention_spatial_lse(const void* qkv, void* out, float* lse, int n_img, int S, int C, ctrlv_stream_t stream) {
  return attention_spatial_launch(qkv, out, lse, n_img, S, C, false, (hipStream_t)stream);
}
extern "C" int ctrlv_attention_spatial_prescaled(const void* qkv, void* out, int n_img, int S, int C, ctrlv_stream_t stream) {
  return attention_spatial_launch(qkv, out, nullptr, n_img, S, C, true, (hipStream_t)stream);
}

extern "C" int ctrlv_attention_temporal(const void* qkv, void* out, int B, int F, int S, int C,
                                        ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(qkv && out, "attention_temporal: null pointer");
  CTRLV_CHECK_SHAPE(B > 0 && S > 0 && C > 0 && C % 64 == 0, "attention_temporal: C=%d must be a multiple of 64", C);
  CTRLV_CHECK_SHAPE(F > 0 && F <= 32, "attention_temporal: F=%d frames must be in [1, 32]", F);
  const long nprob = (long)B * S * (C / 64);
  hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)((nprob + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const el_t*)qkv, (el_t*)out, B, F, S, C);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
