// Backward of the self-attention cores (training step): head_dim 64, bf16 in/out, fp32 softmax and
// accumulation.  Same MFMA idiom as attention.hip -- S^T tiles with one query (or key) per lane, the accumulator tile
// re-used as the B operand of the next MFMA after a bf16 pack, transposed fragments through ds_read_b64_tr_b16.
//
//   P_ij  = exp2(s_ij * scale * log2e - L_i)          L_i = m_i + log2(l_i) saved by the forward (ctrlv_attention_spatial_lse)
//   D_i   = sum_d dO_id * O_id
//   dV_j  = sum_i P_ij dO_i          dP_ij = dO_i . V_j          dS_ij = P_ij (dP_ij - D_i) * scale
//   dQ_i  = sum_j dS_ij K_j          dK_j  = sum_i dS_ij Q_i
//
// spatial, two kernels (no atomics, deterministic):
//   dq   : query-stationary, 4 waves x 32 queries per workgroup, loops over 64-key tiles (K row-major + K for
//          transposed reads + V row-major by LDS-DMA); also writes D.                       24 MFMAs per 32 x 64 block
//   dkdv : key-stationary, 4 waves x 32 keys, loops over 64-query tiles (Q, dO each in both forms, L and D strips).
//                                                                                              32 MFMAs per 32 x 64 block
// temporal: one wave per (clip, pixel, head) like the forward; both orientations of the 32 x 32 score tile are
//   computed (lane = query for dQ, lane = key for dK / dV), L and D go through a small LDS strip, no saved L needed.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr float kScaleLog2 = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) * log2(e)
constexpr float kScale = 0.125f;

__device__ __forceinline__ float half_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// transposed fragment (A operand, 32 columns x 16 rows of a row-major [rows][64] bf16 tile), see attention.hip
__device__ __forceinline__ elx8 t_frag(const char* tile, int off_lo, int off_hi) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off_lo));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + off_hi));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(elx8, v);
}
// byte offset of (row, d) in a tile staged for transposed reads: 16-B chunk index ^= ((row >> 1) & 1) << 2
__device__ __forceinline__ int t_off(int row, int d) {
  const int chunk = (d >> 3) ^ (((row >> 1) & 1) << 2);
  return row * 128 + chunk * 16 + (d & 7) * 2;
}
__device__ __forceinline__ elx8 pack8(const f32x16& p, int s) {
  elx8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (el_native_t)p[8 * s + j];
  return r;
}
__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
  float fa[8], fb[8];
  unpack_elx8(a, fa);
  unpack_elx8(b, fb);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s = __builtin_fmaf(fa[j], fb[j], s);
  return s;
}
// store a transposed accumulator pair (lane = row r, 2 x 16 registers = 64 columns) as one bf16 row
__device__ __forceinline__ void store_row64(el_t* dst, const f32x16 (&acc)[2], int hsel, float scale) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dcol = dt * 32 + 8 * q + 4 * hsel;
      uint2 pk = make_uint2(pack_elx2(acc[dt][4 * q] * scale, acc[dt][4 * q + 1] * scale),
                            pack_elx2(acc[dt][4 * q + 2] * scale, acc[dt][4 * q + 3] * scale));
      *(uint2*)(dst + dcol) = pk;
    }
}

// ------------------------------------------------------------------------------------------ spatial: dQ (+ D)
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const el_t* __restrict__ qkv, const el_t* __restrict__ out,
                                                             const el_t* __restrict__ dout, const float* __restrict__ lse,
                                                             el_t* __restrict__ dqkv, float* __restrict__ delta, int S,
                                                             int C) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 2 x (K rows 8 KiB | K for tr reads 8 KiB | V rows 8 KiB)
  constexpr int SLOT = 24576;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5, sw = (lane >> 1) & 7;
  const int head = blockIdx.y, img = blockIdx.z;
  const long row0 = (long)img * S;
  const int ld = 3 * C;
  const int qrow = blockIdx.x * 128 + wid * 32 + r32;
  const bool qok = qrow < S;

  // Q and dO fragments of this lane's query (B operands), D = dO . O, L
  elx8 qf[4], dof[4];
  float dpart = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4 q = make_uint4(0, 0, 0, 0), g = q, o = q;
    if (qok) {
      q = *(const uint4*)(qkv + (row0 + qrow) * ld + head * 64 + 16 * ks + 8 * hsel);
      g = *(const uint4*)(dout + (row0 + qrow) * C + head * 64 + 16 * ks + 8 * hsel);
      o = *(const uint4*)(out + (row0 + qrow) * C + head * 64 + 16 * ks + 8 * hsel);
    }
    qf[ks] = __builtin_bit_cast(elx8, q);
    dof[ks] = __builtin_bit_cast(elx8, g);
    dpart += dot8(g, o);
  }
  const float D = half_sum(dpart);
  const long stat = ((long)img * gridDim.y + head) * S + qrow;
  const float L = qok ? lse[stat] : 0.f;
  if (qok && hsel == 0) delta[stat] = D;

  const int prow = lane >> 3, pslot = lane & 7;
  const __amdgpu_buffer_rsrc_t rs_kv =
      __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + row0 * ld), 0, (int)((long)S * ld * 2), 0x00020000);
  const int rt0 = wid * 8 + prow;
  const unsigned k_rm = (unsigned)(rt0 * ld + C + head * 64 + (pslot ^ ((rt0 >> 1) & 7)) * 8) * 2u;
  const unsigned k_tr = (unsigned)(rt0 * ld + C + head * 64 + (pslot ^ (((rt0 >> 1) & 1) << 2)) * 8) * 2u;
  const unsigned v_rm = k_rm + (unsigned)(C * 2);
  const int tile_bytes = 64 * ld * 2;
  const int full_tiles = S / 64;
  auto issue = [&](int t, int stage) {
    char* st = smem + stage * SLOT;
    const bool ragged = t >= full_tiles;   // keys >= S must fall past num_records: the tile offset joins the lane offset
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int so = t * tile_bytes + q * (tile_bytes >> 1);
      const unsigned add = ragged ? (unsigned)so : 0u;
      const int sso = ragged ? 0 : so;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(st + (q * 4 + wid) * 1024), 16, k_rm + add, sso, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(st + 8192 + (q * 4 + wid) * 1024), 16, k_tr + add, sso, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(st + 16384 + (q * 4 + wid) * 1024), 16, v_rm + add, sso, 0, 0);
    }
  };

  f32x16 dq[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[dt][e] = 0.f;
  const int i16 = lane & 15;
  const int vkey = 4 * hsel + (i16 >> 2);
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);

  auto tile = [&](int t, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const char* krm = smem + (t & 1) * SLOT;
    const char* ktr = krm + 8192;
    const char* vrm = krm + 16384;
    f32x16 sacc[2], dp[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { sacc[kt][e] = 0.f; dp[kt][e] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int fo = (kt * 32 + r32) * 128 + (((ks * 2 + hsel) ^ sw) * 16);
        const elx8 kf = *(const elx8*)(krm + fo);
        const elx8 vf = *(const elx8*)(vrm + fo);
        sacc[kt] = mfma_32x32x16(kf, qf[ks], sacc[kt]);
        dp[kt] = mfma_32x32x16(vf, dof[ks], dp[kt]);
      }
    }
    // dS^T in place of the scores (lane = query: L and D are per-lane scalars)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e], kScaleLog2, -L));
        float ds = p * (dp[kt][e] - D);
        if (MASKED) {      // ragged last key tile: K / V rows are zeros there, but p may be huge
          const int key = t * 64 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hsel;
          if (key >= S) ds = 0.f;
        }
        sacc[kt][e] = ds;
      }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const elx8 pf = pack8(sacc[kt], s);
        const int kb = kt * 32 + 16 * s + vkey;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const elx8 kf = t_frag(ktr, t_off(kb, dt * 32 + vcol), t_off(kb + 8, dt * 32 + vcol));
          dq[dt] = mfma_32x32x16(kf, pf, dq[dt]);
        }
      }
  };

  const int nt = (S + 63) / 64;
  issue(0, 0);
  for (int t = 0; t < full_tiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
    tile(t, std::false_type{});
  }
  if (full_tiles < nt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    tile(full_tiles, std::true_type{});
  }
  if (qok) store_row64(dqkv + (row0 + qrow) * ld + head * 64, dq, hsel, kScale);
}

// ------------------------------------------------------------------------------------------ spatial: dK, dV
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const el_t* __restrict__ qkv, const el_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               el_t* __restrict__ dqkv, int S, int C) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 2 x (Q rows | Q tr | dO rows | dO tr | L | D)
  constexpr int SLOT = 4 * 8192 + 512;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5, sw = (lane >> 1) & 7;
  const int head = blockIdx.y, img = blockIdx.z;
  const long row0 = (long)img * S;
  const int ld = 3 * C;
  const int krow = blockIdx.x * 128 + wid * 32 + r32;
  const bool kok = krow < S;

  elx8 kf[4], vf[4];       // this lane's key: B operands of S = Q.K^T and dP = dO.V^T
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4 k = make_uint4(0, 0, 0, 0), v = k;
    if (kok) {
      k = *(const uint4*)(qkv + (row0 + krow) * ld + C + head * 64 + 16 * ks + 8 * hsel);
      v = *(const uint4*)(qkv + (row0 + krow) * ld + 2 * C + head * 64 + 16 * ks + 8 * hsel);
    }
    kf[ks] = __builtin_bit_cast(elx8, k);
    vf[ks] = __builtin_bit_cast(elx8, v);
  }

  const int prow = lane >> 3, pslot = lane & 7;
  const __amdgpu_buffer_rsrc_t rs_q =
      __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + row0 * ld), 0, (int)((long)S * ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g =
      __builtin_amdgcn_make_buffer_rsrc((void*)(dout + row0 * C), 0, (int)((long)S * C * 2), 0x00020000);
  const long stat0 = ((long)img * gridDim.y + head) * S;
  const __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)(lse + stat0), 0, S * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)(delta + stat0), 0, S * 4, 0x00020000);
  const int rt0 = wid * 8 + prow;
  const int sw_rm = (pslot ^ ((rt0 >> 1) & 7)) * 8, sw_tr = (pslot ^ (((rt0 >> 1) & 1) << 2)) * 8;
  const unsigned q_rm = (unsigned)(rt0 * ld + head * 64 + sw_rm) * 2u, q_tr = (unsigned)(rt0 * ld + head * 64 + sw_tr) * 2u;
  const unsigned g_rm = (unsigned)(rt0 * C + head * 64 + sw_rm) * 2u, g_tr = (unsigned)(rt0 * C + head * 64 + sw_tr) * 2u;
  auto issue = [&](int t, int stage) {
    char* st = smem + stage * SLOT;
    // (the tile offset is part of the lane offset: only that is range-checked, queries >= S must read zeros)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned rq = (unsigned)((t * 64 + q * 32) * ld * 2), rg = (unsigned)((t * 64 + q * 32) * C * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, LDS_PTR(st + (q * 4 + wid) * 1024), 16, q_rm + rq, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_q, LDS_PTR(st + 8192 + (q * 4 + wid) * 1024), 16, q_tr + rq, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, LDS_PTR(st + 16384 + (q * 4 + wid) * 1024), 16, g_rm + rg, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_g, LDS_PTR(st + 24576 + (q * 4 + wid) * 1024), 16, g_tr + rg, 0, 0, 0);
    }
    if (wid == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_l, LDS_PTR(st + 32768), 4, (unsigned)((t * 64 + lane) * 4), 0, 0, 0);
    if (wid == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, LDS_PTR(st + 32768 + 256), 4, (unsigned)((t * 64 + lane) * 4), 0, 0, 0);
  };

  f32x16 dk[2], dv[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[dt][e] = 0.f; dv[dt][e] = 0.f; }
  const int i16 = lane & 15;
  const int vkey = 4 * hsel + (i16 >> 2);
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);

  auto tile = [&](int t) {
    const char* qrm = smem + (t & 1) * SLOT;
    const char* qtr = qrm + 8192;
    const char* grm = qrm + 16384;
    const char* gtr = qrm + 24576;
    const float* Ls = (const float*)(qrm + 32768);
    const float* Ds = Ls + 64;
    f32x16 sacc[2], dp[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { sacc[qt][e] = 0.f; dp[qt][e] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int fo = (qt * 32 + r32) * 128 + (((ks * 2 + hsel) ^ sw) * 16);
        const elx8 qa = *(const elx8*)(qrm + fo);
        const elx8 ga = *(const elx8*)(grm + fo);
        sacc[qt] = mfma_32x32x16(qa, kf[ks], sacc[qt]);
        dp[qt] = mfma_32x32x16(ga, vf[ks], dp[qt]);
      }
    }
    // lane = key; register e of block qt is query qt*32 + 8*(e>>2) + 4*hsel + (e&3): L, D come as 4-vectors
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 l4 = *(const float4*)(Ls + qt * 32 + 8 * q4 + 4 * hsel);
        const float4 d4 = *(const float4*)(Ds + qt * 32 + 8 * q4 + 4 * hsel);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = 4 * q4 + r;
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[qt][e], kScaleLog2, -lv[r]));
          sacc[qt][e] = p;
          dp[qt][e] = p * (dp[qt][e] - dvv[r]);
        }
      }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const elx8 pf = pack8(sacc[qt], s);
        const elx8 dsf = pack8(dp[qt], s);
        const int qb = qt * 32 + 16 * s + vkey;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int o_lo = t_off(qb, dt * 32 + vcol), o_hi = t_off(qb + 8, dt * 32 + vcol);
          dv[dt] = mfma_32x32x16(t_frag(gtr, o_lo, o_hi), pf, dv[dt]);
          dk[dt] = mfma_32x32x16(t_frag(qtr, o_lo, o_hi), dsf, dk[dt]);
        }
      }
  };

  const int nt = (S + 63) / 64;
  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
    tile(t);
  }
  if (kok) {
    store_row64(dqkv + (row0 + krow) * ld + C + head * 64, dk, hsel, kScale);
    store_row64(dqkv + (row0 + krow) * ld + 2 * C + head * 64, dv, hsel, 1.0f);
  }
}

// ------------------------------------------------------------------------------------------ temporal
// One wave per (clip, pixel, head).  LDS per wave: Q, K, dO each as a row tile (fragment reads) and as a tile for
// transposed reads, V as a row tile: 7 x 4 KiB, plus 256 B for L and D.  Rows >= F are zeros (out of range).
__global__ __launch_bounds__(256, 1) void attn_temporal_bwd_kernel(const el_t* __restrict__ qkv, const el_t* __restrict__ out,
                                                                   const el_t* __restrict__ dout, el_t* __restrict__ dqkv,
                                                                   int B, int F, int S, int C) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int WAVE_LDS = 7 * 4096 + 256;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int heads = C >> 6;
  const long nprob = (long)B * S * heads;
  const long pid = (long)blockIdx.x * 4 + wid;
  if (pid >= nprob) return;  // whole-wave exit (no workgroup barrier below)
  const int head = (int)(pid % heads);
  const long bs = pid / heads;
  const int b = (int)(bs / S), s = (int)(bs % S);
  const int ld = 3 * C;
  const int r32 = lane & 31, hsel = lane >> 5, sw = (lane >> 1) & 7;

  char* base = smem + wid * WAVE_LDS;
  char* q_rm = base, *q_tr = base + 4096, *k_rm = base + 8192, *k_tr = base + 12288, *v_rm = base + 16384;
  char* g_rm = base + 20480, *g_tr = base + 24576;
  float* Ls = (float*)(base + 28672);
  float* Ds = Ls + 32;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(qkv + (long)b * F * S * ld), 0, (int)((long)F * S * ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(dout + (long)b * F * S * C), 0, (int)((long)F * S * C * 2), 0x00020000);
  const int prow = lane >> 3, pslot = lane & 7;
  // piece q covers frames 8 q + prow; row-tile swizzle: chunk ^= (row >> 1) & 7 (row = 8 q + prow -> (prow >> 1) | 4*(q & 1));
  // transposed-read swizzle: chunk ^= ((row >> 1) & 1) << 2
  const unsigned row_q = (unsigned)((prow * S + s) * ld + head * 64) * 2u;
  const unsigned row_g = (unsigned)((prow * S + s) * C + head * 64) * 2u;
  const int f8q = 8 * S * ld * 2, f8g = 8 * S * C * 2;
  const unsigned tr_sw = (unsigned)((pslot ^ (((prow >> 1) & 1) << 2)) * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned rm_sw = (unsigned)((pslot ^ ((prow >> 1) | ((q & 1) << 2))) * 16);
    const unsigned oq = row_q + (unsigned)(q * f8q), og = row_g + (unsigned)(q * f8g);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(q_rm + q * 1024), 16, oq + rm_sw, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(k_rm + q * 1024), 16, oq + rm_sw, C * 2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(v_rm + q * 1024), 16, oq + rm_sw, C * 4, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, LDS_PTR(g_rm + q * 1024), 16, og + rm_sw, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(q_tr + q * 1024), 16, oq + tr_sw, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(k_tr + q * 1024), 16, oq + tr_sw, C * 2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsg, LDS_PTR(g_tr + q * 1024), 16, og + tr_sw, 0, 0, 0);
  }
  // D_i = dO_i . O_i for frame r32 (half of the 64 columns per half-wave), O straight from memory
  float dpart = 0.f;
  if (r32 < F) {
    const long orow = ((long)(b * F + r32) * S + s) * C + head * 64;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 o = *(const uint4*)(out + orow + 16 * ks + 8 * hsel);
      const uint4 g = *(const uint4*)(dout + orow + 16 * ks + 8 * hsel);
      dpart += dot8(g, o);
    }
  }
  const float D = half_sum(dpart);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- orientation A (lane = query): S^T = K.Q^T, softmax statistics, dP^T = V.dO^T, dS^T, dQ^T = K^T.dS^T
  f32x16 sacc, dp;
#pragma unroll
  for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int fo = r32 * 128 + (((ks * 2 + hsel) ^ sw) * 16);
    const elx8 kf = *(const elx8*)(k_rm + fo);
    const elx8 qf = *(const elx8*)(q_rm + fo);
    const elx8 vf = *(const elx8*)(v_rm + fo);
    const elx8 gf = *(const elx8*)(g_rm + fo);
    sacc = mfma_32x32x16(kf, qf, sacc);
    dp = mfma_32x32x16(vf, gf, dp);
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
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], kScaleLog2, -mx));
    sacc[e] = p;
    rs_ += p;
  }
  const float l_tot = half_sum(rs_);
  const float inv = 1.0f / l_tot;
  if (hsel == 0) { Ls[r32] = mx + __builtin_amdgcn_logf(l_tot); Ds[r32] = D; }
#pragma unroll
  for (int e = 0; e < 16; ++e) sacc[e] = sacc[e] * inv * (dp[e] - D);     // dS^T (masked keys: p = 0)
  const int i16 = lane & 15;
  const int vkey = 4 * hsel + (i16 >> 2);
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
  f32x16 acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[dt][e] = 0.f;
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const elx8 pf = pack8(sacc, st);
    const int kb = 16 * st + vkey;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      acc[dt] = mfma_32x32x16(t_frag(k_tr, t_off(kb, dt * 32 + vcol), t_off(kb + 8, dt * 32 + vcol)), pf, acc[dt]);
  }
  if (r32 < F) store_row64(dqkv + ((long)(b * F + r32) * S + s) * ld + head * 64, acc, hsel, kScale);

  // ---- orientation B (lane = key): S = Q.K^T, dP = dO.V^T with L, D per register; dV^T = dO^T.P, dK^T = Q^T.dS
  __builtin_amdgcn_wave_barrier();      // Ls / Ds written above are read below by other lanes of this wave
#pragma unroll
  for (int e = 0; e < 16; ++e) { sacc[e] = 0.f; dp[e] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int fo = r32 * 128 + (((ks * 2 + hsel) ^ sw) * 16);
    const elx8 qa = *(const elx8*)(q_rm + fo);
    const elx8 kb_ = *(const elx8*)(k_rm + fo);
    const elx8 ga = *(const elx8*)(g_rm + fo);
    const elx8 vb = *(const elx8*)(v_rm + fo);
    sacc = mfma_32x32x16(qa, kb_, sacc);
    dp = mfma_32x32x16(ga, vb, dp);
  }
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const float4 l4 = *(const float4*)(Ls + 8 * q4 + 4 * hsel);
    const float4 d4 = *(const float4*)(Ds + 8 * q4 + 4 * hsel);
    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = 4 * q4 + r;
      // (padding queries >= F: Q and dO rows are zeros, L = log2(F) is finite, so p is finite and dS = p * (0 - 0) = 0)
      const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], kScaleLog2, -lv[r]));
      sacc[e] = p;
      dp[e] = p * (dp[e] - dvv[r]);
    }
  }
  f32x16 dk[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[dt][e] = 0.f; dk[dt][e] = 0.f; }
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const elx8 pf = pack8(sacc, st);
    const elx8 dsf = pack8(dp, st);
    const int qb = 16 * st + vkey;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int o_lo = t_off(qb, dt * 32 + vcol), o_hi = t_off(qb + 8, dt * 32 + vcol);
      acc[dt] = mfma_32x32x16(t_frag(g_tr, o_lo, o_hi), pf, acc[dt]);
      dk[dt] = mfma_32x32x16(t_frag(q_tr, o_lo, o_hi), dsf, dk[dt]);
    }
  }
  if (r32 < F) {
    el_t* drow = dqkv + ((long)(b * F + r32) * S + s) * ld + head * 64;
    store_row64(drow + C, dk, hsel, kScale);
    store_row64(drow + 2 * C, acc, hsel, 1.0f);
  }
}

}  // namespace

extern "C" size_t ctrlv_attention_bwd_scratch_floats(int n_img, int S, int C) { return (size_t)n_img * (C / 64) * S; }

extern "C" int ctrlv_attention_spatial_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                           float* delta, int n_img, int S, int C, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(qkv && out && dout && lse && dqkv && delta, "attention_spatial_bwd: null pointer");
  CTRLV_CHECK_SHAPE(n_img > 0 && S > 0 && C > 0 && C % 64 == 0, "attention_spatial_bwd: C=%d must be a multiple of 64", C);
  CTRLV_CHECK_SHAPE(n_img <= 65535 && C / 64 <= 65535, "attention_spatial_bwd: grid too large");
  CTRLV_CHECK_SHAPE((long)S * 3 * C * 2 < 0x7FFFFFFFL, "attention_spatial_bwd: one image's qkv must stay below 2 GiB");
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  constexpr int kSmemDq = 2 * 24576, kSmemKv = 2 * (4 * 8192 + 512);
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmemKv));
    attr_set[dev] = true;
  }
  dim3 grid((S + 127) / 128, C / 64, n_img);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), kSmemDq, (hipStream_t)stream, (const el_t*)qkv, (const el_t*)out,
                     (const el_t*)dout, lse, (el_t*)dqkv, delta, S, C);
  CTRLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), kSmemKv, (hipStream_t)stream, (const el_t*)qkv,
                     (const el_t*)dout, lse, (const float*)delta, (el_t*)dqkv, S, C);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_attention_temporal_bwd(const void* qkv, const void* out, const void* dout, void* dqkv, int B, int F, int S,
                                            int C, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(qkv && out && dout && dqkv, "attention_temporal_bwd: null pointer");
  CTRLV_CHECK_SHAPE(B > 0 && S > 0 && C > 0 && C % 64 == 0, "attention_temporal_bwd: C=%d must be a multiple of 64", C);
  CTRLV_CHECK_SHAPE(F > 0 && F <= 32, "attention_temporal_bwd: F=%d frames must be in [1, 32]", F);
  CTRLV_CHECK_SHAPE((long)F * S * 3 * C * 2 < 0x7FFFFFFFL, "attention_temporal_bwd: one clip's qkv must stay below 2 GiB");
  constexpr int kSmem = 4 * (7 * 4096 + 256);
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  const int dev = ctrlv_current_device();
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)attn_temporal_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
    attr_set[dev] = true;
  }
  const long nprob = (long)B * S * (C / 64);
  hipLaunchKernelGGL(attn_temporal_bwd_kernel, dim3((unsigned)((nprob + 3) / 4)), dim3(256), kSmem, (hipStream_t)stream,
                     (const el_t*)qkv, (const el_t*)out, (const el_t*)dout, (el_t*)dqkv, B, F, S, C);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
