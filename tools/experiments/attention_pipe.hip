// Spatial self-attention, cross-tile software-pipelined variant (head_dim 64, 64 query rows per wave, ONE wave per SIMD).
//
// Why: the kernels of attention.hip run [16 K.Q^T MFMAs][~240 softmax VALU][16 P.V MFMAs] per 64-key tile in every wave.
// A wave issues in order, so its own VALU work never overlaps its own MFMAs, and two such waves per SIMD overlap each
// other's phases only partly: measured 4390 cycles per pair of tiles per SIMD against 2048 cycles of matrix-pipe work
// (MFMA busy 47 % at the sustained clock).  The issue bound is ~1470 cycles per tile: 32 MFMAs hold the vector issue for
// 8 cycles each, 64 v_exp_f32 cost 8, the other ~176 VALU 4 (MI355X_MICROARCH.md, "vector-instruction ISSUE cost").
//
// Here one wave owns the SIMD (512 registers) and carries TWO tiles at once, so that the instruction stream of an
// iteration consists of two INDEPENDENT halves that the scheduler may interleave freely in one basic block:
//     matrix stream : O += V(t-1)^T . P(t-1)          (P of the previous tile, already checked / rescaled)
//                     S(t+1) = K(t+1) . Q^T           (scores of the next tile)
//     vector stream : P(t) = exp2(S(t) * scale - m), row sums                (scores computed one iteration earlier)
// and the sum-triggered rescale test of attention.hip sits at the END of the iteration, before P(t) is ever used.
// K/V tiles: 8-slot LDS-DMA ring (tiles t-1 .. t+1 live, t+2 and t+3 in flight), one s_barrier per tile.
//
// STATUS: EXPERIMENT, off by default (CTRLV_ATTN_X=1 selects it; all spatial-attention tests pass with it).  Measured on
// MI355X at S = 9216: 7.5 ms against 5.5-5.7 ms for attention.hip's two-waves-per-SIMD kernel.  The emitted steady-state
// block does have the intended order (M vEvEvv M vEvEvv ...), and two hidden serialisations were found and removed on the
// way -- the compiler makes the ds_read_tr BUILTIN wait for every outstanding LDS-DMA (vmcnt(0): one HBM round trip per
// tile; fixed with asm reads + a manual lgkmcnt wait) -- but at one wave per SIMD hipcc's register allocator keeps the
// score accumulators in AGPRs and pays 64 v_accvgpr_read per tile plus ~100 more instructions around the P pack, so the
// wave issues ~520 instructions per tile where the two-wave kernel issues 268.  Forcing VGPR destinations (asm MFMAs,
// -amdgpu-mfma-vgpr-form) only moves the copies elsewhere.  The structure is right for a hand-allocated (assembly) kernel;
// as compiled HIP it loses.  See DESIGN.md section 8.
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
__device__ __forceinline__ bf16x8 vt_frag(const char* vt, int voff_lo, int voff_hi) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vt + voff_lo));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vt + voff_hi));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
// The same two transposed block reads issued through inline asm: the compiler makes the ds_read_tr builtin wait for EVERY
// outstanding LDS-DMA (s_waitcnt vmcnt(0)) because it cannot prove that the ring slot being read is not the one being
// filled -- here that would be a full HBM round trip per tile.  The asm form is invisible to that logic; `tr_wait` is the
// matching manual lgkmcnt wait, tied to the fragments by data dependence.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ void vt_frag_issue(const char* vt, int voff_lo, int voff_hi, u32x2_t& lo, u32x2_t& hi) {
  const unsigned a_lo = (unsigned)(size_t)LDS_PTR(vt + voff_lo), a_hi = (unsigned)(size_t)LDS_PTR(vt + voff_hi);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a_lo));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a_hi));
}
__device__ __forceinline__ bf16x8 vt_frag_join(const u32x2_t& lo, const u32x2_t& hi) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  const u32x4_ v = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ int v_off(int key, int d) {
  const int chunk = (d >> 3) ^ (((key >> 1) & 1) << 2);
  return key * 128 + chunk * 16 + (d & 7) * 2;
}
__device__ __forceinline__ bf16x8 pack_p(const f32x16& p, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)p[8 * s + j];
  return r;
}

constexpr int kSlots = 8, kSlotBytes = 16384;     // one workgroup per CU: 128 of the 160 KiB

__global__ __launch_bounds__(256, 1) void attn_spatial64x_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                 int S, int C) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 8 x (K 8 KiB | V 8 KiB) ring
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r32 = lane & 31, hsel = lane >> 5, sw = (lane >> 1) & 7;
  const int head = blockIdx.y, img = blockIdx.z;
  const long row0 = (long)img * S;
  const int ld = 3 * C;
  const bf16_t* qp = qkv + head * 64;

  int qrow[2];
  bf16x8 qf[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    qrow[rb] = blockIdx.x * 256 + wid * 64 + rb * 32 + r32;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qrow[rb] < S) v = *(const uint4*)(qp + (row0 + qrow[rb]) * ld + 16 * ks + 8 * hsel);
      qf[rb][ks] = __builtin_bit_cast(bf16x8, v);
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
  const int nt = (S + 63) / 64;
  auto issue_full = [&](int t) {             // a full tile: the tile offset is a scalar, no per-lane arithmetic
    char* ks_ = smem + (t & (kSlots - 1)) * kSlotBytes;
    char* vs_ = ks_ + 8192;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int so = t * tile_bytes + q * (tile_bytes >> 1);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(ks_ + (q * 4 + wid) * 1024), 16, koff, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_kv, LDS_PTR(vs_ + (q * 4 + wid) * 1024), 16, voff, so, 0, 0);
    }
  };
  auto issue = [&](int t) {
    char* ks_ = smem + (t & (kSlots - 1)) * kSlotBytes;
    char* vs_ = ks_ + 8192;
    const bool ragged = t >= full_tiles;   // keys >= S must fall past num_records (zeros): fold the tile offset into voffset
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
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const int i16 = lane & 15;
  const int vkey = 4 * hsel + (i16 >> 2);
  const int vcol = 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
  constexpr float kSumLimit = 4096.0f;

  // scores of tile t (raw, masked on the ragged tail unless FULL)
  auto scores = [&](f32x16 (&sacc)[2][2], int t, auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    const char* kst = smem + (t & (kSlots - 1)) * kSlotBytes;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[rb][kt][e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 kf = *(const bf16x8*)(kst + (kt * 32 + r32) * 128 + (((ks * 2 + hsel) ^ sw) * 16));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          sacc[rb][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[rb][ks], sacc[rb][kt], 0, 0, 0);
      }
    }
    if (!FULL && t >= full_tiles) {
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
  auto exp_sum = [&](f32x16 (&sacc)[2], float m) -> float {
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    const float nm = -m;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 16; e += 4) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e], kScaleLog2, nm));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e + 1], kScaleLog2, nm));
        const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e + 2], kScaleLog2, nm));
        const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][e + 3], kScaleLog2, nm));
        sacc[kt][e] = p0; sacc[kt][e + 1] = p1; sacc[kt][e + 2] = p2; sacc[kt][e + 3] = p3;
        r0 += p0; r1 += p1; r2 += p2; r3 += p3;
      }
    return (r0 + r1) + (r2 + r3);
  };
  // slow path: true row max folded in, O and l rescaled, tile exponentiated again (scores recomputed from the K tile)
  auto rescale = [&](f32x16 (&sacc)[2][2], int t, float (&rs)[2]) {
    scores(sacc, t, std::false_type{});
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      float mx = sacc[rb][0][0];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sacc[rb][kt][e]);
      mx = half_max(mx) * kScaleLog2;
      const float m_new = fmaxf(m_run[rb], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
      m_run[rb] = m_new;
      l_run[rb] *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[rb][dt][e] *= alpha;
      rs[rb] = exp_sum(sacc[rb], m_run[rb]);
    }
  };
  auto pack_tile = [&](const f32x16 (&sacc)[2][2], bf16x8 (&pb)[2][2][2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) pb[rb][kt][s] = pack_p(sacc[rb][kt], s);
  };
  auto pv = [&](const bf16x8 (&pb)[2][2][2], int t) {
    const char* vst = smem + (t & (kSlots - 1)) * kSlotBytes + 8192;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kb = kt * 32 + 16 * s + vkey;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const bf16x8 vf = vt_frag(vst, v_off(kb, dt * 32 + vcol), v_off(kb + 8, dt * 32 + vcol));
          oacc[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[0][kt][s], oacc[0][dt], 0, 0, 0);
          oacc[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[1][kt][s], oacc[1][dt], 0, 0, 0);
        }
      }
  };
  // ---- prologue: tiles 0..2 in flight; tile 0 goes through the slow path (m = -inf), its P is parked
  issue(0);
  if (nt > 1) issue(1);
  if (nt > 2) issue(2);
  if (nt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  f32x16 sa[2][2], sb[2][2];
  bf16x8 pprev[2][2][2];
  {
    float rs[2];
    rescale(sa, 0, rs);
    l_run[0] += rs[0];
    l_run[1] += rs[1];
    pack_tile(sa, pprev);
  }
  if (nt > 1) {                      // tiles 1 and 2 landed (3 in flight) before the first pipelined iteration
    if (nt > 3) issue(3);
    if (nt > 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    scores(sa, 1, std::false_type{});
  }
  // ---- steady state.  Entering iteration t: `cur` = raw scores of tile t, pprev = P(t-1), tiles <= t+1 issued.
  // ---- steady-state iteration, interleaved BY HAND: 32 slots, each = one MFMA of the matrix stream followed by the softmax
  // arithmetic of two score elements, fenced with sched_barrier so that the emitted order is the written order (the machine
  // scheduler, left alone or steered with sched_group_barrier, clusters the MFMAs and exposes every fragment read).
  // Matrix stream: 8 groups of 4 MFMAs alternating  P.V(t-1) chunk (kt, s)  /  K(t+1).Q^T k-step ks; the two fragments of
  // group g+1 are read from LDS while group g runs (double buffer).  Requires t + 3 < full_tiles.
  auto iteration_full = [&](f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], int t) {
    const char* vst = smem + ((t - 1) & (kSlots - 1)) * kSlotBytes + 8192;
    const char* kst = smem + ((t + 1) & (kSlots - 1)) * kSlotBytes;
    bf16x8 fr[2][2];
    u32x2_t vr[4];                                         // raw halves of the two V^T fragments in flight
    auto load_group = [&](int g, bf16x8 (&f)[2]) {
      if (g & 1) {
        const int ks = g >> 1;
        f[0] = *(const bf16x8*)(kst + (r32) * 128 + (((ks * 2 + hsel) ^ sw) * 16));
        f[1] = *(const bf16x8*)(kst + (32 + r32) * 128 + (((ks * 2 + hsel) ^ sw) * 16));
      } else {
        const int c = g >> 1, kb = (c >> 1) * 32 + 16 * (c & 1) + vkey;
        vt_frag_issue(vst, v_off(kb, vcol), v_off(kb + 8, vcol), vr[0], vr[1]);
        vt_frag_issue(vst, v_off(kb, 32 + vcol), v_off(kb + 8, 32 + vcol), vr[2], vr[3]);
      }
    };
    auto land_group = [&](int g, bf16x8 (&f)[2]) {        // V^T fragments: manual wait, then assemble
      if ((g & 1) == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0]), "+v"(vr[1]), "+v"(vr[2]), "+v"(vr[3]));
        f[0] = vt_frag_join(vr[0], vr[1]);
        f[1] = vt_frag_join(vr[2], vr[3]);
      }
    };
    float racc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float nm[2] = {-m_run[0], -m_run[1]};
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    load_group(0, fr[0]);
    land_group(0, fr[0]);
    float xarg[2] = {__builtin_fmaf(cur[0][0][0], kScaleLog2, nm[0]), __builtin_fmaf(cur[0][0][1], kScaleLog2, nm[0])};
    float pe_prev[2] = {0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g == 0) issue_full(t + 3);
      if (g >= 1) land_group(g, fr[g & 1]);
      if (g + 1 < 8) load_group(g + 1, fr[(g + 1) & 1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rb = j & 1, hi = j >> 1;
        if (g & 1) {          // scores of the next tile: fragment hi = key half kt, k-step ks = g >> 1
          const int ks = g >> 1;
          nxt[rb][hi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g & 1][hi], qf[rb][ks], ks == 0 ? zero : nxt[rb][hi], 0, 0, 0);
        } else {              // P.V of the previous tile: fragment hi = d half dt, chunk c = (kt, s)
          const int c = g >> 1;
          oacc[rb][hi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g & 1][hi], pprev[rb][c >> 1][c & 1], oacc[rb][hi], 0, 0, 0);
        }
        // vector stream, software-pipelined over the slots so that no instruction consumes a result of its own slot (one
        // wave per SIMD: nobody else fills a dependency stall): slot k scales the elements of slot k+1, exponentiates
        // those of slot k and sums those of slot k-1
        const int slot = g * 4 + j;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (slot >= 1) {
            const int i = (slot - 1) * 2 + u;
            racc[i >> 5][i & 3] += pe_prev[u];
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = slot * 2 + u, r = i >> 5, kt = (i >> 4) & 1, e = i & 15;
          const float pe = __builtin_amdgcn_exp2f(xarg[u]);
          cur[r][kt][e] = pe;
          pe_prev[u] = pe;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (slot + 1 < 32) {
            const int i = (slot + 1) * 2 + u, r = i >> 5, kt = (i >> 4) & 1, e = i & 15;
            xarg[u] = __builtin_fmaf(cur[r][kt][e], kScaleLog2, nm[r]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    racc[1][2] += pe_prev[0];                            // elements 62, 63 (slot 31)
    racc[1][3] += pe_prev[1];
    float rs[2];
    rs[0] = (racc[0][0] + racc[0][1]) + (racc[0][2] + racc[0][3]);
    rs[1] = (racc[1][0] + racc[1][1]) + (racc[1][2] + racc[1][3]);
    if (!__all(rs[0] <= kSumLimit && rs[1] <= kSumLimit)) rescale(cur, t, rs);
    l_run[0] += rs[0];
    l_run[1] += rs[1];
    pack_tile(cur, pprev);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // tile t+2 landed (t+3 may stay in flight)
    asm volatile("s_barrier" ::: "memory");
  };
  // general iteration (tail, ragged last tile): same pipeline, compiler-scheduled
  auto iteration = [&](f32x16 (&cur)[2][2], f32x16 (&nxt)[2][2], int t) {
    // entering: tiles <= t+1 landed and visible, t+2 in flight; the slot of tile t+3 (= tile t-5's) is long free
    if (t + 3 < nt) issue(t + 3);
    pv(pprev, t - 1);
    if (t + 1 < nt) scores(nxt, t + 1, std::false_type{});
    float rs[2];
    rs[0] = exp_sum(cur[0], m_run[0]);
    rs[1] = exp_sum(cur[1], m_run[1]);
    if (!__all(rs[0] <= kSumLimit && rs[1] <= kSumLimit)) rescale(cur, t, rs);
    l_run[0] += rs[0];
    l_run[1] += rs[1];
    pack_tile(cur, pprev);
    if (t + 2 < nt) {                // the next iteration reads K(t+2): own pieces retired, everyone's visible
      if (t + 3 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
  };
  int t = 1;
  for (; t + 4 < full_tiles; t += 2) {
    iteration_full(sa, sb, t);
    iteration_full(sb, sa, t + 1);
  }
  for (; t + 1 < nt; t += 2) {
    iteration(sa, sb, t);
    iteration(sb, sa, t + 1);
  }
  if (t < nt) { iteration(sa, sb, t); ++t; }
  if (nt > 0) pv(pprev, nt - 1);

#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const float inv = 1.0f / half_sum(l_run[rb]);
    if (qrow[rb] < S) {
      bf16_t* op = out + (row0 + qrow[rb]) * C + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dcol = dt * 32 + 8 * q + 4 * hsel;
          uint2 pk = make_uint2(pack_bf16x2(oacc[rb][dt][4 * q] * inv, oacc[rb][dt][4 * q + 1] * inv),
                                pack_bf16x2(oacc[rb][dt][4 * q + 2] * inv, oacc[rb][dt][4 * q + 3] * inv));
          *(uint2*)(op + dcol) = pk;
        }
    }
  }
}

}  // namespace

int ctrlv_attention_spatial_pipelined(const void* qkv, void* out, int n_img, int S, int C, hipStream_t stream) {
  static bool attr_set[CTRLV_MAX_DEVICES] = {};
  const int dev = ctrlv_current_device();
  constexpr int smem = kSlots * kSlotBytes;
  if (!attr_set[dev]) {
    CTRLV_HIP_TRY(hipFuncSetAttribute((const void*)attn_spatial64x_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    attr_set[dev] = true;
  }
  dim3 grid((S + 255) / 256, C / 64, n_img);
  hipLaunchKernelGGL(attn_spatial64x_kernel, grid, dim3(256), smem, stream, (const bf16_t*)qkv, (bf16_t*)out, S, C);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
