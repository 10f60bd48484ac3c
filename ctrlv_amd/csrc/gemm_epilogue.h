// Shared epilogue of the gather-GEMM kernels (see gemm.hip header).  Each lane owns output row m of every 32x32 MFMA
// sub-tile (weight tile = A operand) and accumulator quad q holds columns n0 + 8q + 4*(lane>>5) + {0..3}.
//   v = s_acc*acc + s1*R1[m,n] + s2*R2[m,n] + V[vidx(m), n]   (acc already contains the bias: gemm.hip starts from it);  optional SiLU; optional GEGLU pairing.
#pragma once
#include "common.h"

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const ctrlv_gemm_desc& d, f32x16 (&acc)[TM][TN], int bm, int bn,
                                              int wr, int wc, int WTM, int WTN, int r32, int hsel,
                                              const char* gelu_tab) {
  // ---- epilogue: lane owns row m; accumulator quad q holds columns n0 + 8q + 4h + {0..3}
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = bm + wr * WTM + i * 32 + r32;
    if (m >= d.M) continue;
    const float* vrow = nullptr;
    if (d.vmode == 1) {
      vrow = d.V + (long)((m / d.vdiv) % d.vmod) * d.ldv;
    } else if (d.vmode == 2) {
      vrow = d.V + (long)(((long)(m / d.vdiv) * d.vS + (m % d.vS)) % d.vmod) * d.ldv;
    }
    if (d.geglu) {
      // weight rows are interleaved in 16-row (value, gate) blocks: quads 0,1 of a sub-tile are values, quads 2,3 gates
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int ncol = bn + wc * WTN + j * 32 + 8 * q + 4 * hsel;  // value column in the interleaved weight order
          if (ncol >= d.N) continue;
          const int ocol = ((bn + wc * WTN + j * 32) >> 1) + 8 * q + 4 * hsel;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = acc[i][j][4 * q + e], g = acc[i][j][4 * (q + 2) + e];
            o[e] = geglu_tab(a, g, gelu_tab);
          }
          if (ocol < d.n_store) {
            uint2 pk = make_uint2(pack_elx2(o[0], o[1]), pack_elx2(o[2], o[3]));
            *(uint2*)((el_t*)d.out + (long)m * d.ldo + ocol) = pk;
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ncol = bn + wc * WTN + j * 32 + 8 * q + 4 * hsel;
          if (ncol >= d.N || ncol >= d.n_store) continue;
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * q + e];
          // The same operation sequence as the ping-pong kernels' epilogue (v_pk_mul_f32, then one v_pk_fma_f32 per
          // residual): t = acc * s_acc ROUNDED, then fma(s, R, t).  Left to the contraction heuristics the compiler
          // formed fma(acc, s_acc, s1 * R) here -- one fp32 ulp apart, which after the bf16 store made a layer served by
          // this kernel (small M) and by the ping-pong kernel (large M) differ in an occasional last bit.
          {
#pragma clang fp contract(off)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = o[e] * (ncol < d.n_scale2 ? d.s_acc2 : d.s_acc);
          }
          if (d.R1) {
            const uint2 rv = *(const uint2*)((const el_t*)d.R1 + (long)m * d.ldr1 + ncol);
            o[0] = __builtin_fmaf(d.s1, el_lo_f32(rv.x), o[0]);
            o[1] = __builtin_fmaf(d.s1, el_hi_f32(rv.x), o[1]);
            o[2] = __builtin_fmaf(d.s1, el_lo_f32(rv.y), o[2]);
            o[3] = __builtin_fmaf(d.s1, el_hi_f32(rv.y), o[3]);
            if (d.R1_lo) {      // SPLIT trunk operand: R1 + R1_lo, one fma per plane (the ping-pong epilogue's sequence)
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
          if (vrow) {
            const float4 vv = *(const float4*)(vrow + ncol);
            o[0] += vv.x; o[1] += vv.y; o[2] += vv.z; o[3] += vv.w;
          }
          if (d.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = silu_f(o[e]);
          }
          if (d.out_f32 == 1) {
            *(float4*)((float*)d.out + (long)m * d.ldo + ncol) = make_float4(o[0], o[1], o[2], o[3]);
          } else {
            uint2 pk = make_uint2(pack_elx2(o[0], o[1]), pack_elx2(o[2], o[3]));
            if (d.out_f32 & 2) asm volatile("" ::"v"(pk.x), "v"(pk.y));   // profiling aid: compute, do not store
            else *(uint2*)((el_t*)d.out + (long)m * d.ldo + ncol) = pk;
            if (d.out_lo) {     // SPLIT output: lo = rne(v - hi)  (common.h split_lo8, four elements)
#pragma clang fp contract(off)
              const float l0 = o[0] - el_lo_f32(pk.x), l1 = o[1] - el_hi_f32(pk.x), l2 = o[2] - el_lo_f32(pk.y),
                          l3 = o[3] - el_hi_f32(pk.y);
              *(unsigned*)((lo_t*)d.out_lo + (long)m * d.ldo + ncol) = pack_lo4(l0, l1, l2, l3);   // (one byte per element)
            }
          }
        }
      }
    }
  }
}
