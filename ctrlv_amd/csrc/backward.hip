// Backward kernels -- first slice of the training path (BASELINE config 5, SURVEY.md 8 rows a11 / f3:
// tools/train_video_controlnet.py:451-488 needs ControlNet dgrad + wgrad and UNet up-path dgrad).
//
// What exists here: everything a SpatioTemporalResBlock and the ControlNet zero-convs need to backpropagate --
//   * DGRAD of the gather-GEMM family needs no new kernel: it is the forward kernel (ctrlv_gemm) on role-swapped weights
//     (Linear: W^T; 3x3 / temporal conv: taps reversed, channels transposed -- ctrlv_amd/autograd.py packs them);
//   * ctrlv_gemm_wgrad: dW[n][tap*Cin + c] += sum_m dY[m][n] * A[src(m, tap)][c]  (all gather modes of the forward GEMM);
//     the contraction runs over ROWS, so both MFMA operands are column reads of row-major tiles: the tiles are staged
//     row-major in LDS and read with ds_read_b64_tr_b16 (the attention kernel's V^T recipe); M is split across workgroups
//     and partial results are accumulated with fp32 atomics (dW must be zeroed by the caller);
//   * ctrlv_colsum: dbias[n] = sum_m dY[m][n], and the per-clip form for the row-vector operand V (temb / cross-attention
//     vectors): dV[idx(m)][n] += dY[m][n];
//   * GroupNorm(+SiLU) backward: split reduction like the forward (per-chunk per-channel partial sums -> per-group means
//     and dgamma / dbeta -> streaming dx pass), statistics (mean, rstd) re-used from the forward's table.
// This slice is written for correctness first (parity against torch.autograd on the oracle, tests/test_backward_gpu.py);
// attention / LayerNorm / GEGLU backward and the tuned (LDS-DMA, persistent) wgrad schedule are the next steps.
#include "common.h"

namespace {

__device__ __forceinline__ int tr_off(int row, int col) {      // [rows][64] bf16 tile, 128-B rows, tr-read swizzle
  const int chunk = (col >> 3) ^ (((row >> 1) & 1) << 2);
  return row * 128 + chunk * 16 + (col & 7) * 2;
}
// 32 (columns cb .. cb+31) x 16 (rows kb .. kb+15) MFMA operand from a row-major tile: two transposed 4x16 block reads
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int kb, int cb, int lane) {
  const int hsel = lane >> 5, i16 = lane & 15;
  const int row = kb + 4 * hsel + (i16 >> 2);
  const int col = cb + 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + tr_off(row, col)));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + tr_off(row + 8, col)));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

struct WgradArgs {
  const bf16_t* A; const bf16_t* A2; const bf16_t* dY; float* dW;
  int M, N, Cin, taps, lda, lda2, c_split, ldy, mode, H, Wd, Ho, Wo, stride, up, F, S, rows_per_slab;
};

// source row of output row m for tap t, or -1 (zero padding / frame edge)
__device__ __forceinline__ long wgrad_src(const WgradArgs& a, int m, int tap) {
  if (a.mode == 0) return m;
  if (a.mode == 2) {
    const int f = (m / a.S) % a.F, df = tap - 1;
    return (f + df >= 0 && f + df < a.F) ? (long)m + (long)df * a.S : -1;
  }
  const int hw = a.Ho * a.Wo;
  const int n_img = m / hw, rem = m - n_img * hw;
  const int yo = rem / a.Wo, xo = rem - yo * a.Wo;
  const int ky = tap / 3 - 1, kx = tap % 3 - 1;
  if (a.up) {       // nearest x2 fused: the conv runs on the (2H x 2W) grid, source pixel = coordinate >> 1
    const int yi = yo + ky, xi = xo + kx;
    if (yi < 0 || yi >= 2 * a.H || xi < 0 || xi >= 2 * a.Wd) return -1;
    return ((long)n_img * a.H + (yi >> 1)) * a.Wd + (xi >> 1);
  }
  const int yi = yo * a.stride + ky, xi = xo * a.stride + kx;
  if (yi < 0 || yi >= a.H || xi < 0 || xi >= a.Wd) return -1;
  return ((long)n_img * a.H + yi) * a.Wd + xi;
}

// grid (ceil(N/64), Ktot/64, slabs); 256 threads; output tile 64 (n) x 64 (k), one 32x32 MFMA accumulator per wave
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
  __shared__ __attribute__((aligned(1024))) char ty[4096];   // dY rows [32][64 n]
  __shared__ __attribute__((aligned(1024))) char ta[4096];   // A  rows [32][64 k]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int tap = k0 / a.Cin, c0 = k0 - tap * a.Cin;        // Cin % 64 == 0: a 64-wide K tile lies inside one tap
  const int m_lo = blockIdx.z * a.rows_per_slab;
  const int m_hi = min(a.M, m_lo + a.rows_per_slab);
  const int r = tid >> 3, ch = tid & 7;                      // this thread stages row r, 16-byte chunk ch of both tiles
  const int nh = wid & 1, kh = wid >> 1;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int sw_off = r * 128 + ((ch ^ (((r >> 1) & 1) << 2)) * 16);
  for (int m0 = m_lo; m0 < m_hi; m0 += 32) {
    const int m = m0 + r;
    uint4 vy = make_uint4(0, 0, 0, 0), va = make_uint4(0, 0, 0, 0);
    if (m < m_hi) {
      if (n0 + ch * 8 < a.N) vy = *(const uint4*)(a.dY + (long)m * a.ldy + n0 + ch * 8);
      const long src = wgrad_src(a, m, tap);
      if (src >= 0) {
        const int c = c0 + ch * 8;
        va = (a.A2 != nullptr && c >= a.c_split) ? *(const uint4*)(a.A2 + src * a.lda2 + (c - a.c_split))
                                                 : *(const uint4*)(a.A + src * a.lda + c);
      }
    }
    __syncthreads();                                         // previous chunk's fragment reads are done
    *(uint4*)(ty + sw_off) = vy;
    *(uint4*)(ta + sw_off) = va;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 fy = tr_frag(ty, 16 * s, 32 * nh, lane);  // A operand: i = n
      const bf16x8 fa = tr_frag(ta, 16 * s, 32 * kh, lane);  // B operand: j = k
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy, fa, acc, 0, 0, 0);
    }
  }
  // D[i][j]: lane holds column j = lane % 32, rows i = (e & 3) + 8 * (e >> 2) + 4 * (lane / 32)
  const int j = k0 + 32 * kh + (lane & 31);
  const long ktot = (long)a.taps * a.Cin;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int n = n0 + 32 * nh + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    if (n < a.N) atomicAdd(a.dW + (long)n * ktot + j, acc[e]);
  }
}

// out[idx(m)][n] += sum over this block's rows of x[m][n];  idx(m) = vmode ? (m / vdiv) % vmod : 0
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int M, int N, int ldx, int rows_per_block,
                                                     int vmode, int vdiv, int vmod, float scale, float* __restrict__ out,
                                                     int ldo) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
  if (n >= N) return;
  float acc = 0.f;
  int cur = -1;
  for (int m = m_lo; m < m_hi; ++m) {
    const int idx = vmode ? (m / vdiv) % vmod : 0;
    if (idx != cur) {
      if (cur >= 0) atomicAdd(out + (long)cur * ldo + n, acc * scale);
      cur = idx;
      acc = 0.f;
    }
    acc += bf16_to_f32(x[(long)m * ldx + n]);
  }
  if (cur >= 0) atomicAdd(out + (long)cur * ldo + n, acc * scale);
}

// out[0] += scale * sum_i dy[i] * (p[i] - q[i])      (gradient of a folded AlphaBlender's mixing weight)
__global__ __launch_bounds__(256) void dot_diff_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ p,
                                                       const bf16_t* __restrict__ q, size_t n, float scale,
                                                       float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  const size_t nv = n >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    float a[8], b[8], c[8];
    unpack_bf16x8(((const uint4*)dy)[i], a);
    unpack_bf16x8(((const uint4*)p)[i], b);
    unpack_bf16x8(((const uint4*)q)[i], c);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += a[e] * (b[e] - c[e]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const size_t i = (nv << 3) + threadIdx.x;
    acc += bf16_to_f32(dy[i]) * (bf16_to_f32(p[i]) - bf16_to_f32(q[i]));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale);
}

// ------------------------------------------------------------------------------------------------ GroupNorm backward
struct GnB { int n_img, S, C, ips, rows_per_chunk, n_chunks; };

__device__ __forceinline__ float dsilu(float z) {            // d/dz [z * sigmoid(z)]
  const float sg = 1.0f / (1.0f + __expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}

// pass 1: per (image, chunk, channel): s1 = sum dz * xhat, s2 = sum dz      grid (n_chunks, n_img), C/8 * RPP threads
__global__ void gn_bwd_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, GnB s,
                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, int silu, float* __restrict__ part) {
  extern __shared__ float red[];      // [RPP][C][2]
  const int CV = s.C / 8, RPP = blockDim.x / CV;
  const int tid = threadIdx.x, col = tid % CV, rsub = tid / CV;
  const int n = blockIdx.y, chunk = blockIdx.x, stat = n / s.ips, cpg = s.C / 32, c0 = col * 8;
  float mean[8], rstd[8], g[8], b[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int gi = (c0 + e) / cpg;
    mean[e] = stats[((long)stat * 32 + gi) * 2];
    rstd[e] = stats[((long)stat * 32 + gi) * 2 + 1];
    g[e] = gamma[c0 + e];
    b[e] = beta[c0 + e];
    s1[e] = s2[e] = 0.f;
  }
  const int r0 = chunk * s.rows_per_chunk, r1 = min(s.S, r0 + s.rows_per_chunk);
  for (int r = r0 + rsub; r < r1; r += RPP) {
    const long row = (long)n * s.S + r;
    float fx[8], fd[8];
    unpack_bf16x8(*(const uint4*)(x + row * s.C + c0), fx);
    unpack_bf16x8(*(const uint4*)(dy + row * s.C + c0), fd);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (fx[e] - mean[e]) * rstd[e];
      const float dz = silu ? fd[e] * dsilu(xh * g[e] + b[e]) : fd[e];
      s1[e] += dz * xh;
      s2[e] += dz;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[((rsub * s.C) + c0 + e) * 2] = s1[e];
    red[((rsub * s.C) + c0 + e) * 2 + 1] = s2[e];
  }
  __syncthreads();
  for (int c = tid; c < s.C; c += blockDim.x) {
    float a1 = 0.f, a2 = 0.f;
    for (int rs = 0; rs < RPP; ++rs) { a1 += red[(rs * s.C + c) * 2]; a2 += red[(rs * s.C + c) * 2 + 1]; }
    const long o = (((long)n * s.n_chunks + chunk) * s.C + c) * 2;
    part[o] = a1;
    part[o + 1] = a2;
  }
}

// pass 2a: per (statistics row, group): m1 = mean(dz * gamma), m2 = mean(dz * gamma * xhat)    grid n_stat, 256 threads
__global__ __launch_bounds__(256) void gn_bwd_group_kernel(GnB s, const float* __restrict__ part,
                                                           const float* __restrict__ gamma, float* __restrict__ gmean) {
  __shared__ double dred[8][32][2];
  const int tid = threadIdx.x, stat = blockIdx.x, g = tid & 31, sl = tid >> 5, cpg = s.C / 32;
  const int tot = s.ips * s.n_chunks;              // (image, chunk) pairs of this statistics row, contiguous in `part`
  double a1 = 0.0, a2 = 0.0;
  for (int k = sl; k < tot; k += 8)
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      const float* p = part + (((long)stat * tot + k) * s.C + c) * 2;
      a1 += (double)gamma[c] * (double)p[1];
      a2 += (double)gamma[c] * (double)p[0];
    }
  dred[sl][g][0] = a1;
  dred[sl][g][1] = a2;
  __syncthreads();
  if (tid < 32) {
    a1 = a2 = 0.0;
    for (int k = 0; k < 8; ++k) { a1 += dred[k][tid][0]; a2 += dred[k][tid][1]; }
    const double cnt = (double)cpg * s.S * s.ips;
    gmean[((long)stat * 32 + tid) * 2] = (float)(a1 / cnt);
    gmean[((long)stat * 32 + tid) * 2 + 1] = (float)(a2 / cnt);
  }
}
// pass 2b: dgamma[c] += sum s1, dbeta[c] += sum s2 over every (image, chunk)               grid ceil(C/256)
__global__ __launch_bounds__(256) void gn_bwd_affine_kernel(GnB s, const float* __restrict__ part, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= s.C) return;
  double a1 = 0.0, a2 = 0.0;
  const long tot = (long)s.n_img * s.n_chunks;
  for (long k = 0; k < tot; ++k) {
    a1 += (double)part[(k * s.C + c) * 2];
    a2 += (double)part[(k * s.C + c) * 2 + 1];
  }
  dgamma[c] += (float)a1;
  dbeta[c] += (float)a2;
}
// pass 3: dx = rstd * (dz * gamma - m1 - xhat * m2)
__global__ void gn_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, GnB s,
                                    const float* __restrict__ stats, const float* __restrict__ gmean,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                    bf16_t* __restrict__ dx) {
  const int CV = s.C / 8, RPP = blockDim.x / CV;
  const int tid = threadIdx.x, col = tid % CV, rsub = tid / CV;
  const int n = blockIdx.y, chunk = blockIdx.x, stat = n / s.ips, cpg = s.C / 32, c0 = col * 8;
  float mean[8], rstd[8], g[8], b[8], m1[8], m2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int gi = (c0 + e) / cpg;
    mean[e] = stats[((long)stat * 32 + gi) * 2];
    rstd[e] = stats[((long)stat * 32 + gi) * 2 + 1];
    m1[e] = gmean[((long)stat * 32 + gi) * 2];
    m2[e] = gmean[((long)stat * 32 + gi) * 2 + 1];
    g[e] = gamma[c0 + e];
    b[e] = beta[c0 + e];
  }
  const int r0 = chunk * s.rows_per_chunk, r1 = min(s.S, r0 + s.rows_per_chunk);
  for (int r = r0 + rsub; r < r1; r += RPP) {
    const long row = (long)n * s.S + r;
    float fx[8], fd[8];
    unpack_bf16x8(*(const uint4*)(x + row * s.C + c0), fx);
    unpack_bf16x8(*(const uint4*)(dy + row * s.C + c0), fd);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (fx[e] - mean[e]) * rstd[e];
      const float dz = silu ? fd[e] * dsilu(xh * g[e] + b[e]) : fd[e];
      fx[e] = rstd[e] * (dz * g[e] - m1[e] - xh * m2[e]);
    }
    *(uint4*)(dx + row * s.C + c0) = pack_bf16x8(fx);
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// One wave per row (grid-stride), 8 * NV columns per lane like the forward.  Recomputes mean / rstd of x (+ V row), then
//   dz = dy * gamma;  dx = rstd * (dz - mean(dz) - xhat * mean(dz * xhat));  dgamma += dy * xhat;  dbeta += dy
// (dgamma / dbeta: per-lane column partials over the wave's rows, one atomic per column per wave at the end).
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, int M, int C,
                                                     const float* __restrict__ gamma, float eps, const float* __restrict__ V,
                                                     int vdiv, int vmod, int ldv, bf16_t* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int CV = C >> 3;
  float g[NV][8], ag[NV][8], ab[NV][8];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cv = lane + k * 64;
      g[k][e] = cv < CV ? gamma[cv * 8 + e] : 0.f;
      ag[k][e] = ab[k][e] = 0.f;
    }
  const float inv_c = 1.0f / (float)C;
  for (long m = (long)blockIdx.x * 4 + wid; m < M; m += (long)gridDim.x * 4) {
    const float* vrow = V ? V + (long)((m / vdiv) % vmod) * ldv : nullptr;
    float fx[NV][8], fd[NV][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
        unpack_bf16x8(*(const uint4*)(x + m * C + cv * 8), fx[k]);
        unpack_bf16x8(*(const uint4*)(dy + m * C + cv * 8), fd[k]);
        if (vrow) {
#pragma unroll
          for (int e = 0; e < 8; ++e) fx[k][e] += vrow[cv * 8 + e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) fx[k][e] = fd[k][e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += fx[k][e];
    }
    const float mean = wave_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = fx[k][e] - mean; q += d * d; }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_c + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (fx[k][e] - mean) * rstd;        // (columns past C: fd = g = 0 contribute nothing)
        const float dz = fd[k][e] * g[k][e];
        fx[k][e] = xh;
        s1 += dz;
        s2 += dz * xh;
        ag[k][e] += fd[k][e] * xh;
        ab[k][e] += fd[k][e];
      }
    const float m1 = wave_sum(s1) * inv_c, m2 = wave_sum(s2) * inv_c;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (fd[k][e] * g[k][e] - m1 - fx[k][e] * m2);
        *(uint4*)(dx + m * C + cv * 8) = pack_bf16x8(o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int cv = lane + k * 64;
    if (cv < CV) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        atomicAdd(dgamma + cv * 8 + e, ag[k][e]);
        atomicAdd(dbeta + cv * 8 + e, ab[k][e]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ GEGLU backward
// raw: the projection output [M][2I] in the packed (16-value, 16-gate) column-block order (the forward GEMM without its
// GEGLU epilogue); du: gradient of u = a * gelu(g), [M][I].  draw (same layout as raw): da = du * gelu(g),
// dg = du * a * (Phi(g) + g * phi(g)).
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ raw, const bf16_t* __restrict__ du, long M,
                                                        int I, bf16_t* __restrict__ draw) {
  const long total = M * (I >> 3);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (I >> 3);
    const int j0 = (int)(i % (I >> 3)) * 8;            // 8 consecutive u columns: inside one 16-column block
    const int blk = j0 >> 4, r0 = j0 & 15;
    const long base = m * (2L * I) + blk * 32 + r0;
    float a[8], g[8], d[8], oa[8], og[8];
    unpack_bf16x8(*(const uint4*)(raw + base), a);
    unpack_bf16x8(*(const uint4*)(raw + base + 16), g);
    unpack_bf16x8(*(const uint4*)(du + m * I + j0), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const f32x4_t gv = {g[e], g[e], g[e], g[e]};
      const float Phi = gelu_phi4(gv).x;
      const float phi = 0.3989422804014327f * __expf(-0.5f * g[e] * g[e]);
      oa[e] = d[e] * (g[e] * Phi);
      og[e] = d[e] * a[e] * (Phi + g[e] * phi);
    }
    *(uint4*)(draw + base) = pack_bf16x8(oa);
    *(uint4*)(draw + base + 16) = pack_bf16x8(og);
  }
}

}  // namespace

extern "C" int ctrlv_layernorm_bwd(const void* x, const void* dy, int M, int C, const float* gamma, float eps, const float* V,
                                   int vdiv, int vmod, int ldv, void* dx, float* dgamma, float* dbeta,
                                   ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && dy && gamma && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
  CTRLV_CHECK_SHAPE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "layernorm_bwd: C=%d must be a multiple of 8, <= 2048", C);
  if (V) CTRLV_CHECK_ARG(vdiv > 0 && vmod > 0 && ldv >= C, "layernorm_bwd: bad row-vector table");
  const int nv = (C / 8 + 63) / 64;
  long blocks = ((long)M + 3) / 4;
  if (blocks > 256 * 4) blocks = 256 * 4;
  hipStream_t st = (hipStream_t)stream;
#define LNB_LAUNCH(NV)                                                                                                  \
  hipLaunchKernelGGL(ln_bwd_kernel<NV>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, M, C, \
                     gamma, eps, V, vdiv, vmod, ldv, (bf16_t*)dx, dgamma, dbeta)
  switch (nv) {
    case 1: LNB_LAUNCH(1); break;
    case 2: LNB_LAUNCH(2); break;
    case 3: LNB_LAUNCH(3); break;
    default: LNB_LAUNCH(4); break;
  }
#undef LNB_LAUNCH
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_geglu_bwd(const void* raw, const void* du, size_t M, int I, void* draw, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(raw && du && draw, "geglu_bwd: null pointer");
  CTRLV_CHECK_SHAPE(M > 0 && I > 0 && I % 16 == 0, "geglu_bwd: inner dim %d must be a multiple of 16", I);
  size_t blocks = (M * (I / 8) + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)raw,
                     (const bf16_t*)du, (long)M, I, (bf16_t*)draw);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_gemm_wgrad(const ctrlv_gemm_desc* dp, const void* dY, int ldy, float* dW, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(dp && dY && dW, "gemm_wgrad: null pointer");
  const ctrlv_gemm_desc& d = *dp;
  CTRLV_CHECK_ARG(d.A != nullptr, "gemm_wgrad: A must be non-null");
  CTRLV_CHECK_SHAPE(d.M > 0 && d.N > 0 && d.Cin > 0 && d.Cin % 64 == 0, "gemm_wgrad: Cin=%d must be a positive multiple of 64", d.Cin);
  CTRLV_CHECK_SHAPE(d.N % 8 == 0 && ldy % 8 == 0 && d.lda % 8 == 0, "gemm_wgrad: N, ldy, lda must be multiples of 8");
  CTRLV_CHECK_SHAPE((d.mode == 0 && d.taps == 1) || (d.mode == 1 && d.taps == 9) || (d.mode == 2 && d.taps == 3),
                    "gemm_wgrad: mode / taps mismatch");
  if (d.A2) CTRLV_CHECK_SHAPE(d.c_split % 64 == 0 && d.lda2 % 8 == 0, "gemm_wgrad: bad concat split");
  WgradArgs a;
  a.A = (const bf16_t*)d.A; a.A2 = (const bf16_t*)d.A2; a.dY = (const bf16_t*)dY; a.dW = dW;
  a.M = d.M; a.N = d.N; a.Cin = d.Cin; a.taps = d.taps; a.lda = d.lda; a.lda2 = d.lda2; a.c_split = d.c_split; a.ldy = ldy;
  a.mode = d.mode; a.H = d.H; a.Wd = d.Wd; a.Ho = d.Ho; a.Wo = d.Wo; a.stride = d.stride ? d.stride : 1; a.up = d.up;
  a.F = d.F; a.S = d.S;
  const int ktiles = d.taps * d.Cin / 64, ntiles = (d.N + 63) / 64;
  // enough M slabs to fill the chip (~1024 workgroups), each a multiple of 32 rows
  int slabs = 1024 / (ktiles * ntiles);
  if (slabs < 1) slabs = 1;
  int rps = ((d.M + slabs - 1) / slabs + 31) / 32 * 32;
  if (rps < 32) rps = 32;
  slabs = (d.M + rps - 1) / rps;
  a.rows_per_slab = rps;
  CTRLV_CHECK_SHAPE(slabs <= 65535 && ktiles <= 65535, "gemm_wgrad: grid too large");
  hipLaunchKernelGGL(wgrad_kernel, dim3(ntiles, ktiles, slabs), dim3(256), 0, (hipStream_t)stream, a);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_colsum(const void* x, int M, int N, int ldx, int vmode, int vdiv, int vmod, float scale, float* out,
                            int ldo, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && out, "colsum: null pointer");
  CTRLV_CHECK_SHAPE(M > 0 && N > 0 && ldx >= N, "colsum: bad shape");
  CTRLV_CHECK_ARG(vmode == 0 || (vmode == 1 && vdiv > 0 && vmod > 0), "colsum: vmode must be 0 or 1 with vdiv, vmod > 0");
  const int rpb = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, M, N, ldx, rpb, vmode, vdiv, vmod, scale, out, ldo);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_dot_diff(const void* dy, const void* p, const void* q, size_t n, float scale, float* out,
                              ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(dy && p && q && out, "dot_diff: null pointer");
  size_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(dot_diff_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)p, (const bf16_t*)q, n, scale, out);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_groupnorm_bwd_scratch_floats(int n_img, int S, int C, int imgs_per_stat) {
  const int chunks = ctrlv_groupnorm_chunks(n_img, S, C, imgs_per_stat);
  if (chunks < 0) return chunks;
  const long need = (long)n_img * chunks * C * 2 + (long)(n_img / imgs_per_stat) * 64;
  CTRLV_CHECK_SHAPE(need < (1L << 31), "groupnorm_bwd: scratch too large");
  return (int)need;
}

extern "C" int ctrlv_groupnorm_bwd(const void* x, const void* dy, int n_img, int S, int C, int imgs_per_stat,
                                   const float* fwd_partials, const float* gamma, const float* beta, int silu, void* dx,
                                   float* dgamma, float* dbeta, float* scratch, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && dy && fwd_partials && gamma && beta && dx && dgamma && dbeta && scratch, "groupnorm_bwd: null pointer");
  const int chunks = ctrlv_groupnorm_chunks(n_img, S, C, imgs_per_stat);
  if (chunks < 0) return chunks;
  CTRLV_CHECK_SHAPE(n_img % imgs_per_stat == 0, "groupnorm_bwd: n_img %% imgs_per_stat != 0");
  GnB s;
  s.n_img = n_img; s.S = S; s.C = C; s.ips = imgs_per_stat; s.n_chunks = chunks;
  s.rows_per_chunk = (S + chunks - 1) / chunks;       // own chunking; only the (mean, rstd) table of the forward is reused
  const float* stats = fwd_partials + (size_t)n_img * chunks * 64;       // (mean, rstd) table behind the forward partials
  float* part = scratch;
  float* gmean = scratch + (size_t)n_img * chunks * C * 2;
  const int CV = C / 8, RPP = CV >= 256 ? 1 : 256 / CV, nt = CV * RPP;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(chunks, n_img), dim3(nt), (size_t)RPP * C * 2 * sizeof(float), st,
                     (const bf16_t*)x, (const bf16_t*)dy, s, stats, gamma, beta, silu, part);
  CTRLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(n_img / imgs_per_stat), dim3(256), 0, st, s, part, gamma, gmean);
  CTRLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, st, s, part, dgamma, dbeta);
  CTRLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(chunks, n_img), dim3(nt), 0, st, (const bf16_t*)x, (const bf16_t*)dy, s, stats,
                     gmean, gamma, beta, silu, (bf16_t*)dx);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
