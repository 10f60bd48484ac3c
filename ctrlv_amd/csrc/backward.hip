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
#include "wgrad_pp.h"

namespace {

__device__ __forceinline__ int tr_off(int row, int col) {      // [rows][64] bf16 tile, 128-B rows, tr-read swizzle
  const int chunk = (col >> 3) ^ (((row >> 1) & 1) << 2);
  return row * 128 + chunk * 16 + (col & 7) * 2;
}
// 32 (columns cb .. cb+31) x 16 (rows kb .. kb+15) MFMA operand from a row-major tile: two transposed 4x16 block reads
__device__ __forceinline__ elx8 tr_frag(const char* tile, int kb, int cb, int lane) {
  const int hsel = lane >> 5, i16 = lane & 15;
  const int row = kb + 4 * hsel + (i16 >> 2);
  const int col = cb + 16 * ((lane >> 4) & 1) + 4 * (i16 & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + tr_off(row, col)));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + tr_off(row + 8, col)));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(elx8, v);
}

struct WgradArgs {
  const el_t* A; const el_t* A2; const el_t* dY; float* dW; float* dbias; float scale; int torch_layout;
  float* part;        // deterministic mode: slab partials [slabs][N][Ktot] fp32 (+ [slabs][N] bias sums behind them), else null
  int M, N, Cin, taps, lda, lda2, c_split, ldy, mode, H, Wd, Ho, Wo, stride, up, F, S, rows_per_slab;
  int ntiles, ktiles, slabs;     // grid decomposition (the launch is one-dimensional: wgrad_kernel deals it out XCD-aware)
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

// grid (ceil(N/128), ceil(Ktot/256), slabs); 256 threads = 2 (n) x 2 (k) waves; output tile 128 (n) x 256 (k), a wave owns
// 64 x 128 = 2 x 4 MFMA accumulators (6 transposed fragment reads per 8 MFMAs).  Rows advance in chunks of 64: the
// next chunk's global loads (4 dY + 8 A pieces of 16 B per thread) are issued into registers before the current chunk
// is consumed from LDS.  LDS: dY as 2 panels, A as 4 panels of [64 rows][64 columns] (the tr-read layout of tr_off);
// every 64-column K panel lies inside one tap (Cin % 64 == 0), so the tap / channel offset is a per-thread constant.
// (The first version -- 64 x 64 tile, one accumulator per wave, no prefetch -- ran at ~10 % of the MFMA peak and was
// 12.6 % of the cfg5 training step.)
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
  __shared__ __attribute__((aligned(1024))) char ty[2 * 8192];   // dY rows: 2 panels [64][64 n]
  __shared__ __attribute__((aligned(1024))) char ta[4 * 8192];   // A  rows: 4 panels [64][64 k]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // XCD-aware order (round 5): the workgroups of ONE row slab -- its (n tile, k tile) pairs, which all stream the same rows of
  // dY and A -- are dealt to ONE XCD (consecutive ids after common.h xcd_remap), so a slab's rows are fetched into one L2
  // instead of all eight.  With the hardware's round-robin placement every XCD pulled every slab through the fabric: at 87
  // FLOP per staged byte (128 x 256 tile, 64-row chunks) that is what the kernel ran at (537 TFLOP/s ~ 6 TB/s).
  const int ntiles_ = a.ntiles, ktiles_ = a.ktiles;
  const int lin = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int bx = lin % ntiles_, by = (lin / ntiles_) % ktiles_, bz = lin / (ntiles_ * ktiles_);
  const int n0 = bx * 128, k0 = by * 256;
  const long ktot = (long)a.taps * a.Cin;
  const int m_lo = bz * a.rows_per_slab;
  const int m_hi = min(a.M, m_lo + a.rows_per_slab);
  const int nh = wid & 1, kh = wid >> 1;
  // staging roles: dY piece i of this thread = row (tid >> 4) + 16 i, 16-B chunk (tid & 15) of 16;
  //                A  piece i                = row (tid >> 5) +  8 i, 16-B chunk (tid & 31) of 32
  const int yr = tid >> 4, yc = tid & 15, ar = tid >> 5, ac = tid & 31;
  const int ycol = n0 + yc * 8;
  const bool y_ok = ycol < a.N;
  const int kcol = k0 + ac * 8;                              // global K column of this thread's A pieces
  const bool a_ok = kcol < ktot;
  const int tap = a_ok ? kcol / a.Cin : 0, c = kcol - tap * a.Cin;
  const bool second = a.A2 != nullptr && c >= a.c_split;
  const el_t* abase = second ? a.A2 + (c - a.c_split) : a.A + c;
  const long ald = second ? a.lda2 : a.lda;
  auto sw = [](int r, int ch) { return r * 128 + ((ch ^ (((r >> 1) & 1) << 2)) * 16); };
  // bias gradient (column sums of dY) rides along in the workgroups of the first K tile: they stream dY anyway
  const bool do_bias = a.dbias != nullptr && by == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  uint4 vy[4], va[8];
  auto load = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + yr + 16 * i;
      vy[i] = (y_ok && m < m_hi) ? *(const uint4*)(a.dY + (long)m * a.ldy + ycol) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + ar + 8 * i;
      va[i] = make_uint4(0, 0, 0, 0);
      if (a_ok && m < m_hi) {
        const long src = wgrad_src(a, m, tap);
        if (src >= 0) va[i] = *(const uint4*)(abase + src * ald);
      }
    }
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  load(m_lo);
  for (int m0 = m_lo; m0 < m_hi; m0 += 64) {
    __syncthreads();                                         // previous chunk's fragment reads are done
#pragma unroll
    for (int i = 0; i < 4; ++i) *(uint4*)(ty + (yc >> 3) * 8192 + sw(yr + 16 * i, yc & 7)) = vy[i];
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float f[8];
        unpack_elx8(vy[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += f[e];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) *(uint4*)(ta + (ac >> 3) * 8192 + sw(ar + 8 * i, ac & 7)) = va[i];
    __syncthreads();
    if (m0 + 64 < m_hi) load(m0 + 64);                       // in flight while this chunk is consumed
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      elx8 fy[2], fa[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) fy[i] = tr_frag(ty + nh * 8192, 16 * s, 32 * i, lane);          // A operand: i = n
#pragma unroll
      for (int j = 0; j < 4; ++j) fa[j] = tr_frag(ta + (kh * 2 + (j >> 1)) * 8192, 16 * s, 32 * (j & 1), lane);   // B: j = k
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_32x32x16(fy[i], fa[j], acc[i][j]);
    }
  }
  // D[i][j]: lane holds column j = lane % 32, rows i = (e & 3) + 8 * (e >> 2) + 4 * (lane / 32)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long k = (long)k0 + 128 * kh + 32 * j + (lane & 31);
      if (k >= ktot) continue;
      long col = k;                                          // packed (tap-major) K order ...
      if (a.torch_layout) {                                  // ... or the parameter's own [N][Cin][taps] layout
        const int tp = (int)(k / a.Cin);
        col = (k - (long)tp * a.Cin) * a.taps + tp;
      }
      if (a.part) {     // deterministic: this workgroup's own tile of its slab's partial matrix (packed K order, unscaled)
        float* ps = a.part + (long)bz * a.N * ktot;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = n0 + 64 * nh + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (n < a.N) ps[(long)n * ktot + k] = acc[i][j][e];
        }
        continue;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + 64 * nh + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (n < a.N) atomicAdd(a.dW + (long)n * ktot + col, acc[i][j][e] * a.scale);
      }
    }
  if (do_bias) {                                             // fold the 16 row-threads of every column group through LDS
    __syncthreads();
    float* red = (float*)ta;                                 // [16 row threads][128 columns]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[yr * 128 + yc * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && n0 + tid < a.N) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 128 + tid];
      if (a.part) a.part[(long)a.slabs * a.N * ktot + (long)bz * a.N + n0 + tid] = t;
      else atomicAdd(a.dbias + n0 + tid, t * a.scale);
    }
  }
}

// Deterministic wgrad, second kernel: dW (+ dbias) += scale * the slab partials summed IN SLAB ORDER, one thread per (n, 4 k)
// -- a single writer per element, so the accumulated gradient has the same bits in every run (the atomics of the one-kernel
// form arrive in a different order every time).  Also does the [N][Cin][taps] scatter of torch_layout.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int slabs, int N, long ktot, int Cin,
                                                           int taps, int torch_layout, float scale, float* __restrict__ dW,
                                                           float* __restrict__ dbias, int assign, int bias_rows) {
  const long k4 = ktot >> 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx < (long)N * k4) {
    const long n = idx / k4, k = (idx - n * k4) * 4;
    const float* p = part + n * ktot + k;
    float4 a = *(const float4*)p;
    for (int sidx = 1; sidx < slabs; ++sidx) {
      const float4 b = *(const float4*)(p + (long)sidx * N * ktot);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (!torch_layout) {
      float4* o = (float4*)(dW + n * ktot + k);
      float4 v = assign ? make_float4(0.f, 0.f, 0.f, 0.f) : *o;
      v.x += a.x * scale; v.y += a.y * scale; v.z += a.z * scale; v.w += a.w * scale;
      *o = v;
    } else {
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long kk = k + e;
        const int tp = (int)(kk / Cin);
        float* o = dW + n * ktot + (kk - (long)tp * Cin) * taps + tp;
        *o = (assign ? 0.f : *o) + av[e] * scale;
      }
    }
  }
  if (dbias && idx < N) {
    const float* pb = part + (long)slabs * N * ktot;
    float t = 0.f;
    for (int sidx = 0; sidx < bias_rows; ++sidx) t += pb[(long)sidx * N + idx];      // (wgrad_pp: one row per (slab, k tile))
    dbias[idx] = (assign ? 0.f : dbias[idx]) + t * scale;
  }
}

// out[idx(m)][n] += sum over this block's rows of x[m][n];  idx(m) = vmode ? (m / vdiv) % vmod : 0
// 64 column groups of 8 (16-B loads) x 4 row lanes per workgroup; the row lanes are folded through LDS so that a
// workgroup issues ONE atomic per column (same-address float atomics execute serially at the memory side: the first
// version -- one column per thread, 2-B loads, an atomic per 256 rows -- cost 124 us per call on average).
// `part` (deterministic mode, non-null): the block writes its sums to part[blockIdx.y][N] instead (the launcher makes
// rows_per_block divide vdiv, so a block never straddles two table rows) and colsum_reduce_kernel adds them in block order.
__global__ __launch_bounds__(256) void colsum_kernel(const el_t* __restrict__ x, int M, int N, int ldx, int rows_per_block,
                                                     int vmode, int vdiv, int vmod, float scale, float* __restrict__ out,
                                                     int ldo, float* __restrict__ part) {
  __shared__ float red[4][64][8];
  __shared__ int red_idx[4];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 64 + tx) * 8;
  const int m_lo = blockIdx.y * rows_per_block, m_hi = min(M, m_lo + rows_per_block);
  const bool col_ok = n0 < N;                       // (N % 8 == 0)
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int cur = -1;
  auto flush = [&](int idx) {
    if (idx >= 0 && col_ok) {
      if (part) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[(long)blockIdx.y * N + n0 + e] = acc[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(out + (long)idx * ldo + n0 + e, acc[e] * scale);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  };
  for (int m = m_lo + ty; m < m_hi; m += 4) {
    const int idx = vmode ? (m / vdiv) % vmod : 0;
    if (idx != cur) {
      flush(cur);
      cur = idx;
    }
    if (col_ok) {
      float f[8];
      unpack_elx8(*(const uint4*)(x + (long)m * ldx + n0), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += f[e];
    }
  }
  // fold the four row lanes when they ended in the same table row (always, unless a row-group boundary fell inside the
  // last four rows of this block); otherwise every lane flushes its own tail
  if (tx == 0) red_idx[ty] = cur;
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx][e] = acc[e];
  __syncthreads();
  if (part) {
    // deterministic mode: a block lies inside ONE table row (rows_per_block divides vdiv), so the four row lanes always fold
    // -- a lane that saw no row (blocks of fewer than four rows: M < 4, M % rows_per_block in {1, 2, 3}) holds zeros -- and
    // row lane 0 is the single writer of the block's slot (plain stores from several lanes would overwrite each other)
    if (ty == 0 && col_ok) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        part[(long)blockIdx.y * N + n0 + e] = red[0][tx][e] + red[1][tx][e] + red[2][tx][e] + red[3][tx][e];
    }
    return;
  }
  const bool same = red_idx[0] == red_idx[1] && red_idx[1] == red_idx[2] && red_idx[2] == red_idx[3];
  if (same) {
    if (ty == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = red[0][tx][e] + red[1][tx][e] + red[2][tx][e] + red[3][tx][e];
      flush(cur);
    }
  } else {
    flush(cur);
  }
}
// out[idx(b)][n] += scale * sum over the blocks b of table row idx, in a FIXED order (single writer per element): 32 columns x
// 8 block lanes per workgroup -- lane l adds blocks l, l + 8, .. in order, the eight lane sums are added in lane order.
// (The first version walked all blocks in one thread per column, a 64-bit division per block: 125 us per call at M = 230 400,
// 6 ms of the cfg5 step for 48 calls.)
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int nblocks, int rows_per_block, int N,
                                                            int vmode, int vdiv, int vmod, float scale, float* __restrict__ out,
                                                            int ldo) {
  __shared__ float red[8][32];
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + lx, idx = blockIdx.y;
  float t = 0.f;
  if (n < N) {
    for (int b = ly; b < nblocks; b += 8) {
      const int bi = vmode ? (int)(((unsigned)b * (unsigned)rows_per_block / (unsigned)vdiv) % (unsigned)vmod) : 0;
      if (bi == idx) t += part[(long)b * N + n];
    }
  }
  red[ly][lx] = t;
  __syncthreads();
  if (ly == 0 && n < N) {
    float v = red[0][lx];
#pragma unroll
    for (int l = 1; l < 8; ++l) v += red[l][lx];
    out[(long)idx * ldo + n] += v * scale;
  }
}

// out[0] += scale * sum_i dy[i] * (p[i] - q[i])      (gradient of a folded AlphaBlender's mixing weight)
__global__ __launch_bounds__(256) void dot_diff_kernel(const el_t* __restrict__ dy, const el_t* __restrict__ p,
                                                       const el_t* __restrict__ q, size_t n, float scale,
                                                       float* __restrict__ out, float* __restrict__ part) {
  __shared__ float red[4];
  float acc = 0.f;
  const size_t nv = n >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    float a[8], b[8], c[8];
    unpack_elx8(((const uint4*)dy)[i], a);
    unpack_elx8(((const uint4*)p)[i], b);
    unpack_elx8(((const uint4*)q)[i], c);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += a[e] * (b[e] - c[e]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const size_t i = (nv << 3) + threadIdx.x;
    acc += el_to_f32(dy[i]) * (el_to_f32(p[i]) - el_to_f32(q[i]));
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (part) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];       // deterministic: summed in block order below
    else atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale);
  }
}
__global__ __launch_bounds__(64) void dot_diff_reduce_kernel(const float* __restrict__ part, int nblocks, float scale,
                                                             float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double t = 0.0;
  for (int b = 0; b < nblocks; ++b) t += (double)part[b];
  out[0] += (float)t * scale;
}

// ------------------------------------------------------------------------------------------------ GroupNorm backward
struct GnB { int n_img, S, C, ips, rows_per_chunk, n_chunks; };

__device__ __forceinline__ float dsilu(float z) {            // d/dz [z * sigmoid(z)]
  const float sg = 1.0f / (1.0f + __expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}

// pass 1: per (image, chunk, channel): s1 = sum dz * xhat, s2 = sum dz      grid (n_chunks, n_img), C/8 * RPP threads
__global__ void gn_bwd_partial_kernel(const el_t* __restrict__ x, const el_t* __restrict__ dy, GnB s,
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
    unpack_elx8(*(const uint4*)(x + row * s.C + c0), fx);
    unpack_elx8(*(const uint4*)(dy + row * s.C + c0), fd);
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

// pass 2a: per (statistics row, group): m1 = mean(dz * gamma), m2 = mean(dz * gamma * xhat)    grid (32, n_stat), 256 threads
// (one workgroup per (row, GROUP): the 5-D norms of a one-clip batch have a single statistics row, and one workgroup
// walking all of its partials serially took up to 0.45 ms)
__global__ __launch_bounds__(256) void gn_bwd_group_kernel(GnB s, const float* __restrict__ part,
                                                           const float* __restrict__ gamma, float* __restrict__ gmean) {
  __shared__ double dred[256][2];
  const int tid = threadIdx.x, g = blockIdx.x, stat = blockIdx.y, cpg = s.C / 32;
  const int tot = s.ips * s.n_chunks;              // (image, chunk) pairs of this statistics row, contiguous in `part`
  double a1 = 0.0, a2 = 0.0;
  for (int i = tid; i < tot * cpg; i += 256) {
    const int k = i / cpg, c = g * cpg + i % cpg;
    const float2 p = *(const float2*)(part + (((long)stat * tot + k) * s.C + c) * 2);
    a1 += (double)gamma[c] * (double)p.y;
    a2 += (double)gamma[c] * (double)p.x;
  }
  dred[tid][0] = a1;
  dred[tid][1] = a2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { dred[tid][0] += dred[tid + o][0]; dred[tid][1] += dred[tid + o][1]; }
    __syncthreads();
  }
  if (tid == 0) {
    const double cnt = (double)cpg * s.S * s.ips;
    gmean[((long)stat * 32 + g) * 2] = (float)(dred[0][0] / cnt);
    gmean[((long)stat * 32 + g) * 2 + 1] = (float)(dred[0][1] / cnt);
  }
}
// pass 2b: dgamma[c] += sum s1, dbeta[c] += sum s2 over every (image, chunk): 64 channels x 4 row lanes per workgroup,
// gridDim.y row groups, one atomic per channel per workgroup            grid (ceil(C/64), <= 16)
// (round 5: no atomics -- every row group writes its own [2][C] slice of `lvl2`, gn_bwd_affine_fold_kernel adds the slices
//  in order: the parameter gradients are bit-reproducible)
__global__ __launch_bounds__(256) void gn_bwd_affine_kernel(GnB s, const float* __restrict__ part, float* __restrict__ lvl2) {
  __shared__ float red[4][64][2];
  const int tx = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float a1 = 0.f, a2 = 0.f;
  const long tot = (long)s.n_img * s.n_chunks;
  if (c < s.C)
    for (long k = blockIdx.y * 4 + rl; k < tot; k += gridDim.y * 4) {
      const float2 p = *(const float2*)(part + (k * s.C + c) * 2);
      a1 += p.x;
      a2 += p.y;
    }
  red[rl][tx][0] = a1;
  red[rl][tx][1] = a2;
  __syncthreads();
  if (rl == 0 && c < s.C) {
    lvl2[((long)blockIdx.y * 2) * s.C + c] = red[0][tx][0] + red[1][tx][0] + red[2][tx][0] + red[3][tx][0];
    lvl2[((long)blockIdx.y * 2 + 1) * s.C + c] = red[0][tx][1] + red[1][tx][1] + red[2][tx][1] + red[3][tx][1];
  }
}
// out0[c] += sum_g lvl2[g][0][c], out1[c] += sum_g lvl2[g][1][c]   (groups in order; one thread per column: single writer)
__global__ __launch_bounds__(256) void affine_fold_kernel(const float* __restrict__ lvl2, int groups, int C, float* __restrict__ out0,
                                                          float* __restrict__ out1) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int g = 0; g < groups; ++g) { a += lvl2[((long)g * 2) * C + c]; b += lvl2[((long)g * 2 + 1) * C + c]; }
  out0[c] += a;
  out1[c] += b;
}
// pass 3: dx = rstd * (dz * gamma - m1 - xhat * m2)  [+ add: the gradient arriving at x through its OTHER consumer -- the
// skip connection around the branch this norm opens -- summed here instead of in a separate pass over both tensors]
__global__ void gn_bwd_apply_kernel(const el_t* __restrict__ x, const el_t* __restrict__ dy, GnB s,
                                    const float* __restrict__ stats, const float* __restrict__ gmean,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                    const el_t* __restrict__ add, el_t* __restrict__ dx) {
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
    unpack_elx8(*(const uint4*)(x + row * s.C + c0), fx);
    unpack_elx8(*(const uint4*)(dy + row * s.C + c0), fd);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (fx[e] - mean[e]) * rstd[e];
      const float dz = silu ? fd[e] * dsilu(xh * g[e] + b[e]) : fd[e];
      fx[e] = rstd[e] * (dz * g[e] - m1[e] - xh * m2[e]);
    }
    if (add) {
      float fa[8];
      unpack_elx8(*(const uint4*)(add + row * s.C + c0), fa);
#pragma unroll
      for (int e = 0; e < 8; ++e) fx[e] += fa[e];
    }
    *(uint4*)(dx + row * s.C + c0) = pack_elx8(fx);
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
// One wave per row (grid-stride), 8 * NV columns per lane like the forward.  Recomputes mean / rstd of x (+ V row), then
//   dz = dy * gamma;  dx = rstd * (dz - mean(dz) - xhat * mean(dz * xhat));  dgamma += dy * xhat;  dbeta += dy
// dgamma / dbeta: per-lane column partials over the wave's rows go to a scratch row per wave ([waves][2C] fp32) and
// ln_bwd_reduce_kernel folds the rows with 16 atomics per column.  (One atomic per column PER WAVE -- 4096 same-address
// memory-side atomics per column -- made this kernel 10x slower than its HBM traffic: 0.93 ms at C = 320, M = 230 k.)
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const el_t* __restrict__ x, const el_t* __restrict__ dy, int M, int C,
                                                     const float* __restrict__ gamma, float eps, const float* __restrict__ V,
                                                     int vdiv, int vmod, int ldv, const el_t* __restrict__ add,
                                                     el_t* __restrict__ dx, float* __restrict__ part) {
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
        unpack_elx8(*(const uint4*)(x + m * C + cv * 8), fx[k]);
        unpack_elx8(*(const uint4*)(dy + m * C + cv * 8), fd[k]);
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
        if (add) {                                           // (the skip connection's gradient: see gn_bwd_apply_kernel)
          float fa[8];
          unpack_elx8(*(const uint4*)(add + m * C + cv * 8), fa);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += fa[e];
        }
        *(uint4*)(dx + m * C + cv * 8) = pack_elx8(o);
      }
    }
  }
  float* prow = part + ((long)blockIdx.x * 4 + wid) * (2L * C);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int cv = lane + k * 64;
    if (cv < CV) {
      *(float4*)(prow + cv * 8) = make_float4(ag[k][0], ag[k][1], ag[k][2], ag[k][3]);
      *(float4*)(prow + cv * 8 + 4) = make_float4(ag[k][4], ag[k][5], ag[k][6], ag[k][7]);
      *(float4*)(prow + C + cv * 8) = make_float4(ab[k][0], ab[k][1], ab[k][2], ab[k][3]);
      *(float4*)(prow + C + cv * 8 + 4) = make_float4(ab[k][4], ab[k][5], ab[k][6], ab[k][7]);
    }
  }
}

// out0[c] += sum_r part[r][c], out1[c] += sum_r part[r][C + c]: 64 columns x 4 row lanes per workgroup, gridDim.y row groups
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int rows, int C,
                                                            float* __restrict__ lvl2) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float s = 0.f;
  if (col < 2 * C)
    for (int r = blockIdx.y * 4 + rl; r < rows; r += gridDim.y * 4) s += part[(long)r * (2L * C) + col];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < 2 * C) {
    s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    lvl2[(long)blockIdx.y * 2 * C + col] = s;             // (row group g: [dgamma C | dbeta C]; folded in order, no atomics)
  }
}

// ------------------------------------------------------------------------------------------------ GEGLU backward
// raw: the projection output [M][2I] in the packed (16-value, 16-gate) column-block order (the forward GEMM without its
// GEGLU epilogue); du: gradient of u = a * gelu(g), [M][I].  draw (same layout as raw): da = du * gelu(g),
// dg = du * a * (Phi(g) + g * phi(g)).
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const el_t* __restrict__ raw, const el_t* __restrict__ du, long M,
                                                        int I, el_t* __restrict__ draw) {
  const long total = M * (I >> 3);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (I >> 3);
    const int j0 = (int)(i % (I >> 3)) * 8;            // 8 consecutive u columns: inside one 16-column block
    const int blk = j0 >> 4, r0 = j0 & 15;
    const long base = m * (2L * I) + blk * 32 + r0;
    float a[8], g[8], d[8], oa[8], og[8];
    unpack_elx8(*(const uint4*)(raw + base), a);
    unpack_elx8(*(const uint4*)(raw + base + 16), g);
    unpack_elx8(*(const uint4*)(du + m * I + j0), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const f32x4_t gv = {g[e], g[e], g[e], g[e]};
      const float Phi = gelu_phi4(gv).x;
      const float phi = 0.3989422804014327f * __expf(-0.5f * g[e] * g[e]);
      oa[e] = d[e] * (g[e] * Phi);
      og[e] = d[e] * a[e] * (Phi + g[e] * phi);
    }
    *(uint4*)(draw + base) = pack_elx8(oa);
    *(uint4*)(draw + base + 16) = pack_elx8(og);
  }
}

}  // namespace

static long ln_bwd_blocks(int M) {
  long blocks = ((long)M + 3) / 4;
  return blocks > 256 * 4 ? 256 * 4 : blocks;
}
constexpr int kFoldGroups = 16;       // row groups of the parameter-gradient folds (level-2 slices behind the level-1 partials)
extern "C" size_t ctrlv_layernorm_bwd_scratch_floats(int M, int C) {
  return (size_t)ln_bwd_blocks(M) * 4 * 2 * (size_t)C + (size_t)kFoldGroups * 2 * (size_t)C;
}

extern "C" int ctrlv_layernorm_bwd(const void* x, const void* dy, int M, int C, const float* gamma, float eps, const float* V,
                                   int vdiv, int vmod, int ldv, void* dx, float* dgamma, float* dbeta, float* scratch,
                                   ctrlv_stream_t stream) {
  return ctrlv_layernorm_bwd_add(x, dy, nullptr, M, C, gamma, eps, V, vdiv, vmod, ldv, dx, dgamma, dbeta, scratch, stream);
}
extern "C" int ctrlv_layernorm_bwd_add(const void* x, const void* dy, const void* add, int M, int C, const float* gamma, float eps,
                                       const float* V, int vdiv, int vmod, int ldv, void* dx, float* dgamma, float* dbeta,
                                       float* scratch, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && dy && gamma && dx && dgamma && dbeta && scratch, "layernorm_bwd: null pointer");
  CTRLV_CHECK_SHAPE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "layernorm_bwd: C=%d must be a multiple of 8, <= 2048", C);
  if (V) CTRLV_CHECK_ARG(vdiv > 0 && vmod > 0 && ldv >= C, "layernorm_bwd: bad row-vector table");
  const int nv = (C / 8 + 63) / 64;
  const long blocks = ln_bwd_blocks(M);
  hipStream_t st = (hipStream_t)stream;
#define LNB_LAUNCH(NV)                                                                                                  \
  hipLaunchKernelGGL(ln_bwd_kernel<NV>, dim3((unsigned)blocks), dim3(256), 0, st, (const el_t*)x, (const el_t*)dy, M, C, \
                     gamma, eps, V, vdiv, vmod, ldv, (const el_t*)add, (el_t*)dx, scratch)
  switch (nv) {
    case 1: LNB_LAUNCH(1); break;
    case 2: LNB_LAUNCH(2); break;
    case 3: LNB_LAUNCH(3); break;
    default: LNB_LAUNCH(4); break;
  }
#undef LNB_LAUNCH
  CTRLV_LAUNCH_CHECK();
  const int rows = (int)blocks * 4, groups = rows >= 64 ? kFoldGroups : 1;
  float* lvl2 = scratch + (size_t)rows * 2 * C;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * C + 63) / 64, groups), dim3(256), 0, st, scratch, rows, C, lvl2);
  CTRLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(affine_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, st, lvl2, groups, C, dgamma, dbeta);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_geglu_bwd(const void* raw, const void* du, size_t M, int I, void* draw, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(raw && du && draw, "geglu_bwd: null pointer");
  CTRLV_CHECK_SHAPE(M > 0 && I > 0 && I % 16 == 0, "geglu_bwd: inner dim %d must be a multiple of 16", I);
  size_t blocks = (M * (I / 8) + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const el_t*)raw,
                     (const el_t*)du, (long)M, I, (el_t*)draw);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

// M slabs of a wgrad launch: enough to fill the chip (two workgroups per CU), each a multiple of 64 rows
static int wgrad_slabs(const ctrlv_gemm_desc& d, int* rows_per_slab) {
  const int ktiles = (d.taps * d.Cin + 255) / 256, ntiles = (d.N + 127) / 128;
  int slabs = 512 / (ktiles * ntiles);
  if (slabs < 1) slabs = 1;
  int rps = ((d.M + slabs - 1) / slabs + 63) / 64 * 64;
  if (rps < 64) rps = 64;
  *rows_per_slab = rps;
  return (d.M + rps - 1) / rps;
}

// (the kernel choice is a function of the layer's shape: the pointer-alignment part of ctrlv_wgrad_pp_serves is checked at the
//  launch, against buffers the size query has not seen -- the larger of the two decompositions is reserved)
extern "C" size_t ctrlv_gemm_wgrad_scratch_bytes(const ctrlv_gemm_desc* dp) {
  if (!dp || dp->M <= 0 || dp->N <= 0 || dp->Cin <= 0 || dp->taps <= 0) return 0;
  int rps;
  int slabs = wgrad_slabs(*dp, &rps);
  ctrlv_gemm_desc probe = *dp;
  if (!probe.A) probe.A = (const void*)16;
  size_t bias_rows = (size_t)slabs;
  if (ctrlv_wgrad_pp_serves(probe, (const void*)16, 8)) {
    ctrlv_wgrad_pp_plan_t p;
    ctrlv_wgrad_pp_plan(*dp, &p);
    if (p.slabs > slabs) slabs = p.slabs;
    if ((size_t)p.slabs * p.ktiles > bias_rows) bias_rows = (size_t)p.slabs * p.ktiles;     // (bias partials per (slab, k tile))
  }
  return ((size_t)slabs * (size_t)dp->N * dp->taps * dp->Cin + bias_rows * dp->N) * sizeof(float);
}

extern "C" int ctrlv_gemm_wgrad(const ctrlv_gemm_desc* dp, const void* dY, int ldy, float* dW, float* dbias, float scale,
                                int torch_layout, void* scratch, size_t scratch_bytes, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(dp && dY && dW, "gemm_wgrad: null pointer");
  CTRLV_CHECK_ARG(torch_layout >= 0 && torch_layout <= 3, "gemm_wgrad: torch_layout=%d (bit 0: parameter layout, bit 1: assign)", torch_layout);
  const int assign = (torch_layout >> 1) & 1;
  torch_layout &= 1;
  CTRLV_CHECK_ARG(!assign || scratch, "gemm_wgrad: assign (torch_layout bit 1) needs the deterministic form (scratch)");
  const ctrlv_gemm_desc& d = *dp;
  CTRLV_CHECK_ARG(d.A != nullptr, "gemm_wgrad: A must be non-null");
  CTRLV_CHECK_SHAPE(d.M > 0 && d.N > 0 && d.Cin > 0 && d.Cin % 64 == 0, "gemm_wgrad: Cin=%d must be a positive multiple of 64", d.Cin);
  CTRLV_CHECK_SHAPE(d.N % 8 == 0 && ldy % 8 == 0 && d.lda % 8 == 0, "gemm_wgrad: N, ldy, lda must be multiples of 8");
  CTRLV_CHECK_SHAPE((d.mode == 0 && d.taps == 1) || (d.mode == 1 && d.taps == 9) || (d.mode == 2 && d.taps == 3),
                    "gemm_wgrad: mode / taps mismatch");
  if (d.A2) CTRLV_CHECK_SHAPE(d.c_split % 64 == 0 && d.lda2 % 8 == 0, "gemm_wgrad: bad concat split");
  WgradArgs a;
  a.A = (const el_t*)d.A; a.A2 = (const el_t*)d.A2; a.dY = (const el_t*)dY; a.dW = dW; a.dbias = dbias; a.scale = scale; a.torch_layout = torch_layout;
  a.M = d.M; a.N = d.N; a.Cin = d.Cin; a.taps = d.taps; a.lda = d.lda; a.lda2 = d.lda2; a.c_split = d.c_split; a.ldy = ldy;
  a.mode = d.mode; a.H = d.H; a.Wd = d.Wd; a.Ho = d.Ho; a.Wo = d.Wo; a.stride = d.stride ? d.stride : 1; a.up = d.up;
  a.F = d.F; a.S = d.S;
  const int ktiles = (d.taps * d.Cin + 255) / 256, ntiles = (d.N + 127) / 128;
  int rps;
  int slabs = wgrad_slabs(d, &rps);
  a.rows_per_slab = rps;
  CTRLV_CHECK_SHAPE((long)ntiles * ktiles * slabs < (1L << 30), "gemm_wgrad: grid too large");
  a.ntiles = ntiles; a.ktiles = ktiles; a.slabs = slabs;
  // scratch given: DETERMINISTIC -- slab partials with plain stores, then an ordered sum (wgrad_reduce_kernel); without it
  // the slabs add into dW with fp32 atomics (order varies from run to run)
  a.part = nullptr;
  if (scratch) {
    CTRLV_CHECK_ARG(scratch_bytes >= ctrlv_gemm_wgrad_scratch_bytes(dp), "gemm_wgrad: scratch smaller than ctrlv_gemm_wgrad_scratch_bytes()");
    CTRLV_CHECK_SHAPE((d.taps * d.Cin) % 4 == 0, "gemm_wgrad: K must be a multiple of 4");
    a.part = (float*)scratch;
  }
  int bias_rows = slabs;
  if (ctrlv_wgrad_pp_serves(d, dY, ldy)) {        // the LDS-DMA kernel (wgrad_pp.hip): same partial layout, its own slab count
    ctrlv_wgrad_pp_plan_t p;
    ctrlv_wgrad_pp_plan(d, &p);
    slabs = p.slabs;
    bias_rows = p.slabs * p.ktiles;
    const int rc = ctrlv_wgrad_pp_launch(d, dY, ldy, dW, dbias, scale, torch_layout, a.part, p, stream);
    if (rc != CTRLV_OK) return rc;
  } else {
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)(ntiles * ktiles * slabs)), dim3(256), 0, (hipStream_t)stream, a);
    CTRLV_LAUNCH_CHECK();
  }
  if (a.part) {
    const long ktot = (long)d.taps * d.Cin, n_thr = (long)d.N * (ktot >> 2);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a.part, slabs,
                       d.N, ktot, d.Cin, d.taps, torch_layout, scale, dW, dbias, assign, bias_rows);
    CTRLV_LAUNCH_CHECK();
  }
  return CTRLV_OK;
}

// rows per block of a colsum launch; deterministic mode: a divisor of vdiv, so that a block never straddles two table rows
static int colsum_rpb(int M, int vmode, int vdiv, bool det) {
  int rpb = M >= (1 << 18) ? 1024 : (M >= (1 << 14) ? 256 : 64);           // >= ~225 workgroups per 512 columns at L0
  if (det && vmode) {
    while (rpb > 4 && vdiv % rpb != 0) rpb >>= 1;
    if (vdiv % rpb != 0) rpb = 0;                                          // (odd row groups: not served deterministically)
  }
  return rpb;
}
extern "C" size_t ctrlv_colsum_scratch_floats(int M, int N, int vmode, int vdiv) {
  const int rpb = colsum_rpb(M, vmode, vdiv, true);
  return rpb > 0 ? (size_t)((M + rpb - 1) / rpb) * (size_t)N : 0;
}

extern "C" int ctrlv_colsum(const void* x, int M, int N, int ldx, int vmode, int vdiv, int vmod, float scale, float* out,
                            int ldo, float* scratch, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && out, "colsum: null pointer");
  CTRLV_CHECK_SHAPE(M > 0 && N > 0 && N % 8 == 0 && ldx >= N && ldx % 8 == 0, "colsum: N and ldx must be multiples of 8");
  CTRLV_CHECK_ARG(vmode == 0 || (vmode == 1 && vdiv > 0 && vmod > 0), "colsum: vmode must be 0 or 1 with vdiv, vmod > 0");
  // scratch (ctrlv_colsum_scratch_floats) given: DETERMINISTIC -- per-block sums, then an ordered add (no atomics)
  const int rpb = colsum_rpb(M, vmode, vdiv, scratch != nullptr);
  CTRLV_CHECK_SHAPE(rpb > 0, "colsum: deterministic mode needs a row-group size vdiv=%d divisible by a power of two >= 4", vdiv);
  const int nblocks = (M + rpb - 1) / rpb;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 511) / 512, nblocks), dim3(256), 0, (hipStream_t)stream,
                     (const el_t*)x, M, N, ldx, rpb, vmode, vdiv, vmod, scale, out, ldo, scratch);
  CTRLV_LAUNCH_CHECK();
  if (scratch) {
    const int n_idx = vmode ? (int)(((long)(M - 1) / vdiv + 1) < vmod ? ((long)(M - 1) / vdiv + 1) : vmod) : 1;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 31) / 32, n_idx), dim3(256), 0, (hipStream_t)stream, scratch, nblocks, rpb, N,
                       vmode, vdiv, vmod, scale, out, ldo);
    CTRLV_LAUNCH_CHECK();
  }
  return CTRLV_OK;
}

extern "C" int ctrlv_dot_diff(const void* dy, const void* p, const void* q, size_t n, float scale, float* out,
                              float* scratch, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(dy && p && q && out, "dot_diff: null pointer");
  size_t blocks = (n / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  // scratch (1024 floats) given: DETERMINISTIC -- per-block sums, added in block order by one thread
  hipLaunchKernelGGL(dot_diff_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const el_t*)dy,
                     (const el_t*)p, (const el_t*)q, n, scale, out, scratch);
  CTRLV_LAUNCH_CHECK();
  if (scratch) {
    hipLaunchKernelGGL(dot_diff_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, (int)blocks, scale, out);
    CTRLV_LAUNCH_CHECK();
  }
  return CTRLV_OK;
}

extern "C" int ctrlv_groupnorm_bwd_scratch_floats(int n_img, int S, int C, int imgs_per_stat) {
  const int chunks = ctrlv_groupnorm_chunks(n_img, S, C, imgs_per_stat);
  if (chunks < 0) return chunks;
  const long need = (long)n_img * chunks * C * 2 + (long)(n_img / imgs_per_stat) * 64 + (long)kFoldGroups * 2 * C;
  CTRLV_CHECK_SHAPE(need < (1L << 31), "groupnorm_bwd: scratch too large");
  return (int)need;
}

extern "C" int ctrlv_groupnorm_bwd(const void* x, const void* dy, int n_img, int S, int C, int imgs_per_stat,
                                   const float* fwd_partials, const float* gamma, const float* beta, int silu, void* dx,
                                   float* dgamma, float* dbeta, float* scratch, ctrlv_stream_t stream) {
  return ctrlv_groupnorm_bwd_add(x, dy, nullptr, n_img, S, C, imgs_per_stat, fwd_partials, gamma, beta, silu, dx, dgamma, dbeta,
                                 scratch, stream);
}
extern "C" int ctrlv_groupnorm_bwd_add(const void* x, const void* dy, const void* add, int n_img, int S, int C, int imgs_per_stat,
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
                     (const el_t*)x, (const el_t*)dy, s, stats, gamma, beta, silu, part);
  CTRLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_group_kernel, dim3(32, n_img / imgs_per_stat), dim3(256), 0, st, s, part, gamma, gmean);
  CTRLV_LAUNCH_CHECK();
  {
    const long tot = (long)n_img * s.n_chunks;
    const int groups = tot >= 64 ? kFoldGroups : 1;
    float* lvl2 = gmean + (size_t)(n_img / imgs_per_stat) * 64;
    hipLaunchKernelGGL(gn_bwd_affine_kernel, dim3((C + 63) / 64, groups), dim3(256), 0, st, s, part, lvl2);
    CTRLV_LAUNCH_CHECK();
    hipLaunchKernelGGL(affine_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, st, lvl2, groups, C, dgamma, dbeta);
  }
  CTRLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(chunks, n_img), dim3(nt), 0, st, (const el_t*)x, (const el_t*)dy, s, stats,
                     gmean, gamma, beta, silu, (const el_t*)add, (el_t*)dx);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
