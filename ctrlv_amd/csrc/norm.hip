// GroupNorm(32)(+SiLU) and LayerNorm on channels-last bf16 rows, gfx950.  HBM-bound kernels: 16-byte vector
// loads/stores (8 bf16 per lane), fp32 math, wave64 shuffle reductions, fp64 final combine of the split statistics.
//
// GroupNorm is a split reduction (SURVEY hard part H3: the 5-D temporal norm has only B*32 statistics rows of
// up to 2.3 M elements each): pass 1 writes per-(image, row-chunk, group) partial (mean_c, M2_c = sum (x-mean_c)^2),
// accumulated relative to per-channel pilot values so that mean >> std inputs do not cancel in fp32; pass 1b combines
// the partials of a statistics row (deterministic order, fp64, Chan's formula) into (mean, rstd); pass 2 streams
// x -> y = silu(x*a_c + b_c).
// Algorithmic traffic: 2 reads + 1 write of the tensor (the second read of mid-size tensors is served by L2 /
// Infinity Cache).
#include <stdlib.h>

#include "common.h"

namespace {

struct GnShape {
  int n_img, S, C, imgs_per_stat, c_split, n_chunks, rows_per_chunk, CV, RPP;
  int rev_stats, rev_apply;       // walk the images / chunks from the END (which workgroup reads what first; results unchanged)
};

__device__ __forceinline__ uint4 gn_load(const el_t* x, const el_t* x2, int c_split, int C, long row, int c0) {
  if (x2 != nullptr && c0 >= c_split) return *(const uint4*)(x2 + row * (C - c_split) + (c0 - c_split));
  const int ld = x2 != nullptr ? c_split : C;
  return *(const uint4*)(x + row * ld + c0);
}
// SPLIT inputs (ctrlv_gemm_desc.out_lo): the lo planes of x / x2, same shapes and pitches; a null plane reads as zeros.
struct GnLo { const lo_t* x; const lo_t* x2; };     // (one byte per element: common.h lo_t)
// the lo-plane pointer that belongs to column block c0 (fixed per thread), or null; `first` = that half's first column
__device__ __forceinline__ const lo_t* gn_lo_plane(const GnLo& lo, const el_t* x2, int c_split, int C, int c0, int* ld, int* first) {
  if (x2 != nullptr && c0 >= c_split) { *ld = C - c_split; *first = c_split; return lo.x2; }
  *ld = x2 != nullptr ? c_split : C; *first = 0;
  return lo.x;
}

// Pass 1.  Per (image, row-chunk): every lane accumulates, for its 8 channels, sum (x - K_c) and sum (x - K_c)^2 about a
// per-CHANNEL pilot K_c = x[first row of the chunk][c] (one extra 16-byte load per lane, the same row for every lane of
// a column, so all partial sums of a channel share their pilot): with |mean| >> std, E[x^2] - mean^2 would cancel in
// fp32, the shifted sums do not.  32 threads then turn the channel sums of their group into the chunk's
// (mean_c, M2_c) with Chan's pairwise update (equal counts per channel).
template <bool SPLIT>
__global__ void gn_stats_kernel(const el_t* __restrict__ x, const el_t* __restrict__ x2, GnLo lo, GnShape s,
                                float* __restrict__ partials) {
  extern __shared__ float red[];  // [RPP][C][2] sums, then [C] pilots
  const int tid = threadIdx.x;
  const int col = tid % s.CV, rsub = tid / s.CV;
  const int n = s.rev_stats ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int chunk = s.rev_stats ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int r0 = chunk * s.rows_per_chunk;
  const int r1 = min(s.S, r0 + s.rows_per_chunk);
  const int c0 = col * 8;
  int lo_ld = 0, lo_first = 0;
  const lo_t* const lop = SPLIT ? gn_lo_plane(lo, x2, s.c_split, s.C, c0, &lo_ld, &lo_first) : nullptr;
  // (float2 arithmetic: v_pk_add_f32 / v_pk_fma_f32, two channels per instruction)
  f32x2_t sm2[4], sq2[4], piv2[4];
  {
    const uint4 pv = gn_load(x, x2, s.c_split, s.C, (long)n * s.S + r0, c0);
    float pf[8];
    unpack_elx8(pv, pf);
    // (the pilot is the hi value of the first row: any value near the channel's data serves)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      piv2[e] = f32x2_t{pf[2 * e], pf[2 * e + 1]};
      sm2[e] = f32x2_t{0.f, 0.f};
      sq2[e] = f32x2_t{0.f, 0.f};
    }
  }
  auto accum = [&](const uint4& v, long row) {
    float f[8];
    unpack_elx8(v, f);
    if (SPLIT && lop) add_lo8(f, *(const uint2*)(lop + row * lo_ld + (c0 - lo_first)));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x2_t dl = f32x2_t{f[2 * e], f[2 * e + 1]} - piv2[e];
      sm2[e] += dl;
      sq2[e] += dl * dl;
    }
  };
  // four rows per trip: four independent 16-byte loads in flight per lane (HBM latency ~1 us; one load per trip left
  // the kernel latency-bound at 3.6 TB/s)
  int r = r0 + rsub;
  for (; r + 3 * s.RPP < r1; r += 4 * s.RPP) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = gn_load(x, x2, s.c_split, s.C, (long)n * s.S + r + u * s.RPP, c0);
#pragma unroll
    for (int u = 0; u < 4; ++u) accum(v[u], (long)n * s.S + r + u * s.RPP);
  }
  for (; r < r1; r += s.RPP) accum(gn_load(x, x2, s.c_split, s.C, (long)n * s.S + r, c0), (long)n * s.S + r);
  float* pil = red + (size_t)s.RPP * s.C * 2;     // per channel: [mean | M2] after stage A
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    *(float4*)(red + ((rsub * s.C) + c0 + 2 * e) * 2) = make_float4(sm2[e].x, sq2[e].x, sm2[e].y, sq2[e].y);
    if (rsub == 0) { pil[c0 + 2 * e] = piv2[e].x; pil[c0 + 2 * e + 1] = piv2[e].y; }
  }
  __syncthreads();
  // stage A (all threads, one channel each): sums over the row sub-sets -> channel (mean, M2) in place of the pilot
  const float nrow = (float)(r1 - r0), inv_nrow = 1.0f / nrow;
  for (int c = tid; c < s.C; c += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int rs = 0; rs < s.RPP; ++rs) {
      const float2 v = *(const float2*)(red + (rs * s.C + c) * 2);
      a += v.x;
      b += v.y;
    }
    const float mc = pil[c] + a * inv_nrow;          // channel mean over the chunk
    red[c * 2] = mc;                                 // (row sub-set 0 of this channel was consumed by this thread)
    red[c * 2 + 1] = b - a * a * inv_nrow;           // channel M2 about its own mean (no cancellation: a is small)
  }
  __syncthreads();
  if (tid < 32) {
    const int cpg = s.C / 32;
    float mean = 0.f, m2 = 0.f;          // running (mean, M2) over the channels seen so far, nrow elements each
    for (int k = 0; k < cpg; ++k) {
      const float2 v = *(const float2*)(red + (tid * cpg + k) * 2);
      const float dlt = v.x - mean;
      const float rk = 1.0f / (float)(k + 1);
      m2 += v.y + dlt * dlt * (nrow * (float)k * rk);
      mean += dlt * rk;
    }
    const int stat = n / s.imgs_per_stat;
    const int gchunk = (n % s.imgs_per_stat) * s.n_chunks + chunk;
    const long o = (((long)stat * s.imgs_per_stat * s.n_chunks + gchunk) * 32 + tid) * 2;
    partials[o] = mean;                            // chunk mean of the group
    partials[o + 1] = m2;                          // chunk M2 of the group (about its own mean)
  }
}

// Pass 1b.  Per statistics row: combine its chunk partials (fixed order, fp64) into (mean, rstd) per group, written behind
// the partials.  Keeps the per-workgroup prologue of the apply pass at 64 floats instead of a walk over up to 25 x 36 chunks
// (230 KiB from L2 per workgroup at the L0 temporal norm).  One workgroup per (statistics row, 4 groups): 256 slices of
// the chunk list x 4 groups -- the 5-D norms have 900 chunks per row, 3600 when the producing GEMM wrote them (64-row
// chunks), and one workgroup per row walked them in 113 dependent-latency trips (~50 us; 14 trips now).
constexpr int kFinSlices = 256, kFinGroups = 4;
__global__ __launch_bounds__(1024) void gn_finalize_kernel(GnShape s, const float* __restrict__ partials, float eps,
                                                           float* __restrict__ stats) {
  __shared__ double dred[16][kFinGroups][2];
  const int tid = threadIdx.x, stat = blockIdx.x;
  const int tot_chunks = s.imgs_per_stat * s.n_chunks;
  const int g = blockIdx.y * kFinGroups + (tid & 3), sl = tid >> 2;
  double a = 0.0, b = 0.0;                        // a = sum n_c*mean_c,  b = sum (M2_c + n_c*mean_c^2), exact in fp64
  const float* p = partials + ((long)stat * tot_chunks) * 64;
  const int cpg = s.C / 32;
  const int last_rows = s.S - (s.n_chunks - 1) * s.rows_per_chunk;
  for (int k = sl; k < tot_chunks; k += kFinSlices) {
    const int ck = k % s.n_chunks;
    const int rows = ck == s.n_chunks - 1 ? last_rows : s.rows_per_chunk;
    const double nc = (double)(cpg * rows);
    const float2 pm = *(const float2*)(p + (k * 32 + g) * 2);
    const double mc = (double)pm.x;
    a += nc * mc;
    b += (double)pm.y + nc * mc * mc;
  }
  // slices of one group sit 4 lanes apart: xor-reduce over lane bits 2..5, then the 16 waves through LDS (fixed order)
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
  if ((tid & 63) < 4) { dred[tid >> 6][tid & 3][0] = a; dred[tid >> 6][tid & 3][1] = b; }
  __syncthreads();
  if (tid < kFinGroups) {
    a = 0.0; b = 0.0;
    for (int k = 0; k < 16; ++k) { a += dred[k][tid][0]; b += dred[k][tid][1]; }
    const double cnt = (double)cpg * (double)s.S * (double)s.imgs_per_stat;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((long)stat * 32 + g) * 2] = (float)mean;
    stats[((long)stat * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <bool SPLIT>
__global__ void gn_apply_kernel(const el_t* __restrict__ x, const el_t* __restrict__ x2, GnLo lo, GnShape s,
                                const float* __restrict__ stats, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int silu, el_t* __restrict__ y) {
  const int tid = threadIdx.x;
  const int n = s.rev_apply ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int chunk = s.rev_apply ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int stat = n / s.imgs_per_stat;
  const int col = tid % s.CV, rsub = tid / s.CV;
  const int c0 = col * 8, cpg = s.C / 32;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = (c0 + e) / cpg;
    const float mean = stats[((long)stat * 32 + g) * 2], rstd = stats[((long)stat * 32 + g) * 2 + 1];
    a[e] = rstd * gamma[c0 + e];
    b[e] = beta[c0 + e] - mean * a[e];
  }
  const int r0 = chunk * s.rows_per_chunk;
  const int r1 = min(s.S, r0 + s.rows_per_chunk);
  int lo_ld = 0, lo_first = 0;
  const lo_t* const lop = SPLIT ? gn_lo_plane(lo, x2, s.c_split, s.C, c0, &lo_ld, &lo_first) : nullptr;
  auto apply_row = [&](const uint4& v, long row) {
    float f[8];
    unpack_elx8(v, f);
    if (SPLIT && lop) add_lo8(f, *(const uint2*)(lop + row * lo_ld + (c0 - lo_first)));
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = f[e] * a[e] + b[e];
      f[e] = silu ? silu_f(t) : t;
    }
    *(uint4*)(y + row * s.C + c0) = pack_elx8(f);
  };
  int r = r0 + rsub;
  for (; r + 3 * s.RPP < r1; r += 4 * s.RPP) {      // four loads in flight per lane, as in the statistics pass
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = gn_load(x, x2, s.c_split, s.C, (long)n * s.S + r + u * s.RPP, c0);
#pragma unroll
    for (int u = 0; u < 4; ++u) apply_row(v[u], (long)n * s.S + r + u * s.RPP);
  }
  for (; r < r1; r += s.RPP) apply_row(gn_load(x, x2, s.c_split, s.C, (long)n * s.S + r, c0), (long)n * s.S + r);
}

int gn_shape(int n_img, int S, int C, int imgs_per_stat, int c_split, bool has_x2, GnShape* s) {
  CTRLV_CHECK_SHAPE(n_img > 0 && S > 0 && C > 0 && C % 32 == 0 && C % 8 == 0 && C <= 8192,
                    "groupnorm: C=%d must be a multiple of 32 (and of 8), <= 8192", C);
  CTRLV_CHECK_SHAPE(imgs_per_stat > 0 && n_img % imgs_per_stat == 0, "groupnorm: n_img %% imgs_per_stat != 0");
  if (has_x2) CTRLV_CHECK_SHAPE(c_split > 0 && c_split < C && c_split % 8 == 0, "groupnorm: bad c_split %d", c_split);
  s->n_img = n_img; s->S = S; s->C = C; s->imgs_per_stat = imgs_per_stat; s->c_split = c_split;
  s->CV = C / 8;
  s->RPP = s->CV >= 256 ? 1 : 256 / s->CV;
  {
    // The statistics pass walks the tensor from its END and the apply pass from its start: each pass begins with the part of
    // the tensor the pass (or producer) before it touched last, i.e. what the 256 MB Infinity Cache still holds (L0,
    // 295 MB: 195.6 -> 186.9 us per norm; 14.7 -> 14.4 ms per step; same directions for both passes: no gain).  Which
    // workgroup reads what first does not change any partial sum.  A/B handle: CTRLV_GN_REV bit 0 = statistics pass
    // from the end, bit 1 = apply pass from the end.
    const int rev = ctrlv_debug().gn_rev;
    s->rev_stats = rev & 1; s->rev_apply = (rev >> 1) & 1;
  }
  // The row chunking -- and with it the order of every partial sum -- depends on the image size only, never on how
  // many images are in the batch: a clip's statistics are bit-identical whether it is normalised alone or in a batch
  // (clip independence, tests/test_fullsize_gpu.py).  ~1000-2000 workgroups at the cfg3 batch of 50 images:
  // 256-row chunks at S = 9216, 64 at S = 2304, 32 below.
  {
    const int rpc_big = ctrlv_debug().gn_rows;       // rows per chunk at S >= 4096 (default 256)
    s->rows_per_chunk = S >= 4096 ? rpc_big : (S >= 1024 ? 64 : 32);
  }
  // very large images (the VAE decoder: up to 576 x 1024 pixels per frame): at most 256 chunks per image, so that the
  // finalize pass (one workgroup per statistics row) does not walk tens of thousands of partials
  if ((S + s->rows_per_chunk - 1) / s->rows_per_chunk > 256) s->rows_per_chunk = (((S + 255) / 256) + 255) / 256 * 256;
  if (s->rows_per_chunk > S) s->rows_per_chunk = S;
  s->n_chunks = (S + s->rows_per_chunk - 1) / s->rows_per_chunk;
  return CTRLV_OK;
}

// ------------------------------------------------------------------------------------------------ LayerNorm
template <int NV>
__global__ __launch_bounds__(256) void ln_kernel(const el_t* __restrict__ x, const lo_t* __restrict__ xlo, int M, int C,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 float eps, const float* __restrict__ V, int vdiv, int vmod, int ldv,
                                                 el_t* __restrict__ y) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int CV = C >> 3;
  float g[NV][8], b[NV][8];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int cv = lane + k * 64;
    if (cv < CV) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { g[k][e] = gamma[cv * 8 + e]; b[k][e] = beta[cv * 8 + e]; }
    }
  }
  const float inv_c = 1.0f / (float)C;
  // the row of the NEXT trip is loaded before this trip's two wave reductions (dependent shuffle chains): with one load
  // per trip in flight the kernel was latency-bound
  const long step = (long)gridDim.x * 4;
  long m = (long)blockIdx.x * 4 + wid;
  uint4 nxt[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int cv = lane + k * 64;
    nxt[k] = (m < M && cv < CV) ? *(const uint4*)(x + m * C + cv * 8) : make_uint4(0, 0, 0, 0);
  }
  for (; m < M; m += step) {
    float f[NV][8];
    float s = 0.f;
    const float* vrow = V ? V + (long)((m / vdiv) % vmod) * ldv : nullptr;
    uint4 cur[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int cv = lane + k * 64;
      cur[k] = nxt[k];
      nxt[k] = (m + step < M && cv < CV) ? *(const uint4*)(x + (m + step) * C + cv * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
        const uint4 v = cur[k];
        unpack_elx8(v, f[k]);
        if (xlo) add_lo8(f[k], *(const uint2*)(xlo + m * C + cv * 8));
        if (vrow) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[k][e] += vrow[cv * 8 + e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[k][e];
      }
    }
    const float mean = wave_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dlt = f[k][e] - mean; q += dlt * dlt; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f[k][e] - mean) * rstd * g[k][e] + b[k][e];
        *(uint4*)(y + m * C + cv * 8) = pack_elx8(o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ row softmax
// p[r, :] = softmax(s[r, :]) for fp32 score rows, bf16 probabilities (the single-head, head-dim-512 attention of the VAE's
// mid block: scores come from a GEMM, P feeds the P.V GEMM).  One workgroup per row, the row held in registers between the
// max / sum passes (cols <= 256 x 64); base-2 exponentials of (s - max) * log2(e).
constexpr int kSmxPer = 64;
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int cols, long lds, el_t* __restrict__ p,
                                                           long ldp) {
  __shared__ float red[8];
  const float* row = s + (long)blockIdx.x * lds;
  el_t* out = p + (long)blockIdx.x * ldp;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float v[kSmxPer];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < kSmxPer; k += 4) {
    const int c = (k / 4 * 256 + tid) * 4;
    float4 q = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (c < cols) q = *(const float4*)(row + c);
    v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
    mx = fmaxf(fmaxf(mx, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wid] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < kSmxPer; ++k) {
    v[k] = __builtin_amdgcn_exp2f((v[k] - mx) * 1.44269504088896340736f);      // exp2(-inf) = 0 for the padding
    sum += v[k];
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wid] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
#pragma unroll
  for (int k = 0; k < kSmxPer; k += 4) {
    const int c = (k / 4 * 256 + tid) * 4;
    if (c < cols)
      *(uint2*)(out + c) = make_uint2(pack_elx2(v[k] * inv, v[k + 1] * inv), pack_elx2(v[k + 2] * inv, v[k + 3] * inv));
  }
}

// ---- LayerNorm, several rows per wave (round 4).  ln_kernel above gives one row to a wave: at C = 320 only 40 of its 64
// lanes hold data and every row pays two 6-step wave reductions -- 76 wave-instructions per row, 4.67 TB/s where the
// device copies at 5.3 (tools/stream_bench.py).  Here LPR = 8 / 16 / 32 lanes share a row (C = 320 / 640 / 1280: NCH = 5
// chunks of 16 B per lane, chunk k of lane j = column block j + k * LPR, so every load instruction reads LPR * 16
// contiguous bytes per row) and a wave normalises 64 / LPR rows at once: all lanes busy, log2(LPR)-step reductions, ~32
// wave-instructions per row.  Same arithmetic as ln_kernel (two passes in registers: mean, then centred squares; the
// optional row vector added first), different summation tree.  gamma / beta live in LDS (C floats each).
template <int NCH, bool SPLIT>
__global__ __launch_bounds__(256) void ln_rows_kernel(const el_t* __restrict__ x, const lo_t* __restrict__ xlo, int M, int C, int lpr_log2,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, const float* __restrict__ V, int vdiv, int vmod, int ldv,
                                                      el_t* __restrict__ y) {
  extern __shared__ float gb[];                    // [C] gamma | [C] beta
  for (int i = threadIdx.x; i < C; i += 256) { gb[i] = gamma[i]; gb[C + i] = beta[i]; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int LPR = 1 << lpr_log2, RPW = 64 >> lpr_log2;   // lanes per row, rows per wave
  const int j = lane & (LPR - 1), rsub = lane >> lpr_log2;
  const float inv_c = 1.0f / (float)C;
  const long step = (long)gridDim.x * 4 * RPW;
  long m = ((long)blockIdx.x * 4 + wid) * RPW + rsub;
  uint4 nxt[NCH];
  uint2 nxl[SPLIT ? NCH : 1];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    nxt[k] = m < M ? *(const uint4*)(x + m * C + (j + k * LPR) * 8) : make_uint4(0, 0, 0, 0);
    if (SPLIT) nxl[k] = m < M ? *(const uint2*)(xlo + m * C + (j + k * LPR) * 8) : make_uint2(0, 0);
  }
  for (; m - rsub < M; m += step) {                // (wave-uniform trip count: the shuffles below need every lane)
    uint4 cur[NCH];
    uint2 cul[SPLIT ? NCH : 1];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      cur[k] = nxt[k];
      nxt[k] = (m + step < M) ? *(const uint4*)(x + (m + step) * C + (j + k * LPR) * 8) : make_uint4(0, 0, 0, 0);
      if (SPLIT) {
        cul[k] = nxl[k];
        nxl[k] = (m + step < M) ? *(const uint2*)(xlo + (m + step) * C + (j + k * LPR) * 8) : make_uint2(0, 0);
      }
    }
    const bool ok = m < M;
    const float* vrow = (V && ok) ? V + (long)((m / vdiv) % vmod) * ldv : nullptr;
    float f[NCH][8];
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      unpack_elx8(cur[k], f[k]);
      if (SPLIT) add_lo8(f[k], cul[k]);
      if (vrow) {
        const float4 a = *(const float4*)(vrow + (j + k * LPR) * 8), b = *(const float4*)(vrow + (j + k * LPR) * 8 + 4);
        f[k][0] += a.x; f[k][1] += a.y; f[k][2] += a.z; f[k][3] += a.w;
        f[k][4] += b.x; f[k][5] += b.y; f[k][6] += b.z; f[k][7] += b.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sm += f[k][e];
    }
    for (int o = LPR >> 1; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    const float mean = sm * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dl = f[k][e] - mean; sq += dl * dl; }
    for (int o = LPR >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = rsqrtf(sq * inv_c + eps);
    if (ok) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c0 = (j + k * LPR) * 8;
        const float4 g0 = *(const float4*)(gb + c0), g1 = *(const float4*)(gb + c0 + 4);
        const float4 b0 = *(const float4*)(gb + C + c0), b1 = *(const float4*)(gb + C + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f[k][e] - mean) * rstd * gg[e] + bb[e];
        *(uint4*)(y + m * C + c0) = pack_elx8(o);
      }
    }
  }
}

}  // namespace

extern "C" int ctrlv_softmax_rows(const float* scores, int rows, int cols, long ld_scores, void* probs, long ld_probs,
                                  ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(scores && probs, "softmax_rows: null pointer");
  CTRLV_CHECK_SHAPE(rows > 0 && cols > 0 && cols % 4 == 0 && cols <= kSmxPer * 256 && ld_scores % 4 == 0 && ld_probs % 4 == 0,
                    "softmax_rows: cols=%d must be a multiple of 4, <= %d", cols, kSmxPer * 256);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, scores, cols, ld_scores,
                     (el_t*)probs, ld_probs);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_groupnorm_chunks(int n_img, int S, int C, int imgs_per_stat) {
  GnShape s;
  int rc = gn_shape(n_img, S, C, imgs_per_stat, 0, false, &s);
  return rc < 0 ? rc : s.n_chunks;
}

extern "C" int ctrlv_groupnorm_stats(const void* x, const void* x2, int c_split, int n_img, int S, int C,
                                     int imgs_per_stat, float eps, float* partials, ctrlv_stream_t stream) {
  return ctrlv_groupnorm_stats_split(x, nullptr, x2, nullptr, c_split, n_img, S, C, imgs_per_stat, eps, partials, stream);
}
extern "C" int ctrlv_groupnorm_stats_split(const void* x, const void* x_lo, const void* x2, const void* x2_lo, int c_split,
                                           int n_img, int S, int C, int imgs_per_stat, float eps, float* partials,
                                           ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && partials, "groupnorm_stats: null pointer");
  CTRLV_CHECK_ARG(x2 || !x2_lo, "groupnorm_stats: x2_lo without x2");
  const GnLo lo{(const lo_t*)x_lo, (const lo_t*)x2_lo};
  GnShape s;
  int rc = gn_shape(n_img, S, C, imgs_per_stat, c_split, x2 != nullptr, &s);
  if (rc < 0) return rc;
  const int nt = s.CV * s.RPP;
  const size_t smem = ((size_t)s.RPP * C * 2 + C) * sizeof(float);
  if (x_lo || x2_lo)
    hipLaunchKernelGGL(gn_stats_kernel<true>, dim3(s.n_chunks, n_img), dim3(nt), smem, (hipStream_t)stream,
                       (const el_t*)x, (const el_t*)x2, lo, s, partials);
  else
    hipLaunchKernelGGL(gn_stats_kernel<false>, dim3(s.n_chunks, n_img), dim3(nt), smem, (hipStream_t)stream,
                       (const el_t*)x, (const el_t*)x2, lo, s, partials);
  CTRLV_LAUNCH_CHECK();
  // (mean, rstd) per (statistics row, group), behind the chunk partials
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_img / imgs_per_stat, 32 / kFinGroups), dim3(1024), 0, (hipStream_t)stream, s,
                     partials, eps, partials + (size_t)n_img * s.n_chunks * 64);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_groupnorm_apply(const void* x, const void* x2, int c_split, int n_img, int S, int C,
                                     int imgs_per_stat, const float* partials, const float* gamma, const float* beta,
                                     int silu, void* y, ctrlv_stream_t stream) {
  return ctrlv_groupnorm_apply_split(x, nullptr, x2, nullptr, c_split, n_img, S, C, imgs_per_stat, partials, gamma, beta, silu,
                                     y, stream);
}
extern "C" int ctrlv_groupnorm_apply_split(const void* x, const void* x_lo, const void* x2, const void* x2_lo, int c_split,
                                           int n_img, int S, int C, int imgs_per_stat, const float* partials,
                                           const float* gamma, const float* beta, int silu, void* y, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && partials && gamma && beta && y, "groupnorm_apply: null pointer");
  CTRLV_CHECK_ARG(x2 || !x2_lo, "groupnorm_apply: x2_lo without x2");
  const GnLo lo{(const lo_t*)x_lo, (const lo_t*)x2_lo};
  GnShape s;
  int rc = gn_shape(n_img, S, C, imgs_per_stat, c_split, x2 != nullptr, &s);
  if (rc < 0) return rc;
  const int nt = s.CV * s.RPP;
  if (x_lo || x2_lo)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(s.n_chunks, n_img), dim3(nt), 0, (hipStream_t)stream, (const el_t*)x,
                       (const el_t*)x2, lo, s, partials + (size_t)n_img * s.n_chunks * 64, gamma, beta, silu, (el_t*)y);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(s.n_chunks, n_img), dim3(nt), 0, (hipStream_t)stream, (const el_t*)x,
                       (const el_t*)x2, lo, s, partials + (size_t)n_img * s.n_chunks * 64, gamma, beta, silu, (el_t*)y);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

// GroupNorm whose chunk partials were written by the producing GEMM's epilogue (ctrlv_gemm_desc.gn_partials: 64-row chunks,
// gemm_pp_kernel.h GNS): finalize + apply, no statistics pass over the tensor.
extern "C" int ctrlv_groupnorm_from_partials(const void* x, int n_img, int S, int C, int imgs_per_stat, float eps,
                                             float* partials, const float* gamma, const float* beta, int silu, void* y,
                                             ctrlv_stream_t stream) {
  return ctrlv_groupnorm_from_partials_split(x, nullptr, n_img, S, C, imgs_per_stat, eps, partials, gamma, beta, silu, y, stream);
}
extern "C" int ctrlv_groupnorm_from_partials_split(const void* x, const void* x_lo, int n_img, int S, int C, int imgs_per_stat,
                                                   float eps, float* partials, const float* gamma, const float* beta, int silu,
                                                   void* y, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && partials && gamma && beta && y, "groupnorm_from_partials: null pointer");
  CTRLV_CHECK_SHAPE(S > 0 && S % 64 == 0, "groupnorm_from_partials: S=%d must be a multiple of 64", S);
  GnShape s;
  int rc = gn_shape(n_img, S, C, imgs_per_stat, 0, false, &s);
  if (rc < 0) return rc;
  GnShape sp = s;                       // the producer's chunking
  sp.rows_per_chunk = 64;
  sp.n_chunks = S / 64;
  float* stats = partials + (size_t)n_img * sp.n_chunks * 64;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n_img / imgs_per_stat, 32 / kFinGroups), dim3(1024), 0, (hipStream_t)stream, sp,
                     partials, eps, stats);
  CTRLV_LAUNCH_CHECK();
  const int nt = s.CV * s.RPP;
  if (x_lo)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(s.n_chunks, n_img), dim3(nt), 0, (hipStream_t)stream, (const el_t*)x,
                       (const el_t*)nullptr, GnLo{(const lo_t*)x_lo, nullptr}, s, stats, gamma, beta, silu, (el_t*)y);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(s.n_chunks, n_img), dim3(nt), 0, (hipStream_t)stream, (const el_t*)x,
                       (const el_t*)nullptr, GnLo{nullptr, nullptr}, s, stats, gamma, beta, silu, (el_t*)y);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_layernorm(const void* x, int M, int C, const float* gamma, const float* beta, float eps,
                               const float* V, int vdiv, int vmod, int ldv, void* y, ctrlv_stream_t stream) {
  return ctrlv_layernorm_split(x, nullptr, M, C, gamma, beta, eps, V, vdiv, vmod, ldv, y, stream);
}
extern "C" int ctrlv_layernorm_split(const void* x, const void* x_lo, int M, int C, const float* gamma, const float* beta,
                                     float eps, const float* V, int vdiv, int vmod, int ldv, void* y, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && gamma && beta && y, "layernorm: null pointer");
  CTRLV_CHECK_SHAPE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "layernorm: C=%d must be a multiple of 8, <= 2048", C);
  if (V) CTRLV_CHECK_ARG(vdiv > 0 && vmod > 0 && ldv >= C, "layernorm: bad row-vector table");
  const int nv = (C / 8 + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  // C = 320 / 640 / 1280 (every LayerNorm of the model): 8 / 16 / 32 lanes per row, 5 chunks per lane (ln_rows_kernel)
  const int rows_path = ctrlv_debug().ln_rows;
  const int lpr = C / 40;
  if (rows_path && C % 40 == 0 && (lpr == 8 || lpr == 16 || lpr == 32) && (!V || (ldv % 4 == 0 && ((uintptr_t)V & 15) == 0))) {
    const int lg = lpr == 8 ? 3 : (lpr == 16 ? 4 : 5), rpb = 4 * (64 >> lg);
    long nb = ((long)M + rpb - 1) / rpb;
    if (nb > 256 * 8) nb = 256 * 8;
    if (x_lo)
      hipLaunchKernelGGL((ln_rows_kernel<5, true>), dim3((unsigned)nb), dim3(256), 2 * C * sizeof(float), st, (const el_t*)x,
                         (const lo_t*)x_lo, M, C, lg, gamma, beta, eps, V, vdiv, vmod, ldv, (el_t*)y);
    else
      hipLaunchKernelGGL((ln_rows_kernel<5, false>), dim3((unsigned)nb), dim3(256), 2 * C * sizeof(float), st, (const el_t*)x,
                         (const lo_t*)nullptr, M, C, lg, gamma, beta, eps, V, vdiv, vmod, ldv, (el_t*)y);
    CTRLV_LAUNCH_CHECK();
    return CTRLV_OK;
  }
  long blocks = ((long)M + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
#define LN_LAUNCH(NV)                                                                                             \
  hipLaunchKernelGGL(ln_kernel<NV>, dim3((unsigned)blocks), dim3(256), 0, st, (const el_t*)x, (const lo_t*)x_lo, M, C, gamma, beta, \
                     eps, V, vdiv, vmod, ldv, (el_t*)y)
  switch (nv) {
    case 1: LN_LAUNCH(1); break;
    case 2: LN_LAUNCH(2); break;
    case 3: LN_LAUNCH(3); break;
    default: LN_LAUNCH(4); break;
  }
#undef LN_LAUNCH
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
