// Layout / element-wise kernels of the denoising path (gfx950).  All HBM-bound; 16-byte vectors where the layout
// allows it.  See include/ctrlv_hip.h for the reference call sites each one replaces.
#include "common.h"

namespace {

__device__ __forceinline__ float load_any(const void* p, int dtype, long i) {
  if (dtype == 0) return ((const float*)p)[i];
  if (dtype == 1) return (float)((const _Float16*)p)[i];
  return bf16_to_f32(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void store_any(void* p, int dtype, long i, float v) {
  if (dtype == 0) ((float*)p)[i] = v;
  else if (dtype == 1) ((_Float16*)p)[i] = (_Float16)v;
  else ((bf16_t*)p)[i] = f32_to_bf16(v);
}

// one thread per (image, pixel): reads are coalesced over pixels for every channel plane
__global__ void nchw_to_rows_kernel(const void* __restrict__ src, int dtype, int n_img, int C, int HW,
                                    el_t* __restrict__ dst, int ldc, int c_off) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_img * HW) return;
  const long n = idx / HW, p = idx % HW;
  el_t* d = dst + idx * ldc + c_off;
  for (int c = 0; c < C; ++c) d[c] = f32_to_el(load_any(src, dtype, (n * C + c) * HW + p));
}

__global__ void rows_to_nchw_kernel(const el_t* __restrict__ src, int ldc, int n_img, int C, int HW,
                                    void* __restrict__ dst, int dtype) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_img * HW) return;
  const long n = idx / HW, p = idx % HW;
  const el_t* s = src + idx * ldc;
  for (int c = 0; c < C; ++c) store_any(dst, dtype, (n * C + c) * HW + p, el_to_f32(s[c]));
}

// Conv3d(C, C, (3, 1, 1), padding (1, 0, 0)) over the frames of ONE clip on channels-last rows with C <= 4 channels, written
// as NCHW: the `time_conv_out` that ends AutoencoderKLTemporalDecoder.decode (3 -> 3 channels).  One thread per
// (frame, pixel): 3 taps x C inputs, C outputs, fp32 (the reference runs this conv in the VAE's dtype on NCHW tensors).
__global__ void time_conv_rows_to_nchw_kernel(const el_t* __restrict__ src, int ldc, int n_frames, int C, int HW,
                                              const float* __restrict__ w, const float* __restrict__ bias,
                                              void* __restrict__ dst, int dtype) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_frames * HW) return;
  const int f = (int)(idx / HW);
  const long p = idx % HW;
  float acc[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[o] = o < C ? bias[o] : 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int ff = f + t - 1;
    if (ff < 0 || ff >= n_frames) continue;
    const el_t* s = src + ((long)ff * HW + p) * ldc;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c >= C) break;
      const float x = el_to_f32(s[c]);
#pragma unroll
      for (int o = 0; o < 4; ++o)
        if (o < C) acc[o] = __builtin_fmaf(w[(o * C + c) * 3 + t], x, acc[o]);
    }
  }
  for (int o = 0; o < C; ++o) store_any(dst, dtype, ((long)f * C + o) * HW + p, acc[o]);
}

// tiled transpose for wide tensors (foreign NCHW ControlNet residuals): 64 pixels x 64 channels per block
__global__ __launch_bounds__(256) void nchw_to_rows_tiled_kernel(const void* __restrict__ src, int dtype, int C,
                                                                 int HW, el_t* __restrict__ dst, int ldc,
                                                                 int c_off) {
  __shared__ float tile[64][65];
  const long n = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < 64; c += 4)
    tile[c][tx] = (c0 + c < C && p0 + tx < HW) ? load_any(src, dtype, (n * C + c0 + c) * HW + p0 + tx) : 0.f;
  __syncthreads();
  for (int p = ty; p < 64; p += 4)
    if (p0 + p < HW && c0 + tx < C) dst[(n * HW + p0 + p) * ldc + c_off + c0 + tx] = f32_to_el(tile[tx][p]);
}
__global__ __launch_bounds__(256) void rows_to_nchw_tiled_kernel(const el_t* __restrict__ src, int ldc, int C,
                                                                 int HW, void* __restrict__ dst, int dtype) {
  __shared__ float tile[64][65];
  const long n = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int p = ty; p < 64; p += 4)
    tile[p][tx] = (p0 + p < HW && c0 + tx < C) ? el_to_f32(src[(n * HW + p0 + p) * ldc + c0 + tx]) : 0.f;
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < C && p0 + tx < HW) store_any(dst, dtype, (n * C + c0 + c) * HW + p0 + tx, tile[tx][c]);
}

__global__ void im2col3x3_kernel(const el_t* __restrict__ x, int n_img, int H, int W, int Cp, el_t* __restrict__ col,
                                 int Kp) {
  const int cpr = Kp >> 3;  // 16-B chunks per output row
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)n_img * H * W * cpr;
  if (idx >= total) return;
  const long m = idx / cpr;
  const int j = (int)(idx % cpr);
  const int k0 = j * 8, tap = k0 / Cp, c0 = k0 % Cp;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (tap < 9) {
    const int hw = H * W;
    const long n = m / hw;
    const int rem = (int)(m % hw), y = rem / W, xx = rem % W;
    const int yi = y + tap / 3 - 1, xi = xx + tap % 3 - 1;
    if (yi >= 0 && yi < H && xi >= 0 && xi < W) v = *(const uint4*)(x + ((n * H + yi) * W + xi) * Cp + c0);
  }
  *(uint4*)(col + m * Kp + k0) = v;
}

// (x / y carry no __restrict__: in place -- y == x -- is part of the contract, ctrlv_hip.h; plan.hip add_res uses it)
__global__ void axpby_kernel(const el_t* x, const el_t* __restrict__ r, float a, float b, el_t* y, size_t n) {
  const size_t nv = n >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    float fx[8], fr[8];
    unpack_elx8(((const uint4*)x)[i], fx);
    unpack_elx8(((const uint4*)r)[i], fr);
#pragma unroll
    for (int e = 0; e < 8; ++e) fx[e] = a * fx[e] + b * fr[e];
    ((uint4*)y)[i] = pack_elx8(fx);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const size_t i = (nv << 3) + threadIdx.x;
    y[i] = f32_to_el(a * el_to_f32(x[i]) + b * el_to_f32(r[i]));
  }
}

// (y, y_lo) = split(a * (x + x_lo) + b * r) on a SPLIT skip tensor (common.h split_lo8); n a multiple of 8
// (in place allowed: y == x and ylo == xlo, so none of the four carries __restrict__)
__global__ void axpby_split_kernel(const el_t* x, const lo_t* xlo, const el_t* __restrict__ r, float a, float b, el_t* y,
                                   lo_t* ylo, size_t n) {
  const size_t nv = n >> 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    float fx[8], fr[8];
    unpack_elx8(((const uint4*)x)[i], fx);
    if (xlo) add_lo8(fx, ((const uint2*)xlo)[i]);
    unpack_elx8(((const uint4*)r)[i], fr);
#pragma unroll
    for (int e = 0; e < 8; ++e) fx[e] = a * fx[e] + b * fr[e];
    const uint4 hi = pack_elx8(fx);
    ((uint4*)y)[i] = hi;
    ((uint2*)ylo)[i] = split_lo8(fx, hi);
  }
}

__global__ void silu_kernel(const el_t* __restrict__ x, el_t* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = f32_to_el(silu_f(el_to_f32(x[i])));
}

__global__ void timestep_embedding_kernel(const float* __restrict__ t, int n, int dim, el_t* __restrict__ out) {
  const int half = dim >> 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * half) return;
  const int r = idx / half, i = idx % half;
  const float freq = expf(-9.210340371976184f * (float)i / (float)half);  // ln(10000)
  const float arg = t[r] * freq;
  float s, c;
  sincosf(arg, &s, &c);
  out[(long)r * dim + i] = f32_to_el(c);          // flip_sin_to_cos: [cos | sin]
  out[(long)r * dim + half + i] = f32_to_el(s);
}

__global__ void cfg_euler_kernel(float* __restrict__ lat, const void* __restrict__ pred, int pred_dtype, int cfg,
                                 const float* __restrict__ guidance, int B, int F, int CHW, float sigma,
                                 float sigma_next, el_t* __restrict__ scaled_next) {
  const long total = (long)B * F * CHW;
  const float c_out = -sigma / sqrtf(sigma * sigma + 1.0f);
  const float c_skip = 1.0f / (sigma * sigma + 1.0f);
  const float dt = sigma_next - sigma;
  const float inv_next = 1.0f / sqrtf(sigma_next * sigma_next + 1.0f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int f = (int)((i / CHW) % F);
    float v = load_any(pred, pred_dtype, cfg ? total + i : i);
    if (cfg) {
      const float u = load_any(pred, pred_dtype, i);
      // the reference combines in the model dtype: round the guided prediction like `noise_pred` would be
      v = u + guidance[f] * (v - u);
      if (pred_dtype == 2) v = bf16_to_f32(f32_to_bf16(v));
      else if (pred_dtype == 1) v = (float)(_Float16)v;
    }
    const float x = lat[i];
    const float x0 = v * c_out + x * c_skip;
    const float deriv = (x - x0) / sigma;
    const float xn = x + deriv * dt;
    lat[i] = xn;
    if (scaled_next) scaled_next[i] = f32_to_el(xn * inv_next);
  }
}

}  // namespace

static inline unsigned grid_for(size_t n, int bs, unsigned cap = 256 * 16) {
  size_t g = (n + bs - 1) / bs;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}

extern "C" int ctrlv_nchw_to_rows(const void* src, int src_dtype, int n_img, int C, int HW, void* dst, int ldc,
                                  int c_off, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(src && dst && src_dtype >= 0 && src_dtype <= 2, "nchw_to_rows: bad arguments");
  CTRLV_CHECK_SHAPE(n_img > 0 && C > 0 && HW > 0 && c_off >= 0 && c_off + C <= ldc, "nchw_to_rows: bad shape");
  if (C >= 32) {
    dim3 grid((HW + 63) / 64, (C + 63) / 64, n_img);
    hipLaunchKernelGGL(nchw_to_rows_tiled_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, src_dtype, C, HW,
                       (el_t*)dst, ldc, c_off);
  } else {
    const long total = (long)n_img * HW;
    hipLaunchKernelGGL(nchw_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, src_dtype, n_img, C, HW, (el_t*)dst, ldc, c_off);
  }
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_rows_to_nchw(const void* src, int ldc, int n_img, int C, int HW, void* dst, int dst_dtype,
                                  ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(src && dst && dst_dtype >= 0 && dst_dtype <= 2, "rows_to_nchw: bad arguments");
  CTRLV_CHECK_SHAPE(n_img > 0 && C > 0 && HW > 0 && C <= ldc, "rows_to_nchw: bad shape");
  if (C >= 32) {
    dim3 grid((HW + 63) / 64, (C + 63) / 64, n_img);
    hipLaunchKernelGGL(rows_to_nchw_tiled_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const el_t*)src, ldc, C,
                       HW, dst, dst_dtype);
  } else {
    const long total = (long)n_img * HW;
    hipLaunchKernelGGL(rows_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const el_t*)src, ldc, n_img, C, HW, dst, dst_dtype);
  }
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_time_conv_rows_to_nchw(const void* src, int ldc, int n_frames, int C, int HW, const float* weight,
                                            const float* bias, void* dst, int dst_dtype, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(src && dst && weight && bias && dst_dtype >= 0 && dst_dtype <= 2, "time_conv_rows_to_nchw: bad arguments");
  CTRLV_CHECK_SHAPE(n_frames > 0 && HW > 0 && C > 0 && C <= 4 && C <= ldc, "time_conv_rows_to_nchw: 1 <= C <= 4 channels");
  const long total = (long)n_frames * HW;
  hipLaunchKernelGGL(time_conv_rows_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const el_t*)src, ldc, n_frames, C, HW, weight, bias, dst, dst_dtype);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_im2col3x3(const void* x, int n_img, int H, int W, int Cp, void* col, int Kp,
                               ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && col, "im2col3x3: null pointer");
  CTRLV_CHECK_SHAPE(n_img > 0 && H > 0 && W > 0 && Cp > 0 && Cp % 8 == 0 && Kp % 64 == 0 && Kp >= 9 * Cp,
                    "im2col3x3: Cp must be a multiple of 8 and Kp a multiple of 64 >= 9*Cp");
  const long total = (long)n_img * H * W * (Kp / 8);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const el_t*)x, n_img, H, W, Cp, (el_t*)col, Kp);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_axpby(const void* x, const void* r, float a, float b, void* y, size_t n, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && r && y && n > 0, "axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const el_t*)x, (const el_t*)r, a, b, (el_t*)y, n);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_axpby_split(const void* x, const void* x_lo, const void* r, float a, float b, void* y, void* y_lo,
                                 size_t n, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && r && y && y_lo && n > 0 && n % 8 == 0, "axpby_split: bad arguments (n must be a multiple of 8)");
  hipLaunchKernelGGL(axpby_split_kernel, dim3(grid_for(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, (const el_t*)x,
                     (const lo_t*)x_lo, (const el_t*)r, a, b, (el_t*)y, (lo_t*)y_lo, n);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_silu(const void* x, void* y, size_t n, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(x && y && n > 0, "silu: bad arguments");
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const el_t*)x,
                     (el_t*)y, n);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_timestep_embedding(const float* t, int n, int dim, void* out, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(t && out, "timestep_embedding: null pointer");
  CTRLV_CHECK_SHAPE(n > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: dim must be even");
  const int total = n * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, n,
                     dim, (el_t*)out);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}

extern "C" int ctrlv_cfg_euler_step(float* latents, const void* noise_pred, int pred_dtype, int cfg,
                                    const float* guidance, int B, int F, int CHW, float sigma, float sigma_next,
                                    void* scaled_next, ctrlv_stream_t stream) {
  CTRLV_CHECK_ARG(latents && noise_pred && pred_dtype >= 0 && pred_dtype <= 2, "cfg_euler_step: bad arguments");
  CTRLV_CHECK_ARG(!cfg || guidance, "cfg_euler_step: guidance table required with cfg");
  CTRLV_CHECK_SHAPE(B > 0 && F > 0 && CHW > 0 && sigma > 0.f, "cfg_euler_step: bad shape / sigma");
  const size_t total = (size_t)B * F * CHW;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, latents,
                     noise_pred, pred_dtype, cfg, guidance, B, F, CHW, sigma, sigma_next, (el_t*)scaled_next);
  CTRLV_LAUNCH_CHECK();
  return CTRLV_OK;
}
