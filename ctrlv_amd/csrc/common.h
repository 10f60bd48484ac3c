// Shared device/host helpers for libctrlv_hip.so (gfx950 only: wave64, MFMA 32x32x16 bf16, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ctrlv_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef uint16_t bf16_t;  // raw storage

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 256 B of zeros in device memory: the source of every out-of-bounds / padding row of an LDS-DMA gather.
// (one copy per translation unit: the library is built without relocatable device code)
static __device__ __attribute__((aligned(256), used)) unsigned char g_ctrlv_zeros[256];

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the hardware convert (keeps NaN a NaN, see MI355X_MICROARCH correctness table)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack_bf16x8(const float* f) {
  uint4 v;
  v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
  v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
  return v;
}
// silu(x) = x * sigmoid(x) with the raw v_exp_f32 / v_rcp_f32 (1 ulp each); x -> -inf gives x * 0.
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
// exact-erf GELU, 0.5*x*(1+erf(x/sqrt2)), through erfc(z) = t*(a1+t*(a2+t*(a3+t*(a4+t*a5))))*exp(-z^2), t = 1/(1+p*z)
// (Abramowitz & Stegun 7.1.26, |abs err| <= 1.5e-7).  Written as x*(1 - erfc/2) for x >= 0 and x*erfc/2 for x < 0,
// so there is no cancellation on the negative tail.  13 VALU instructions, 2 of them transcendental (libm erff: ~30).
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
  const float q = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float hc = 0.5f * q * e;           // erfc(z)/2
  return x * (x >= 0.f ? 1.0f - hc : hc);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// XCD-aware bijective block remap (guide T1): blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous chunk of the logical tile order to keep neighbouring tiles in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- host side ----
void ctrlv_set_error(const char* fmt, ...);
#define CTRLV_CHECK_ARG(cond, ...)                  \
  do {                                              \
    if (!(cond)) {                                  \
      ctrlv_set_error(__VA_ARGS__);                 \
      return CTRLV_E_BAD_ARG;                       \
    }                                               \
  } while (0)
#define CTRLV_CHECK_SHAPE(cond, ...)                \
  do {                                              \
    if (!(cond)) {                                  \
      ctrlv_set_error(__VA_ARGS__);                 \
      return CTRLV_E_BAD_SHAPE;                     \
    }                                               \
  } while (0)
#define CTRLV_HIP_TRY(expr)                                                             \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess) {                                                            \
      ctrlv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return CTRLV_E_HIP;                                                               \
    }                                                                                   \
  } while (0)
#define CTRLV_LAUNCH_CHECK() CTRLV_HIP_TRY(hipGetLastError())
