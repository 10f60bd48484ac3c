// Shared device/host helpers for libctrlv_hip.so / libctrlv_hip_f16.so (gfx950 only: wave64, MFMA 32x32x16, LDS-DMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ctrlv_hip.h"

// ---- the activation / weight ELEMENT type.  The library is built twice from these sources:
//   libctrlv_hip.so      el = bf16 (default)             -- BASELINE.json's dtype; inference and the training step
//   libctrlv_hip_f16.so  el = IEEE fp16 (-DCTRLV_ELEM_F16) -- the reference's own autocast dtype (config/a100l.yaml:9,
//                        tools/eval_video_controlnet.py:110-118): same bytes, same MFMA rate, 3 more mantissa bits, which
//                        is what north_star's 1e-3 model-level tolerance needs.
// Everything that touches an element goes through the helpers below; kernels are otherwise type-agnostic (16-bit moves,
// LDS-DMA, ds_read_b64_tr_b16).  fp32 accumulation / statistics in both builds.
#ifdef CTRLV_ELEM_F16
typedef _Float16 el_native_t;
#define CTRLV_ELEM_DTYPE 1                                    /* the ABI's dtype code of the element type */
#define CTRLV_MFMA_32x32x16_ASM "v_mfma_f32_32x32x16_f16"
#else
typedef __bf16 el_native_t;
#define CTRLV_ELEM_DTYPE 2
#define CTRLV_MFMA_32x32x16_ASM "v_mfma_f32_32x32x16_bf16"
#endif
typedef __attribute__((ext_vector_type(8))) el_native_t elx8;
typedef __attribute__((ext_vector_type(4))) el_native_t elx4;
typedef __attribute__((ext_vector_type(2))) el_native_t elx2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef uint16_t el_t;    // raw storage of one element
typedef uint16_t bf16_t;  // raw storage of a genuine bf16 value (dtype code 2 at the model boundary, whatever el is)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 256 B of zeros in device memory: the source of every out-of-bounds / padding row of an LDS-DMA gather.
// (one copy per translation unit: the library is built without relocatable device code)
static __device__ __attribute__((aligned(256), used)) unsigned char g_ctrlv_zeros[256];

// D = A . B + C on the matrix pipe, 32 x 32 x 16, element-type operands, fp32 accumulate
__device__ __forceinline__ f32x16 mfma_32x32x16(const elx8& a, const elx8& b, const f32x16& c) {
#ifdef CTRLV_ELEM_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// genuine bf16 <-> fp32 (boundary dtype code 2): round-to-nearest-even through the hardware convert (keeps NaN a NaN,
// see MI355X_MICROARCH correctness table)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
#ifdef CTRLV_ELEM_F16
// IEEE fp16, round-to-nearest-even (v_cvt_f16_f32; overflow -> inf like the reference's autocast: the largest stored
// |value| of the path is < 10, profiles/r04_storage_precision_study.txt)
__device__ __forceinline__ float el_to_f32(el_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ el_t f32_to_el(float f) { return __builtin_bit_cast(el_t, (_Float16)f); }
__device__ __forceinline__ uint32_t pack_elx2(float lo, float hi) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack_elx8(const uint4& v, float* f) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 a = __builtin_bit_cast(h2, v.x), b = __builtin_bit_cast(h2, v.y), c = __builtin_bit_cast(h2, v.z),
           d = __builtin_bit_cast(h2, v.w);
  f[0] = (float)a.x; f[1] = (float)a.y; f[2] = (float)b.x; f[3] = (float)b.y;
  f[4] = (float)c.x; f[5] = (float)c.y; f[6] = (float)d.x; f[7] = (float)d.y;
}
#else
__device__ __forceinline__ float el_to_f32(el_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ el_t f32_to_el(float f) { return f32_to_bf16(f); }
__device__ __forceinline__ uint32_t pack_elx2(float lo, float hi) {
  // (as ONE two-element conversion: v_cvt_pk_bf16_f32 d, lo, hi.  Written as two scalar conversions joined by shift and or
  //  it compiled to two v_cvt_pk_bf16_f32 x, 0 plus v_lshlrev + v_or_sdwa: four instructions per pair in every epilogue)
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  const b2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack_elx8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
#endif
// the two 16-bit halves of a packed dword as fp32 (the 2-stage epilogue's residual reads)
__device__ __forceinline__ float el_lo_f32(uint32_t v) { return el_to_f32((el_t)(v & 0xffffu)); }
__device__ __forceinline__ float el_hi_f32(uint32_t v) { return el_to_f32((el_t)(v >> 16)); }
__device__ __forceinline__ uint4 pack_elx8(const float* f) {
  uint4 v;
  v.x = pack_elx2(f[0], f[1]); v.y = pack_elx2(f[2], f[3]);
  v.z = pack_elx2(f[4], f[5]); v.w = pack_elx2(f[6], f[7]);
  return v;
}
// ---- SPLIT storage of the residual trunk (round 5; DESIGN.md 4 "trunk").  A trunk tensor may carry a second plane of
// the same shape: hi = rne_el(v), lo = rne(v - hi).  MFMA consumers read the hi plane in place (it IS the element-rounded
// tensor); residual operands and norm inputs read hi + lo.  v - hi is exact in fp32 (hi is v rounded to fewer bits), so lo
// carries one rounding.  Round 6: the lo plane is ONE BYTE per element -- e5m2 ("bf8": fp16's sign / exponent and its two top
// mantissa bits; v_cvt_pk_bf8_f32 / v_cvt_pk_f32_bf8), i.e. hi + lo keeps ~15 significant bits in 3 bytes.  On the oracle
// (tests/trunk_precision_study.py) that has the model-level error of a 16-bit lo plane (5.84e-4 vs 5.88e-4) at half the extra
// bytes; |lo| <= ulp(hi) / 2 is far inside e5m2's range, values under 2^-16 flush to zero (absolute error < 2^-17).
typedef uint8_t lo_t;
__device__ __forceinline__ uint2 split_lo8(const float* v, const uint4& hi) {
#pragma clang fp contract(off)
  float h[8], dl[8];
  unpack_elx8(hi, h);
#pragma unroll
  for (int e = 0; e < 8; ++e) dl[e] = v[e] - h[e];
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_bf8_f32(dl[0], dl[1], w0, false);
  w0 = __builtin_amdgcn_cvt_pk_bf8_f32(dl[2], dl[3], w0, true);
  w1 = __builtin_amdgcn_cvt_pk_bf8_f32(dl[4], dl[5], w1, false);
  w1 = __builtin_amdgcn_cvt_pk_bf8_f32(dl[6], dl[7], w1, true);
  return make_uint2((unsigned)w0, (unsigned)w1);
}
__device__ __forceinline__ lo_t split_lo1(float v, el_t hi) {
#pragma clang fp contract(off)
  const float dl = v - el_to_f32(hi);
  return (lo_t)(__builtin_amdgcn_cvt_pk_bf8_f32(dl, 0.f, 0, false) & 0xff);
}
// four lo values (one 32-bit word of a lo plane) <-> fp32
__device__ __forceinline__ void unpack_lo4(unsigned w, float* l) {
  typedef float f32x2_lo __attribute__((ext_vector_type(2)));
  const f32x2_lo a = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, true);
  l[0] = a.x; l[1] = a.y; l[2] = b.x; l[3] = b.y;
}
__device__ __forceinline__ unsigned pack_lo4(float l0, float l1, float l2, float l3) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(l0, l1, w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(l2, l3, w, true);
  return (unsigned)w;
}
// the eight lo values of a split tensor as fp32
__device__ __forceinline__ void unpack_lo8(const uint2& lo, float* l) {
  typedef float f32x2_lo __attribute__((ext_vector_type(2)));
  const f32x2_lo a = __builtin_amdgcn_cvt_pk_f32_bf8((int)lo.x, false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)lo.x, true);
  const f32x2_lo c = __builtin_amdgcn_cvt_pk_f32_bf8((int)lo.y, false), d = __builtin_amdgcn_cvt_pk_f32_bf8((int)lo.y, true);
  l[0] = a.x; l[1] = a.y; l[2] = b.x; l[3] = b.y; l[4] = c.x; l[5] = c.y; l[6] = d.x; l[7] = d.y;
}
// x = hi + lo of a split tensor (lo may be absent)
__device__ __forceinline__ void add_lo8(float* f, const uint2& lo) {
#pragma clang fp contract(off)
  float l[8];
  unpack_lo8(lo, l);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = f[e] + l[e];
}

// silu(x) = x * sigmoid(x) with the raw v_exp_f32 / v_rcp_f32 (1 ulp each); x -> -inf gives x * 0.
__device__ __forceinline__ float silu_f(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
// erf-GELU  x * Phi(x),  Phi(x) = 0.5*(1+erf(x/sqrt2)),  WITHOUT transcendental instructions: v_exp_f32 / v_rcp_f32 run
// at quarter rate on CDNA and the GEGLU epilogue of the feed-forward GEMMs was bound by them (69 GEMMs per step, 10-30 %
// of their time).  (Phi(x) - 1/2) / x is even and smooth: a degree-12 polynomial in t = 2*x^2/25 - 1 (Chebyshev
// interpolant on |x| <= 5, monomial form, Horner) gives |Phi error| <= 3.9e-7 in fp32 arithmetic; |x| is clamped to 5
// (1 - Phi(5) = 2.9e-7).  |gelu error| <= 2.1e-6 absolute over the whole line -- three orders below the bf16 output
// rounding.  All FMAs, written on float vectors so that the compiler emits v_pk_fma_f32 (2 elements per instruction).
// (Four elements per call: two independent Horner chains interleave, which also fills the 1-wait-state hazard
// between dependent packed-fp32 instructions that the compiler otherwise pads with s_nop.)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// Every fused multiply-add below is EXPLICIT (and contraction is off for the rest): the value of gelu must not depend on
// which kernel inlines this function -- the scalar 2-stage GEMM epilogue and the packed ping-pong epilogue serve the same
// layer at different batch sizes, and "contract where profitable" gave them different roundings in rare elements.
__device__ __forceinline__ f32x4_t fma4(f32x4_t a, f32x4_t b, f32x4_t c) {
  f32x4_t r;
  r.x = __builtin_fmaf(a.x, b.x, c.x); r.y = __builtin_fmaf(a.y, b.y, c.y);
  r.z = __builtin_fmaf(a.z, b.z, c.z); r.w = __builtin_fmaf(a.w, b.w, c.w);
  return r;
}
__device__ __forceinline__ f32x4_t splat4(float v) { return f32x4_t{v, v, v, v}; }
__device__ __forceinline__ f32x4_t gelu_phi4(f32x4_t x) {
#pragma clang fp contract(off)
  f32x4_t xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -5.0f, 5.0f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -5.0f, 5.0f);
  xc.z = __builtin_amdgcn_fmed3f(x.z, -5.0f, 5.0f);
  xc.w = __builtin_amdgcn_fmed3f(x.w, -5.0f, 5.0f);
  const f32x4_t t = fma4(xc * xc, splat4(0.08f), splat4(-1.0f));
  f32x4_t p = fma4(t, splat4(7.353763795e-04f), splat4(-1.676730928e-03f));
  p = fma4(p, t, splat4(1.374596148e-03f));
  p = fma4(p, t, splat4(-2.526916796e-03f));
  p = fma4(p, t, splat4(6.766527425e-03f));
  p = fma4(p, t, splat4(-1.130712498e-02f));
  p = fma4(p, t, splat4(1.623608917e-02f));
  p = fma4(p, t, splat4(-2.321312763e-02f));
  p = fma4(p, t, splat4(3.147675842e-02f));
  p = fma4(p, t, splat4(-4.045128077e-02f));
  p = fma4(p, t, splat4(5.151792988e-02f));
  p = fma4(p, t, splat4(-7.029590756e-02f));
  p = fma4(p, t, splat4(1.413638145e-01f));
  return fma4(xc, p, splat4(0.5f));          // Phi(x)
}
__device__ __forceinline__ f32x4_t gelu_erf4(f32x4_t x) {
#pragma clang fp contract(off)
  return x * gelu_phi4(x);
}
// The same arithmetic, one element at a time (the same fmas in the same order: identical bits).
__device__ __forceinline__ float gelu_phi1(float x) {
#pragma clang fp contract(off)
  const float xc = __builtin_amdgcn_fmed3f(x, -5.0f, 5.0f);
  const float t = __builtin_fmaf(xc * xc, 0.08f, -1.0f);
  float p = __builtin_fmaf(t, 7.353763795e-04f, -1.676730928e-03f);
  p = __builtin_fmaf(p, t, 1.374596148e-03f);
  p = __builtin_fmaf(p, t, -2.526916796e-03f);
  p = __builtin_fmaf(p, t, 6.766527425e-03f);
  p = __builtin_fmaf(p, t, -1.130712498e-02f);
  p = __builtin_fmaf(p, t, 1.623608917e-02f);
  p = __builtin_fmaf(p, t, -2.321312763e-02f);
  p = __builtin_fmaf(p, t, 3.147675842e-02f);
  p = __builtin_fmaf(p, t, -4.045128077e-02f);
  p = __builtin_fmaf(p, t, 5.151792988e-02f);
  p = __builtin_fmaf(p, t, -7.029590756e-02f);
  p = __builtin_fmaf(p, t, 1.413638145e-01f);
  return __builtin_fmaf(xc, p, 0.5f);
}

// ---- Phi(x) from an LDS table (the GEGLU epilogues of the forward GEMMs).
// The polynomial above costs 17 fp32 FMAs per gate (v_pk_fma_f32 issues at half rate, so packing does not help) and the
// GEGLU epilogue of the K = 320 / 640 feed-forward projections was bound by exactly that: 80 gates per lane per tile x
// 17 FMAs x 2 waves per SIMD = ~11 000 of the ~13 000 cycles the epilogue of a 256 x 320 tile took -- 40 % of a K = 320
// tile.  Phi is smooth and bounded, so every GEMM kernel that serves a GEGLU layer builds, once per workgroup, a table
// of (Phi(x_i), Phi(x_i+1) - Phi(x_i)) at x_i = -5.12 + 0.01 i (1024 entries, 8 KiB of LDS, values from gelu_phi1) and
// evaluates Phi by linear interpolation: 4 VALU + one ds_read_b64 + 1 fma.  Interpolation error <= h^2/8 * max|Phi''|
// = 3.0e-6 (|gelu error| <= 1.2e-5 absolute, at |x| ~ 4; the bf16 rounding of the output is >= 2e-3 relative), and the
// clamps reproduce the polynomial's behaviour outside |x| <= 5 (Phi = 2.9e-7 / 1 - 2.9e-7).
// Every operation is explicit, so all kernels produce the same bits (the 2-stage and the ping-pong kernels serve the
// same layer at different batch sizes; tests/test_fullsize_gpu.py clip independence).
constexpr int kGeluTabN = 1024;
constexpr int kGeluTabBytes = kGeluTabN * 8;
__device__ __forceinline__ void gelu_table_fill(char* tab, int tid, int nthreads) {
#pragma clang fp contract(off)
  for (int i = tid; i < kGeluTabN; i += nthreads) {
    const float a = gelu_phi1(__builtin_fmaf((float)i, 0.01f, -5.12f));
    const float b = gelu_phi1(__builtin_fmaf((float)(i + 1), 0.01f, -5.12f));
    *(float2*)(tab + i * 8) = make_float2(a, b - a);
  }
}
__device__ __forceinline__ float gelu_phi_tab(float x, const char* tab) {
#pragma clang fp contract(off)
  float t = __builtin_fmaf(x, 100.0f, 512.0f);
  t = __builtin_amdgcn_fmed3f(t, 0.0f, 1023.99994f);          // (largest float below 1024)
  const float fr = __builtin_amdgcn_fractf(t);
  const int i = (int)t;                                        // truncation == floor: t >= 0
  const float2 e = *(const float2*)(tab + i * 8);
  return __builtin_fmaf(fr, e.y, e.x);
}
// u = a * gelu(g) = a * (g * Phi(g)), the GEGLU gate product in its one fixed operation order
__device__ __forceinline__ float geglu_tab(float a, float g, const char* tab) {
#pragma clang fp contract(off)
  return a * (g * gelu_phi_tab(g, tab));
}
// The same operations in two halves, for epilogues that keep a batch of table reads in flight: geglu_tab(a, g, tab) ==
// geglu_tab_finish(a, g, fr, e) after gelu_tab_lookup(g, tab, fr, e), bit for bit.
__device__ __forceinline__ void gelu_tab_lookup(float x, const char* tab, float& fr, float2& e) {
#pragma clang fp contract(off)
  float t = __builtin_fmaf(x, 100.0f, 512.0f);
  t = __builtin_amdgcn_fmed3f(t, 0.0f, 1023.99994f);
  fr = __builtin_amdgcn_fractf(t);
  e = *(const float2*)(tab + (int)t * 8);
}
__device__ __forceinline__ float geglu_tab_finish(float a, float g, float fr, float2 e) {
#pragma clang fp contract(off)
  // the interpolation as an ordered instruction: left to the compiler it is hoisted to right behind the table read (the
  // wait then sits there too) and paired into v_pk_fma_f32, which takes three v_mov per pair to line the operands up
  float phi;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(phi) : "v"(fr), "v"(e.y), "v"(e.x));
  return a * (g * phi);
}

// acc + a.x + a.y for a packed element pair: v_dot2c_f32_{bf16,f16} against (1, 1) -- one full-rate instruction for two
// fp32 adds (row sums of packed probabilities / gradients beside MFMAs)
typedef __attribute__((ext_vector_type(2))) el_native_t elx2n;
__device__ __forceinline__ float el_pair_sum(elx2n a, float acc) {
  const elx2n ones = {(el_native_t)1.0f, (el_native_t)1.0f};
#ifdef CTRLV_ELEM_F16
  return __builtin_amdgcn_fdot2(a, ones, acc, false);
#else
  return __builtin_amdgcn_fdot2_f32_bf16(a, ones, acc, false);
#endif
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

// ---- in-kernel clock (DIAGNOSTIC BUILD ONLY: -DCTRLV_CLOCK_STAMP, tools/clock_probe.py; the product library contains no
// stamp).  MI355X_MICROARCH "DVFS give-back" item 6: the clock the chip holds inside a kernel is d(s_memtime) /
// d(s_memrealtime) x 100 MHz.  Thread 0 of every workgroup adds its two deltas to a buffer of this translation unit's own
// (nothing else reads it, no output value depends on it); CTRLV_CLOCK_READER(unit) exports its reader.
#ifdef CTRLV_CLOCK_STAMP
static __device__ unsigned long long g_ctrlv_clock[2];
#define CTRLV_CLOCK_BEGIN() \
  const unsigned long long ck_t0_ = __builtin_amdgcn_s_memtime(), ck_r0_ = __builtin_amdgcn_s_memrealtime()
#define CTRLV_CLOCK_END()                                                                  \
  do {                                                                                     \
    if (threadIdx.x == 0) {                                                                \
      atomicAdd(&g_ctrlv_clock[0], __builtin_amdgcn_s_memtime() - ck_t0_);                 \
      atomicAdd(&g_ctrlv_clock[1], __builtin_amdgcn_s_memrealtime() - ck_r0_);             \
    }                                                                                      \
  } while (0)
#define CTRLV_CLOCK_READER(unit)                                                                                   \
  extern "C" int ctrlv_debug_clock_##unit(unsigned long long* out2, int reset) {                                   \
    if (hipDeviceSynchronize() != hipSuccess) return -3;                                                           \
    if (out2 && hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_ctrlv_clock), 16) != hipSuccess) return -3;                 \
    const unsigned long long z[2] = {0, 0};                                                                        \
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_ctrlv_clock), z, 16) != hipSuccess) return -3;                     \
    return 0;                                                                                                      \
  }
#else
#define CTRLV_CLOCK_BEGIN()
#define CTRLV_CLOCK_END()
#define CTRLV_CLOCK_READER(unit)
#endif

// ---- host side ----
// Developer switches: ONE struct, read ONCE from the environment (abi.hip ctrlv_debug()).  Every field is the A/B handle of a
// measured decision; the defaults are the shipped configuration and the only one the test suites exercise.
struct ctrlv_debug_t {
  int w16;          // CTRLV_W16 [1]          0: no layer is given to the 16x16x32 core (gemm.hip w16_tile_of)
  int splitk;       // CTRLV_SPLITK [1]       0: no split contraction of the small-image long-K convs
  int force_tile;   // CTRLV_GEMM_FORCE_TILE [0]  5 / 6: that ping-pong tile for every large launch (tools/shape_table.py)
  int conv_halo;    // CTRLV_CONV_HALO [1]    0: per-tap gather in tap-major K order for the stride-1 3x3 convs
  int gn_fused;     // CTRLV_GN_FUSED [1]     0: every GroupNorm runs its own statistics pass
  int gn_cross;     // CTRLV_GN_CROSS [1]     0: ... except across the res-block -> transformer boundary
  int gn_rev;       // CTRLV_GN_REV [1]       bit 0 / 1: statistics / apply pass walk the tensor from its end
  int gn_rows;      // CTRLV_GN_ROWS [256]    rows per GroupNorm chunk at S >= 4096
  int ln_rows;      // CTRLV_LN_ROWS [1]      0: one row per wave (ln_kernel)
  int ff_fused;     // CTRLV_FF_FUSED [1]     0: the two launches instead of the fused C = 320 feed-forward
  int ff_ln;        // CTRLV_FF_LN [0]        1: LayerNorm folded into the fused feed-forward's prologue
  int pp_balanced;  // CTRLV_PP_BALANCED [1]  0: one persistent workgroup per CU whatever the tile count
  int pp_cgrp;      // CTRLV_PP_CGRP [0]      -1 row-major tile order, 0 traffic model, n = fixed column-group width
  int attn_rows;    // CTRLV_ATTN_ROWS [0]    32 / 64: query rows per wave of the spatial attention (0: by sequence length)
  int temporal_fused;  // CTRLV_TEMPORAL_FUSED [1]  0: q|k|v GEMM + temporal attention + output projection as three launches
  int wgrad_pp;     // CTRLV_WGRAD_PP [1]     0: every weight gradient on the register-staged kernel of backward.hip
  int wgrad_slabs;  // CTRLV_WGRAD_SLABS [0]  n > 0: that many row slabs per wgrad_pp launch (tools/wgrad_bench.py sweeps)
};
const ctrlv_debug_t& ctrlv_debug();
void ctrlv_set_error(const char* fmt, ...);
#define CTRLV_MAX_DEVICES 64
int ctrlv_current_device();   // hipGetDevice clamped to [0, CTRLV_MAX_DEVICES)
int ctrlv_num_cu(int dev);    // cached hipDeviceAttributeMultiprocessorCount
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
