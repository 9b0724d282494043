// Shared device helpers for the VAuLT hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// h16 = the library's 16-bit operand type (GEMM / attention operands, saved activations, data gradients):
//   libvault_hip.so      bf16 (8 significant bits, f32 range)            - the default build
//   libvault_hip_f16.so  IEEE fp16 (11 significant bits, |x| <= 65504)   - compiled from the same sources with -DVAULT_F16
// Both matrix instructions run at the same rate on gfx950 (MI355X_MICROARCH.md "BF16/F16 ... the F16 forms take the same
// cycles"); fp16 operands put logits / loss of the 24-layer stack inside 1e-3 of the fp32 reference (bf16: 4e-3), the
// backward then carries a power-of-two loss scale (engine.py) and conversions SATURATE instead of producing infinities
// (H16_SATURATE below).  Exported names and struct members keep their "bf16" spelling in both builds (one ABI, one header).
#ifdef VAULT_F16
typedef _Float16 h16;
#define MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#else
typedef __bf16 h16;
#define MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#endif
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) h16 h16x4;
typedef __attribute__((ext_vector_type(2))) h16 h16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// D = A(16x32) . B(32x16) + C on the matrix pipe, fp32 accumulate (v_mfma_f32_16x16x32_{bf16,f16})
__device__ __forceinline__ f32x4 mfma16(h16x8 a, h16x8 b, f32x4 c) {
#ifdef VAULT_F16
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// fp16 build: MODE.FP16_OVFL (bit 23) makes every f32 -> f16 conversion of the wave clamp to +-65504 instead of
// overflowing to infinity (true infinities / NaNs pass through): the saturation guard of the producers, at no
// instruction in their loops.  First statement of every kernel that converts to h16.
#ifdef VAULT_F16
#define H16_SATURATE() asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1")
#else
#define H16_SATURATE() ((void)0)
#endif

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))

// 16-byte async global -> LDS copy (global_load_lds_dwordx4).  `lds_dst` must be wave-uniform:
// lane l lands at lds_dst + 16*l; the global address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(void, gsrc), LDS_PTR(void, lds_dst), 16, 0, 0);
}

// Transposed LDS read: per 16-lane group a 4-row x 16-col block of 16-bit elements, lane 4q+p gives
// the address of row q / cols 4p..4p+3, lane i receives column i (rows 0..3).
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds_addr));
}

__device__ __forceinline__ h16x8 cat_tr(s16x4 lo, s16x4 hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(h16x8, v);
}

__device__ __forceinline__ float h16_to_f32(h16 x) { return (float)x; }
__device__ __forceinline__ h16 f32_to_h16(float x) { return (h16)x; }

__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
  h16x2 v = {(h16)lo, (h16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float2 unpack_h16x2(uint32_t u) {
  h16x2 v = __builtin_bit_cast(h16x2, u);
  return make_float2((float)v[0], (float)v[1]);
}

// bf16 hi/lo split of an fp32 value: x ~= hi + lo with |x - hi - lo| <= 2^-17 |x|.  "Split-bf16" GEMMs
// (C = A_hi B_hi + A_lo B_hi + A_hi B_lo, done as ONE bf16 GEMM over a 3x longer contraction with the
// operands laid out [hi | lo | hi] x [hi | hi | lo]) give fp32-class products on the bf16 MFMA path.
__device__ __forceinline__ void split_bf16(float x, h16& hi, h16& lo) {
  hi = (h16)x;
  lo = (h16)(x - (float)hi);
}

// Standard-normal CDF Phi(x) = 0.5 (1 + erf(x / sqrt 2)) through erf's Abramowitz-Stegun 7.1.28
// form 1 - (1 + a1 t + ... + a6 t^6)^-16 (|err| <= 3e-7, i.e. fp32-level, and no exponential: the
// GELU epilogues are VALU-bound, this is ~1/2 the instructions of an exp-based erf).
__device__ __forceinline__ float norm_cdf_f(float x) {
  const float t = fabsf(x) * 0.70710678118654752f;
  float p = 0.0000430638f;
  p = p * t + 0.0002765672f;
  p = p * t + 0.0001520143f;
  p = p * t + 0.0092705272f;
  p = p * t + 0.0422820123f;
  p = p * t + 0.0705230784f;
  p = p * t + 1.0f;
  p = p * p; p = p * p; p = p * p; p = p * p;
  const float h = 0.5f * __builtin_amdgcn_rcpf(p);   // 0.5 * (1 - erf(|x|/sqrt2))
  return x >= 0.f ? 1.0f - h : h;
}
// exact-erf GELU of HF "gelu" (HF:activations.py:70-89) = x Phi(x); derivative Phi(x) + x phi(x)
__device__ __forceinline__ float gelu_f(float x) { return x * norm_cdf_f(x); }
__device__ __forceinline__ float dgelu_f(float x) {
  return norm_cdf_f(x) + x * (0.3989422804014327f * __expf(-0.5f * x * x));
}

// Two-at-a-time forms for the GEMM epilogues (v_pk_fma_f32 / v_pk_mul_f32 process a register pair per
// issue slot; the GELU epilogues are VALU-bound): same polynomial, Phi = 0.5 + copysign(0.5 - h, x).
__device__ __forceinline__ f32x2 norm_cdf_f2(f32x2 x) {
  const f32x2 a = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  const f32x2 t = a * 0.70710678118654752f;
  f32x2 p = {0.0000430638f, 0.0000430638f};
  p = p * t + 0.0002765672f;
  p = p * t + 0.0001520143f;
  p = p * t + 0.0092705272f;
  p = p * t + 0.0422820123f;
  p = p * t + 0.0705230784f;
  p = p * t + 1.0f;
  p = p * p; p = p * p; p = p * p; p = p * p;
  const f32x2 r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
  const f32x2 d = 0.5f - 0.5f * r;   // 0.5 erf(|x|/sqrt2) >= 0
  return f32x2{0.5f + __builtin_copysignf(d[0], x[0]), 0.5f + __builtin_copysignf(d[1], x[1])};
}
// y = gelu(x), dy = gelu'(x) = Phi(x) + x phi(x).  gelu' needs e = exp(-x^2 / 2) anyway, and with u = |x| / sqrt 2 that is
// exp(-u^2): the erf of Abramowitz-Stegun 7.1.26, 1 - (a1 t + .. + a5 t^5) exp(-u^2) with t = 1 / (1 + p u) (|err| <= 1.5e-7),
// comes out of the same exponential - 26 instead of 31 issue slots per pair against the exponential-free form above plus
// its own exp (the GELU epilogue of the FFN-in forward is VALU-bound: ~9 of a tile's 32 us).
#ifndef VAULT_GELU_SHARED_EXP
#define VAULT_GELU_SHARED_EXP 1
#endif
__device__ __forceinline__ void gelu_fwd_f2(f32x2 x, f32x2& y, f32x2& dy) {
#if VAULT_GELU_SHARED_EXP
  const f32x2 a = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  const f32x2 den = a * (0.3275911f * 0.70710678118654752f) + 1.0f;
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 pl = t * 1.061405429f + (-1.453152027f);
  pl = pl * t + 1.421413741f;
  pl = pl * t + (-0.284496736f);
  pl = pl * t + 0.254829592f;
  pl = pl * t;
  const f32x2 q = (x * x) * (-0.5f * 1.4426950408889634f);
  const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const f32x2 d = 0.5f - 0.5f * (pl * e);          // 0.5 erf(|x| / sqrt 2) >= 0
  const f32x2 cdf = {0.5f + __builtin_copysignf(d[0], x[0]), 0.5f + __builtin_copysignf(d[1], x[1])};
#else
  const f32x2 cdf = norm_cdf_f2(x);
  const f32x2 q = (x * x) * (-0.5f * 1.4426950408889634f);
  const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
#endif
  y = x * cdf;
  dy = cdf + x * (e * 0.3989422804014327f);
}

// MXFP8 (OCP microscaling) quantisation of 8 consecutive elements of a 32-element block held by 4 lanes: `amax` is the block's
// max |x| (already reduced over the four).  Shared exponent = floor(log2(amax)) - emax(e4m3 = 8), clamped to E8M0's range
// (amax == 0 -> the smallest scale); elements round to nearest even and saturate at +-448.  Returns the 8 e4m3 bytes,
// e8 = the block's E8M0 scale byte.
__device__ __forceinline__ uint2 mx8_quant8(const float (&x)[8], float amax, int& e8) {
  e8 = 0;
  if (amax > 0.f) {
    const int ex = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xff);   // (16-bit inputs: never f32-subnormal unless 0)
    e8 = ex - 8;
    e8 = e8 < 0 ? 0 : (e8 > 254 ? 254 : e8);
  }
  const float inv = __builtin_bit_cast(float, (uint32_t)(254 - e8) << 23);   // 2^(127 - e8); e8 = 0 -> 2^127
  float y[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) y[e] = fminf(fmaxf(x[e] * inv, -448.f), 448.f);
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[4], y[5], w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y[6], y[7], w1, true);
  return uint2{(uint32_t)w0, (uint32_t)w1};
}

// counter-based keep/drop decision for dropout: a 32-bit mix of (seed, stream, element index).
// The same function regenerates the mask in backward, so no mask tensor is stored.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool dropout_keep(uint32_t seed, uint32_t stream, uint32_t idx, uint32_t thresh) {
  // keep iff hash >= thresh, thresh = p * 2^32
  const uint32_t h = mix32(idx * 0x9E3779B9U + mix32(seed ^ (stream * 0x85EBCA6BU)));
  return h >= thresh;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// lane id recomputed from the hardware (never hoisted, never kept live): persistent kernels use it instead of
// holding threadIdx-derived registers across their whole tile loop
__device__ __forceinline__ int lane_id_volatile() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

#define VAULT_OK 0
#define VAULT_EINVAL 22
