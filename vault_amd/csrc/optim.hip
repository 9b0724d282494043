// Fused HuggingFace-4.48 AdamW over one flat parameter buffer (+ bf16 shadow weights for the GEMMs).
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= step_size * m / (sqrt(v) + eps) ; p -= lr*wd*p
//   step_size = lr (correct_bias=False, the reference default: ref vault/tmsc_utils/trainer.py:69,244-254)
//               or lr*sqrt(1-b2^t)/(1-b1^t), computed on the host.
// One pass: 16 B/param read (p,g,m,v), 12 B written (p,m,v) + 2 B bf16 shadow (+4 B when zeroing g; nothing written for elements
// with g = m = v = 0 and no weight decay: round 6; `zero_mask`, one byte
// per 64 elements, lets the caller skip the zeroing of ranges the next backward STORES into - the un-split weight-gradient
// tiles of the fused train step - instead of accumulating onto).
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, h16* __restrict__ pb, long long n4,
                                                    float step_size, float lr_wd, float b1, float b2, float eps,
                                                    float gscale, int zero_grad, const uint8_t* __restrict__ zmask) {
  H16_SATURATE();
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256ll) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gv = reinterpret_cast<f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    // Elements that never received a gradient (g = m = v = 0: the rows of an embedding table no token has named yet - a fifth
    // of the parameters is the 64,001-row word table, of which a step touches at most B x T rows) stay exactly as they are
    // without weight decay: m' = v' = 0, p' = p - step * 0 / (0 + eps) = p.  Their five stores (p, m, v, the 16-bit shadow, the
    // gradient's zero) are skipped: 18 of the 34 B per parameter, bit-identical results.
    if (lr_wd == 0.f) {
      bool idle = true;
#pragma unroll
      for (int e = 0; e < 4; ++e) idle = idle && gv[e] == 0.f && mv[e] == 0.f && vv[e] == 0.f;
      if (idle) continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = gv[e] * gscale;
      mv[e] = mv[e] * b1 + (1.f - b1) * ge;
      vv[e] = vv[e] * b2 + (1.f - b2) * ge * ge;
      pv[e] = pv[e] - step_size * (mv[e] / (sqrtf(vv[e]) + eps));
      if (lr_wd != 0.f) pv[e] = pv[e] - lr_wd * pv[e];
    }
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (zero_grad && (zmask == nullptr || zmask[i >> 4])) reinterpret_cast<f32x4*>(g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (pb) {
      uint2 w = {pack_h16x2(pv[0], pv[1]), pack_h16x2(pv[2], pv[3])};
      reinterpret_cast<uint2*>(pb)[i] = w;
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, h16* __restrict__ y, long long n4) {
  H16_SATURATE();
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256ll) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[i];
    uint2 w = {pack_h16x2(a[0], a[1]), pack_h16x2(a[2], a[3])};
    reinterpret_cast<uint2*>(y)[i] = w;
  }
}

// Debug census of a 16-bit tensor in the library's operand format (VAULT_H16_CENSUS): what its narrow exponent did to it.
// out[0] += elements at the largest finite magnitude (what a saturating conversion leaves: H16_SATURATE), out[1] += infinities
// and NaNs, out[2] += subnormals (gradual underflow: significant bits already lost), out[3] += zeros.
__global__ __launch_bounds__(256) void h16_census_kernel(const uint16_t* __restrict__ x, long long n,
                                                         unsigned long long* __restrict__ out) {
#ifdef VAULT_F16
  constexpr uint16_t EXP = 0x7C00, MAN = 0x03FF, MAXF = 0x7BFF;
#else
  constexpr uint16_t EXP = 0x7F80, MAN = 0x007F, MAXF = 0x7F7F;
#endif
  unsigned c[4] = {0u, 0u, 0u, 0u};
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256ll) {
    const uint16_t a = x[i] & 0x7FFF;
    c[0] += a == MAXF;
    c[1] += (a & EXP) == EXP;
    c[2] += (a & EXP) == 0 && (a & MAN) != 0;
    c[3] += a == 0;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out + k, (unsigned long long)v);
  }
}

// out[r][3K]: layout 0 (activation/A operand) = [hi | lo | hi], layout 1 (weight/B operand) = [hi | hi | lo]
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, h16* __restrict__ out, long long rows,
                                                     int K, int layout) {
  H16_SATURATE();
  const long long total = rows * (long long)(K / 4);
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256ll) {
    const long long r = i / (K / 4);
    const int c = (int)(i - r * (K / 4)) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * K + c);
    h16 hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_bf16(a[e], hi[e], lo[e]);
    h16* o = out + r * 3 * K + c;
    const uint2 wh = {pack_h16x2((float)hi[0], (float)hi[1]), pack_h16x2((float)hi[2], (float)hi[3])};
    const uint2 wl = {pack_h16x2((float)lo[0], (float)lo[1]), pack_h16x2((float)lo[2], (float)lo[3])};
    *reinterpret_cast<uint2*>(o) = wh;
    *reinterpret_cast<uint2*>(o + K) = layout == 0 ? wl : wh;
    *reinterpret_cast<uint2*>(o + 2 * K) = layout == 0 ? wh : wl;
  }
}

}  // namespace

extern "C" int vault_split3_bf16(const float* x, void* out_bf16, long long rows, int K, int layout, void* stream) {
  if (!x || !out_bf16 || rows <= 0 || K <= 0 || (K & 3)) return VAULT_EINVAL;
  const long long total = rows * (K / 4);
  const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<h16*>(out_bf16), rows, K, layout);
  return (int)hipGetLastError();
}

extern "C" int vault_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, long long n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, float bias_corr_factor,
                                float grad_scale, int zero_grad, const unsigned char* zero_mask, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || (n & 3) || (zero_mask && (n & 63))) return VAULT_EINVAL;
  const long long n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v,
                     reinterpret_cast<h16*>(p_bf16), n4, lr * bias_corr_factor, lr * weight_decay, beta1, beta2, eps,
                     grad_scale, zero_grad, zero_mask);
  return (int)hipGetLastError();
}

extern "C" int vault_h16_census(const void* x_h16, long long n, unsigned long long* out4, void* stream) {
  if (!x_h16 || !out4 || n <= 0) return VAULT_EINVAL;
  const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(h16_census_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint16_t*>(x_h16), n, out4);
  return (int)hipGetLastError();
}

extern "C" int vault_cast_bf16(const float* x, void* y_bf16, long long n, void* stream) {
  if (!x || !y_bf16 || n <= 0 || (n & 3)) return VAULT_EINVAL;
  const long long n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<h16*>(y_bf16), n4);
  return (int)hipGetLastError();
}

// ---- transposed bf16 weight shadow: dst[b][c][r] = src[b][r][c] for `batch` matrices of rows x cols at uniform strides
// (the layers of a stack in the flat bf16 parameter buffer).  The data-gradient GEMMs dX = dY . W then read W^T as a
// forward-form operand ([N = in][K = out], K contiguous): the register-direct GEMM (gemm8w.hip) takes no k-strided
// weights.  64 x 64 tiles through LDS (padded rows), 16-byte global accesses on both sides.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const h16* __restrict__ src, h16* __restrict__ dst, int rows,
                                                             int cols, long long stride_src, long long stride_dst) {
  __shared__ h16 tile[64][72];
  const int tiles_c = cols >> 6;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x - tr * tiles_c;
  const h16* s = src + (size_t)blockIdx.y * stride_src + (size_t)(tr * 64) * cols + tc * 64;
  h16* d = dst + (size_t)blockIdx.y * stride_dst + (size_t)(tc * 64) * rows + tr * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = (t >> 3) + 32 * k, c8 = (t & 7) * 8;
    const h16x8 v = *reinterpret_cast<const h16x8*>(s + (size_t)r * cols + c8);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r][c8 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = (t >> 3) + 32 * k, r8 = (t & 7) * 8;   // output row c (a source column), 8 consecutive source rows
    h16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[r8 + e][c];
    *reinterpret_cast<h16x8*>(d + (size_t)c * rows + r8) = v;
  }
}

extern "C" int vault_transpose_bf16(const void* src, void* dst, int rows, int cols, int batch, long long stride_src,
                                    long long stride_dst, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || (rows & 63) || (cols & 63) || batch <= 0 || batch > 65535 ||
      (stride_src & 7) || (stride_dst & 7))
    return VAULT_EINVAL;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((rows >> 6) * (cols >> 6), batch), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const h16*>(src),
                     reinterpret_cast<h16*>(dst), rows, cols, stride_src, stride_dst);
  return (int)hipGetLastError();
}
