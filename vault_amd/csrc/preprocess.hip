// Image half of the input pipeline on the GPU: what HuggingFace's ViLT image processor does per item on
// the CPU in the reference (ref: vault/models/vault/dataset.py:337-341 -> HF:models/vilt/image_processing_pil_vilt.py:
// resize shorter side -> 384 / longer <= 640 / multiples of 32 with PIL's antialiased bicubic filter, rescale 1/255,
// normalise (x - 0.5) / 0.5, pad to the batch maximum, pixel_mask) for a whole batch of differently sized uint8 images in
// two launches.  Bit-exact with Pillow's 8-bit resampling: 22-bit fixed-point taps (computed by the host exactly as
// Pillow's precompute_coeffs does, in double precision), horizontal pass into a uint8 intermediate, vertical pass,
// rounding + clipping to 0..255 after each; the float32 value of every 8-bit level comes from a 3 x 256 table the host
// fills with HF's arithmetic.  Integer / byte work, HBM-bound: one thread per output pixel (three channels), rows
// contiguous across the wave; the taps of neighbouring outputs overlap, so the source rows are read from HBM once and
// served from L2 / L1 afterwards.
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

constexpr int PREC = 22;

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// tmp[b][y][x][c] = clip8((2^21 + sum_k src[b][y][x0 + k][c] * kk[x][k]) >> 22)   for y < h_in, x < w_out
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                                                       const int* __restrict__ plan,
                                                       const vault_image_desc* __restrict__ desc) {
  const vault_image_desc d = desc[blockIdx.z];
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (y >= d.h_in || x >= d.w_out) return;
  const int x0 = plan[d.hb_off + 2 * x], n = plan[d.hb_off + 2 * x + 1];
  const int* __restrict__ kk = plan + d.hk_off + (size_t)x * d.ksize_h;
  const uint8_t* __restrict__ p = src + d.src_off + ((size_t)y * d.w_in + x0) * 3;
  int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
  for (int k = 0; k < n; ++k) {
    const int w = kk[k];
    a0 += (int)p[3 * k] * w; a1 += (int)p[3 * k + 1] * w; a2 += (int)p[3 * k + 2] * w;
  }
  uint8_t* q = tmp + d.tmp_off + ((size_t)y * d.w_out + x) * 3;
  q[0] = (uint8_t)clip8(a0 >> PREC); q[1] = (uint8_t)clip8(a1 >> PREC); q[2] = (uint8_t)clip8(a2 >> PREC);
}

// out[b][c][y][x] = lut[c][clip8((2^21 + sum_k tmp[b][y0 + k][x][c] * kk[y][k]) >> 22)] inside the image, 0 in the padding
__global__ __launch_bounds__(256) void resize_v_norm_pad_kernel(const uint8_t* __restrict__ tmp, const int* __restrict__ plan,
                                                                const vault_image_desc* __restrict__ desc,
                                                                const float* __restrict__ lut, float* __restrict__ out,
                                                                long long* __restrict__ mask, float* __restrict__ mask_f32,
                                                                int H, int W) {
  __shared__ float slut[3 * 256];
  for (int i = threadIdx.x; i < 3 * 256; i += 256) slut[i] = lut[i];
  __syncthreads();
  const int b = blockIdx.z;
  const vault_image_desc d = desc[b];
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t plane = (size_t)H * W, o = (size_t)b * 3 * plane + (size_t)y * W + x;
  const bool inside = y < d.h_out && x < d.w_out;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f;
  if (inside) {
    const int y0 = plan[d.vb_off + 2 * y], n = plan[d.vb_off + 2 * y + 1];
    const int* __restrict__ kk = plan + d.vk_off + (size_t)y * d.ksize_v;
    const uint8_t* __restrict__ p = tmp + d.tmp_off + ((size_t)y0 * d.w_out + x) * 3;
    const size_t rs = (size_t)d.w_out * 3;
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
    for (int k = 0; k < n; ++k) {
      const int w = kk[k];
      a0 += (int)p[k * rs] * w; a1 += (int)p[k * rs + 1] * w; a2 += (int)p[k * rs + 2] * w;
    }
    v0 = slut[clip8(a0 >> PREC)]; v1 = slut[256 + clip8(a1 >> PREC)]; v2 = slut[512 + clip8(a2 >> PREC)];
  }
  out[o] = v0; out[o + plane] = v1; out[o + 2 * plane] = v2;
  if (mask) mask[(size_t)b * plane + (size_t)y * W + x] = inside ? 1 : 0;
  if (mask_f32) mask_f32[(size_t)b * plane + (size_t)y * W + x] = inside ? 1.f : 0.f;
}

}  // namespace

extern "C" int vault_image_preprocess(const vault_preprocess_args* a, void* stream) {
  if (!a || !a->src || !a->tmp || !a->plan || !a->desc || !a->lut || !a->pixel_values || a->B <= 0 || a->H <= 0 || a->W <= 0 ||
      a->max_h_in <= 0 || a->max_w_out <= 0 || a->max_h_in > 65535 || a->H > 65535 || a->B > 65535 || a->max_w_out > a->W)
    return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(resize_h_kernel, dim3((a->max_w_out + 255) / 256, a->max_h_in, a->B), dim3(256), 0, st, a->src, a->tmp, a->plan,
                     a->desc);
  hipLaunchKernelGGL(resize_v_norm_pad_kernel, dim3((a->W + 255) / 256, a->H, a->B), dim3(256), 0, st, a->tmp, a->plan, a->desc, a->lut,
                     a->pixel_values, reinterpret_cast<long long*>(a->pixel_mask), a->pixel_mask_f32, a->H, a->W);
  return (int)hipGetLastError();
}
