// Image half of the input pipeline on the GPU: what HuggingFace's ViLT image processor does per item on
// the CPU in the reference (ref: vault/models/vault/dataset.py:337-341 -> HF:models/vilt/image_processing_pil_vilt.py:
// resize shorter side -> 384 / longer <= 640 / multiples of 32 with PIL's antialiased bicubic filter, rescale 1/255,
// normalise (x - 0.5) / 0.5, pad to the batch maximum, pixel_mask) for a whole batch of differently sized uint8 images in
// two launches.  Bit-exact with Pillow's 8-bit resampling: 22-bit fixed-point taps (computed by the host exactly as
// Pillow's precompute_coeffs does, in double precision), horizontal pass into a uint8 intermediate, vertical pass,
// rounding + clipping to 0..255 after each; the float32 value of every 8-bit level comes from a 3 x 256 table the host
// fills with HF's arithmetic.  Integer / byte work, HBM-bound: the horizontal pass stages whole rows through LDS (coalesced
// dword traffic on both sides, byte arithmetic on LDS), the vertical pass handles four output pixels per thread (aligned dword
// loads of the padded intermediate rows, 16-byte stores).
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

constexpr int PREC = 22;

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Horizontal pass, one block per source row: the row (w_in x 3 bytes, any alignment) comes into LDS by coalesced dword loads,
// every thread resamples output pixels from LDS bytes, and the finished row (w_out x 3 bytes, rows of the intermediate are
// padded to 4-byte multiples) leaves LDS by coalesced dword stores.
// tmp[b][y][x][c] = clip8((2^21 + sum_k src[b][y][x0 + k][c] * kk[x][k]) >> 22)   for y < h_in, x < w_out
constexpr int H_IN_MAX = 16384;      // bytes of a source row in LDS (w_in <= 5461); wider images: resize_h_wide_kernel
constexpr int H_OUT_MAX = 2048;      // bytes of an intermediate row (w_out <= 682; the processor caps the longer side at 639)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, long long src_bytes,
                                                       uint8_t* __restrict__ tmp, const int* __restrict__ plan,
                                                       const vault_image_desc* __restrict__ desc) {
  __shared__ __attribute__((aligned(16))) uint32_t row_in[H_IN_MAX / 4 + 1];
  __shared__ __attribute__((aligned(16))) uint32_t row_out[H_OUT_MAX / 4];
  const vault_image_desc d = desc[blockIdx.y];
  const int y = blockIdx.x;
  if (y >= d.h_in) return;
  // the launcher chose this kernel from the caller's max_w_in / max_w_out: a descriptor beyond them would overrun the LDS rows
  if (d.w_in * 3 + 3 > H_IN_MAX || d.w_out * 3 + 3 > H_OUT_MAX) return;
  const long long start = d.src_off + (long long)y * d.w_in * 3;
  const long long abase = start & ~3ll;
  const int mis = (int)(start - abase);
  const int nd = (d.w_in * 3 + mis + 3) >> 2;
  for (int i = threadIdx.x; i < nd; i += 256) {
    const long long o = abase + 4ll * i;
    uint32_t w;
    if (o + 4 <= src_bytes) {
      w = *reinterpret_cast<const uint32_t*>(src + o);
    } else {   // the last dword of the last row may reach past the buffer: byte by byte
      w = 0u;
      for (int k = 0; k < 4; ++k)
        if (o + k < src_bytes) w |= (uint32_t)src[o + k] << (8 * k);
    }
    row_in[i] = w;
  }
  __syncthreads();
  const uint8_t* rin = reinterpret_cast<const uint8_t*>(row_in) + mis;
  uint8_t* rout = reinterpret_cast<uint8_t*>(row_out);
  for (int x = threadIdx.x; x < d.w_out; x += 256) {
    const int x0 = plan[d.hb_off + 2 * x], n = plan[d.hb_off + 2 * x + 1];
    const int* __restrict__ kk = plan + d.hk_off + (size_t)x * d.ksize_h;
    const uint8_t* p = rin + x0 * 3;
    int a0 = 1 << (PREC - 1), a1 = a0, a2 = a0;
    for (int k = 0; k < n; ++k) {
      const int w = kk[k];
      a0 += (int)p[3 * k] * w; a1 += (int)p[3 * k + 1] * w; a2 += (int)p[3 * k + 2] * w;
    }
    rout[3 * x] = (uint8_t)clip8(a0 >> PREC); rout[3 * x + 1] = (uint8_t)clip8(a1 >> PREC); rout[3 * x + 2] = (uint8_t)clip8(a2 >> PREC);
  }
  __syncthreads();
  const int stride_t = (d.w_out * 3 + 3) & ~3;
  uint32_t* q = reinterpret_cast<uint32_t*>(tmp + d.tmp_off + (size_t)y * stride_t);
  for (int i = threadIdx.x; i < (stride_t >> 2); i += 256) q[i] = row_out[i];
}

// the same for rows that do not fit the LDS buffers: one thread per output pixel, bytes from global memory
__global__ __launch_bounds__(256) void resize_h_wide_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
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
  const int stride_t = (d.w_out * 3 + 3) & ~3;
  uint8_t* q = tmp + d.tmp_off + (size_t)y * stride_t + (size_t)x * 3;
  q[0] = (uint8_t)clip8(a0 >> PREC); q[1] = (uint8_t)clip8(a1 >> PREC); q[2] = (uint8_t)clip8(a2 >> PREC);
}

// Vertical pass + normalise + pad, FOUR output pixels per thread (output widths are multiples of 4: the processor floors them
// to multiples of size_divisor): 12 consecutive bytes = three aligned dwords per tap row, three float4 stores, two 16-byte
// mask stores.
// out[b][c][y][x] = lut[c][clip8((2^21 + sum_k tmp[b][y0 + k][x][c] * kk[y][k]) >> 22)] inside the image, 0 in the padding
__global__ __launch_bounds__(256) void resize_v_norm_pad_kernel(const uint8_t* __restrict__ tmp, const int* __restrict__ plan,
                                                                const vault_image_desc* __restrict__ desc,
                                                                const float* __restrict__ lut, float* __restrict__ out,
                                                                long long* __restrict__ mask, float* __restrict__ mask_f32,
                                                                int H, int W, h16* __restrict__ unfold, int ps) {
  H16_SATURATE();
  __shared__ float slut[3 * 256];
  for (int i = threadIdx.x; i < 3 * 256; i += 256) slut[i] = lut[i];
  __syncthreads();
  const int b = blockIdx.z;
  const vault_image_desc d = desc[b];
  const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
  if (x >= W) return;
  const size_t plane = (size_t)H * W, o = (size_t)b * 3 * plane + (size_t)y * W + x;
  const bool inside = y < d.h_out && x < d.w_out;          // (w_out % 4 == 0: a group of four never straddles the edge)
  f32x4 v[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (inside) {
    const int y0 = plan[d.vb_off + 2 * y], n = plan[d.vb_off + 2 * y + 1];
    const int* __restrict__ kk = plan + d.vk_off + (size_t)y * d.ksize_v;
    const int stride_t = (d.w_out * 3 + 3) & ~3;
    const uint8_t* __restrict__ p = tmp + d.tmp_off + (size_t)y0 * stride_t + (size_t)x * 3;
    int acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 1 << (PREC - 1);
    for (int k = 0; k < n; ++k) {
      const int w = kk[k];
      const uint32_t* r = reinterpret_cast<const uint32_t*>(p + (size_t)k * stride_t);
      const uint32_t w0 = r[0], w1 = r[1], w2 = r[2];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[e] += (int)((w0 >> (8 * e)) & 255u) * w;
        acc[4 + e] += (int)((w1 >> (8 * e)) & 255u) * w;
        acc[8 + e] += (int)((w2 >> (8 * e)) & 255u) * w;
      }
    }
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c][px] = slut[256 * c + clip8(acc[3 * px + c] >> PREC)];
  }
  if (out) {
    *reinterpret_cast<f32x4*>(out + o) = v[0];
    *reinterpret_cast<f32x4*>(out + o + plane) = v[1];
    *reinterpret_cast<f32x4*>(out + o + 2 * plane) = v[2];
  }
  if (unfold) {   // four consecutive x of one patch row: four consecutive k of the unfold row, per channel
    const int gw = W / ps, py = y / ps, px = x / ps;
    const size_t row = ((size_t)b * (H / ps) + py) * gw + px;
    h16* u = unfold + row * (size_t)(3 * ps * ps) + (size_t)(y - py * ps) * ps + (x - px * ps);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *reinterpret_cast<uint2*>(u + (size_t)c * ps * ps) = uint2{pack_h16x2(v[c][0], v[c][1]), pack_h16x2(v[c][2], v[c][3])};
  }
  const size_t mo = (size_t)b * plane + (size_t)y * W + x;
  if (mask) {
    const long long m = inside ? 1 : 0;
    long long* q = mask + mo;
    q[0] = m; q[1] = m; q[2] = m; q[3] = m;
  }
  if (mask_f32) *reinterpret_cast<f32x4*>(mask_f32 + mo) = inside ? f32x4{1.f, 1.f, 1.f, 1.f} : f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace

extern "C" int vault_image_preprocess(const vault_preprocess_args* a, void* stream) {
  if (!a || !a->src || !a->tmp || !a->plan || !a->desc || !a->lut || (!a->pixel_values && !a->patch_unfold_bf16) ||
      (a->patch_unfold_bf16 && (a->ps <= 0 || (a->ps & 3) || a->H % a->ps || a->W % a->ps)) || a->B <= 0 || a->H <= 0 || a->W <= 0 ||
      a->max_h_in <= 0 || a->max_w_out <= 0 || a->max_h_in > 65535 || a->H > 65535 || a->B > 65535 || a->max_w_out > a->W ||
      (a->W & 3) || a->src_bytes <= 0 || a->max_w_in <= 0)
    return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->max_w_in * 3 + 3 <= H_IN_MAX && a->max_w_out * 3 + 3 <= H_OUT_MAX)
    hipLaunchKernelGGL(resize_h_kernel, dim3(a->max_h_in, a->B), dim3(256), 0, st, a->src, a->src_bytes, a->tmp, a->plan, a->desc);
  else
    hipLaunchKernelGGL(resize_h_wide_kernel, dim3((a->max_w_out + 255) / 256, a->max_h_in, a->B), dim3(256), 0, st, a->src, a->tmp,
                       a->plan, a->desc);
  hipLaunchKernelGGL(resize_v_norm_pad_kernel, dim3((a->W / 4 + 255) / 256, a->H, a->B), dim3(256), 0, st, a->tmp, a->plan, a->desc,
                     a->lut, a->pixel_values, reinterpret_cast<long long*>(a->pixel_mask), a->pixel_mask_f32, a->H, a->W,
                     reinterpret_cast<h16*>(a->patch_unfold_bf16), a->ps);
  return (int)hipGetLastError();
}
