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

// the writes of one group of four output pixels (x % 4 == 0) of row y of image b: f32 NCHW canvas and / or the 16-bit patch
// unfold, pixel mask as int64 and / or f32 - shared by the two-pass and the fused kernels
__device__ __forceinline__ void emit_pixels(int b, int y, int x, bool inside, const f32x4 (&v)[3], float* __restrict__ out,
                                            long long* __restrict__ mask, float* __restrict__ mask_f32, int H, int W,
                                            h16* __restrict__ unfold, int ps) {
  const size_t plane = (size_t)H * W, o = (size_t)b * 3 * plane + (size_t)y * W + x;
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
  emit_pixels(b, y, x, inside, v, out, mask, mask_f32, H, W, unfold, ps);
}

// ---- The fused form: both passes of one band of FB_ROWS output rows of one image inside ONE workgroup, the 8-bit intermediate
// in LDS (never in HBM).  Chosen by the launcher when the batch's widest band fits (fused_lds_bytes); HBM traffic = the source
// bytes once (+ the rows two neighbouring bands share: ksize_v - 1 of ~FB_ROWS * scale) and the outputs once.
//   stage : FB_G source rows (contiguous bytes of the packed image) staged by 16-byte loads, the next group's loads in flight
//           (registers) while this group is resampled
//   inter : the horizontally resampled rows of the band, w_out x 3 bytes each (rows padded to dwords) - what the two-pass form
//           keeps in `tmp`
// Horizontal pass: an item = one output x of TWO staged rows; its (<= 8 per chunk) tap weights come from an LDS copy of the
// plan, its 24 source bytes per row as 7 aligned dwords funnel-shifted (v_alignbyte) to the pixel's byte offset - byte reads
// would issue 3.5 x the LDS instructions.  Vertical pass: four output pixels of one row per item, as the two-pass kernel;
// item order = 4-pixel group inside a patch row fastest, then the row: a wave writes 8 rows x 64 bytes of one patch and channel.
// byte x tap on the full-rate 24-bit multiplier (a 32-bit v_mul_lo_u32 issues at quarter rate): taps are 22-bit fixed point
// weights of magnitude < 2 (the host plan checks |tap| < 2^23 before it offers the fused form)
__device__ __forceinline__ int mac24(int acc, uint32_t byte, int w) { return acc + __mul24((int)byte, w); }

#ifndef FB_ABL
#define FB_ABL 0          // development: 1 no horizontal arithmetic, 2 no vertical arithmetic, 4 no output writes, 8 no source loads
#endif
// KC taps x 3 channels of TWO staged rows (rowb bytes apart) for one output x: 3 KC source bytes per row, read as aligned
// dwords and funnel-shifted to the pixel's byte offset; `kvalid` <= KC taps have weights (the rest multiply by zero)
template <int KC>
__device__ __forceinline__ void hchunk(const uint8_t* sp, int rowb, const int* tp, int kvalid, int (&a)[2][3]) {
  constexpr int ND = (3 * KC + 3) / 4;
  int wt[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) wt[k] = (k < kvalid) ? tp[k] : 0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint8_t* s = sp + r * rowb;
    const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(s) & 3u;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(s - sh);
    uint32_t w[ND + 1], dd[ND];
#pragma unroll
    for (int j = 0; j < ND + 1; ++j) w[j] = q[j];
#pragma unroll
    for (int j = 0; j < ND; ++j) dd[j] = __builtin_amdgcn_alignbyte(w[j + 1], w[j], sh);
#pragma unroll
    for (int k = 0; k < KC; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int e = 3 * k + c;
        a[r][c] = mac24(a[r][c], (dd[e >> 2] >> (8 * (e & 3))) & 255u, wt[k]);
      }
  }
}

constexpr int FB_ROWS = 32, FB_G = 8, FB_THREADS = 512, FB_PF = 4;   // FB_PF 16-byte prefetch registers per thread and group

struct FusedLds { int taps_h, bounds_h, taps_v, bounds_v, stage, inter, total; };
// (all offsets in bytes, 16-byte aligned; the float table of the 8-bit levels overlays `stage` after the horizontal pass)
__host__ __device__ inline FusedLds fused_lds_layout(int max_w_in, int max_w_out, int ksize_max, int band_rows_max) {
  auto up = [](int v) { return (v + 15) & ~15; };
  FusedLds l;
  l.taps_h = 0;
  l.bounds_h = l.taps_h + up(4 * max_w_out * ksize_max);
  l.taps_v = l.bounds_h + up(4 * max_w_out);
  l.bounds_v = l.taps_v + up(4 * FB_ROWS * ksize_max);
  l.stage = l.bounds_v + up(4 * FB_ROWS);
  int stage_bytes = FB_G * max_w_in * 3 + 16 + 3 * ksize_max + 64;      // + misalignment + reads of zero-weight taps past the end
  if (stage_bytes < 3 * 256 * 4) stage_bytes = 3 * 256 * 4;
  l.inter = l.stage + up(stage_bytes);
  l.total = l.inter + up(band_rows_max * ((max_w_out * 3 + 3) & ~3));
  return l;
}

__global__ __launch_bounds__(FB_THREADS) void resize_fused_kernel(const uint8_t* __restrict__ src, long long src_bytes,
                                                                  const int* __restrict__ plan,
                                                                  const vault_image_desc* __restrict__ desc,
                                                                  const float* __restrict__ lut, float* __restrict__ out,
                                                                  long long* __restrict__ mask, float* __restrict__ mask_f32,
                                                                  int H, int W, h16* __restrict__ unfold, int ps, int max_w_in,
                                                                  int max_w_out, int ksize_max, int band_rows_max) {
  H16_SATURATE();
  extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
  const FusedLds L = fused_lds_layout(max_w_in, max_w_out, ksize_max, band_rows_max);
  int* const htap = reinterpret_cast<int*>(fsm + L.taps_h);
  int* const hbnd = reinterpret_cast<int*>(fsm + L.bounds_h);        // x0 | n << 16
  int* const vtap = reinterpret_cast<int*>(fsm + L.taps_v);
  int* const vbnd = reinterpret_cast<int*>(fsm + L.bounds_v);        // (y0 - first source row of the band) | n << 16
  uint8_t* const stage = fsm + L.stage;
  uint8_t* const inter = fsm + L.inter;
  const float* const slut = reinterpret_cast<const float*>(stage);
  const int t = threadIdx.x, b = blockIdx.y, y_lo = blockIdx.x * FB_ROWS;
  const vault_image_desc d = desc[b];
  int rows_here = d.h_out - y_lo;
  if (rows_here > FB_ROWS) rows_here = FB_ROWS;
  const int stride_t = (d.w_out * 3 + 3) & ~3;
  // descriptors beyond what the launcher sized the LDS for (its caller's maxima): the band is written as padding
  bool ok = rows_here > 0 && d.w_in <= max_w_in && d.w_out <= max_w_out && d.ksize_h <= ksize_max && d.ksize_v <= ksize_max;
  int ys = 0, nrows = 0;
  if (ok) {
    const int last = y_lo + rows_here - 1;
    ys = plan[d.vb_off + 2 * y_lo];
    nrows = plan[d.vb_off + 2 * last] + plan[d.vb_off + 2 * last + 1] - ys;      // (Pillow's bounds are monotone in y)
    ok = nrows > 0 && nrows <= band_rows_max && ys + nrows <= d.h_in;
  }
  if (ok) {
    const int ksh = d.ksize_h, ksv = d.ksize_v, rowb = d.w_in * 3;
    // ---- horizontal pass over the band's source rows, FB_G at a time
    const int ngroups = (nrows + FB_G - 1) / FB_G;
    uint4 pf[FB_PF];
    int mis = 0, mis_next = 0;
    auto prefetch = [&](int g) {
      int nr = nrows - g * FB_G; if (nr > FB_G) nr = FB_G;
      const long long start = d.src_off + (long long)(ys + g * FB_G) * rowb;
      const unsigned long long addr = reinterpret_cast<unsigned long long>(src) + (unsigned long long)start;
      mis_next = (int)(addr & 15ull);
      const long long o0 = start - mis_next;                      // offset of the first 16-byte unit relative to src (>= -15)
      const int n16 = (nr * rowb + mis_next + 15) >> 4;
#pragma unroll
      for (int j = 0; j < FB_PF; ++j) {
        const int i = t + j * FB_THREADS;
        if (i < n16) {
          const long long o = o0 + 16ll * i;
          if (o >= 0 && o + 16 <= src_bytes) {
            pf[j] = *reinterpret_cast<const uint4*>(src + o);
          } else {                                                // the first / last unit may reach outside the buffer
            unsigned long long lo = 0ull, hi = 0ull;
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {
              if (o + k >= 0 && o + k < src_bytes) lo |= (unsigned long long)src[o + k] << (8 * k);
              if (o + 8 + k >= 0 && o + 8 + k < src_bytes) hi |= (unsigned long long)src[o + 8 + k] << (8 * k);
            }
            pf[j] = uint4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
          }
        }
      }
    };
    if (!(FB_ABL & 8)) prefetch(0); else { for (int j = 0; j < FB_PF; ++j) pf[j] = uint4{1u, 2u, 3u, 4u}; }
    // (the first group's source loads are in flight while the plan comes in)
    for (int i = t; i < d.w_out * ksh; i += FB_THREADS) htap[i] = plan[d.hk_off + i];
    for (int i = t; i < d.w_out; i += FB_THREADS) hbnd[i] = plan[d.hb_off + 2 * i] | (plan[d.hb_off + 2 * i + 1] << 16);
    for (int i = t; i < rows_here * ksv; i += FB_THREADS) vtap[i] = plan[d.vk_off + (size_t)y_lo * ksv + i];
    if (t < rows_here) vbnd[t] = (plan[d.vb_off + 2 * (y_lo + t)] - ys) | (plan[d.vb_off + 2 * (y_lo + t) + 1] << 16);
    const int x_first = t % d.w_out, r_first = 2 * (t / d.w_out), x_step = FB_THREADS % d.w_out, r_step = 2 * (FB_THREADS / d.w_out);
    for (int g = 0; g < ngroups; ++g) {
      int nr = nrows - g * FB_G; if (nr > FB_G) nr = FB_G;
      __syncthreads();                                            // the previous group's readers are done (and the plan copies)
      mis = mis_next;
      {
        const int n16 = (nr * rowb + mis + 15) >> 4;
#pragma unroll
        for (int j = 0; j < FB_PF; ++j) {
          const int i = t + j * FB_THREADS;
          if (i < n16) reinterpret_cast<uint4*>(stage)[i] = pf[j];
        }
      }
      __syncthreads();
      if (g + 1 < ngroups && !(FB_ABL & 8)) prefetch(g + 1);
      if (FB_ABL & 1) continue;
      // items (x, row pair) = t, t + FB_THREADS, ..: walked without a division per item
      for (int x = x_first, r0 = r_first; r0 < nr; x += x_step, r0 += r_step) {
        if (x >= d.w_out) { x -= d.w_out; r0 += 2; if (r0 >= nr) break; }
        const int hb = hbnd[x], x0 = hb & 0xffff;
        const int s0 = mis + __mul24(r0, rowb) + 3 * x0, tap0 = __mul24(x, ksh);
        int a[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) a[r][c] = 1 << (PREC - 1);
        for (int k0 = 0; k0 < ksh; k0 += 8) {
          const int kc = ksh - k0;                                 // (uniform: the compiled-out taps of a 5 / 6 / 7-tap filter
          const uint8_t* sp = stage + s0 + 3 * k0;                 //  are a fifth to a third of the arithmetic)
          const int* tp = htap + tap0 + k0;
          if (kc >= 8) hchunk<8>(sp, rowb, tp, 8, a);
          else if (kc == 7) hchunk<7>(sp, rowb, tp, 7, a);
          else if (kc == 6) hchunk<6>(sp, rowb, tp, 6, a);
          else if (kc == 5) hchunk<5>(sp, rowb, tp, 5, a);
          else if (kc == 4) hchunk<4>(sp, rowb, tp, 4, a);
          else hchunk<4>(sp, rowb, tp, kc, a);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
          if (r0 + r < nr) {
            uint8_t* q = inter + __mul24(g * FB_G + r0 + r, stride_t) + 3 * x;
            q[0] = (uint8_t)clip8(a[r][0] >> PREC); q[1] = (uint8_t)clip8(a[r][1] >> PREC); q[2] = (uint8_t)clip8(a[r][2] >> PREC);
          }
      }
    }
    __syncthreads();                                              // inter complete; stage free for the table of levels
    for (int i = t; i < 3 * 256; i += FB_THREADS) reinterpret_cast<float*>(stage)[i] = lut[i];
  }
  __syncthreads();
  // ---- vertical pass + normalise + pad + the writes, the band's FB_ROWS x W canvas pixels
  const int cw = unfold ? ps : 32, gpc = cw >> 2;                 // 4-pixel groups per cell row (cell = patch)
  const int cells = (W + cw - 1) / cw;
  const int band_rows = (H - y_lo) < FB_ROWS ? (H - y_lo) : FB_ROWS;
  const int items = cells * FB_ROWS * gpc;
  const bool gpc_pow2 = (gpc & (gpc - 1)) == 0;
  const int gpc_sh = 31 - __clz(gpc);
  for (int i = t; i < items; i += FB_THREADS) {
    const int q1 = gpc_pow2 ? (i >> gpc_sh) : (i / gpc);
    const int xg = i - q1 * gpc, r = q1 % FB_ROWS, pc = q1 / FB_ROWS;
    const int x = pc * cw + 4 * xg, y = y_lo + r;
    if (r >= band_rows || x >= W) continue;
    const bool inside = ok && r < rows_here && x < d.w_out && !(FB_ABL & 2);
    f32x4 v[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (inside) {
      const int vb = vbnd[r], y0 = vb & 0xffff, n = vb >> 16;
      const int* kk = vtap + __mul24(r, d.ksize_v);
      const uint8_t* p = inter + __mul24(y0, stride_t) + x * 3;
      int acc[12];
#pragma unroll
      for (int e = 0; e < 12; ++e) acc[e] = 1 << (PREC - 1);
      for (int k = 0; k < n; ++k) {
        const int w = kk[k];
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p + __mul24(k, stride_t));
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[e] = mac24(acc[e], (w0 >> (8 * e)) & 255u, w);
          acc[4 + e] = mac24(acc[4 + e], (w1 >> (8 * e)) & 255u, w);
          acc[8 + e] = mac24(acc[8 + e], (w2 >> (8 * e)) & 255u, w);
        }
      }
#pragma unroll
      for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c][px] = slut[256 * c + clip8(acc[3 * px + c] >> PREC)];
    }
    if (!(FB_ABL & 4) || v[0][0] == 123.f) emit_pixels(b, y, x, inside, v, out, mask, mask_f32, H, W, unfold, ps);
  }
}

}  // namespace

namespace {
constexpr int FUSED_LDS_MAX = 160 * 1024;
// the fused form needs the caller's band / tap maxima (ABI 12 members; zero = unknown: two-pass form), one band in LDS and a
// staged group of rows within the prefetch registers
int fused_lds_bytes(const vault_preprocess_args* a) {
  if (a->ksize_max <= 0 || a->band_rows_max <= 0 || a->max_w_in > 65535 || a->band_rows_max > 65535) return 0;
  if ((long long)FB_G * a->max_w_in * 3 + 31 > (long long)FB_PF * FB_THREADS * 16) return 0;
  const long long inter = (long long)a->band_rows_max * ((a->max_w_out * 3 + 3) & ~3);
  const long long taps = 4ll * a->max_w_out * a->ksize_max;
  if (inter > FUSED_LDS_MAX || taps > FUSED_LDS_MAX) return 0;
  const FusedLds l = fused_lds_layout(a->max_w_in, a->max_w_out, a->ksize_max, a->band_rows_max);
  return l.total <= FUSED_LDS_MAX ? l.total : 0;
}
}  // namespace

extern "C" int vault_image_preprocess_is_fused(const vault_preprocess_args* a) { return a && fused_lds_bytes(a) > 0 ? 1 : 0; }

extern "C" int vault_image_preprocess(const vault_preprocess_args* a, void* stream) {
  if (!a || !a->src || !a->plan || !a->desc || !a->lut || (!a->pixel_values && !a->patch_unfold_bf16) ||
      (a->patch_unfold_bf16 && (a->ps <= 0 || (a->ps & 3) || a->H % a->ps || a->W % a->ps)) || a->B <= 0 || a->H <= 0 || a->W <= 0 ||
      a->max_h_in <= 0 || a->max_w_out <= 0 || a->max_h_in > 65535 || a->H > 65535 || a->B > 65535 || a->max_w_out > a->W ||
      (a->W & 3) || a->src_bytes <= 0 || a->max_w_in <= 0)
    return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int lds = fused_lds_bytes(a);
  if (lds > 0) {
    static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev);
    bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(resize_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         FUSED_LDS_MAX);
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(resize_fused_kernel, dim3((a->H + FB_ROWS - 1) / FB_ROWS, a->B), dim3(FB_THREADS), lds, st, a->src, a->src_bytes,
                       a->plan, a->desc, a->lut, a->pixel_values, reinterpret_cast<long long*>(a->pixel_mask), a->pixel_mask_f32, a->H,
                       a->W, reinterpret_cast<h16*>(a->patch_unfold_bf16), a->ps, a->max_w_in, a->max_w_out, a->ksize_max,
                       a->band_rows_max);
    return (int)hipGetLastError();
  }
  if (!a->tmp) return VAULT_EINVAL;
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
