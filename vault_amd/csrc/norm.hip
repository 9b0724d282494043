// LayerNorm forward/backward and column-sum (bias gradient) kernels, fp32 statistics.
// One wave per row, 16-byte accesses; H must be a multiple of 256 (64 lanes x float4).
//
// Replaces nn.LayerNorm at HF:models/vilt/modeling_vilt.py:431-447,637 (pre-LN, eps 1e-12) and
// HF:models/roberta/modeling_roberta.py:339,397 (post-LN, eps 1e-5) and their backward.
#include <cstdlib>
#include <algorithm>
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

struct RowMap {  // logical row r -> physical row (r / rpg) * gstride + goff + r % rpg ; rpg == 0: identity
  int rpg, gstride, goff;
  __device__ __forceinline__ size_t operator()(int r) const {
    if (rpg == 0) return (size_t)r;
    const int gq = r / rpg;
    return (size_t)gq * gstride + goff + (r - gq * rpg);
  }
};

template <int VPT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, RowMap xmap,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int rows, int H,
                                                     h16* __restrict__ y_bf16, float* __restrict__ y_f32, RowMap ymap,
                                                     const float* __restrict__ post_add,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     uint32_t drop_thresh, uint32_t drop_seed, uint32_t drop_stream,
                                                     float drop_scale, h16* __restrict__ y_split3,
                                                     uint8_t* __restrict__ y_q, uint8_t* __restrict__ y_scale) {
  H16_SATURATE();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + xmap(row) * H;
  f32x4 v[VPT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    v[j] = *reinterpret_cast<const f32x4*>(xr + (lane + 64 * j) * 4);
    s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
  }
  const float mu = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[j][e] - mu;
      q += d * d;
    }
  const float var = wave_sum(q) / (float)H;
  const float rs = 1.0f / sqrtf(var + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  const size_t orow = ymap(row);
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int c = (lane + 64 * j) * 4;
    const f32x4 gw = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 bw = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rs * gw[e] + bw[e];
    if (drop_thresh != 0u) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = dropout_keep(drop_seed, drop_stream, (uint32_t)(orow * H + c + e), drop_thresh) ? o[e] * drop_scale : 0.f;
    }
    if (post_add) {
      const f32x4 pa = *reinterpret_cast<const f32x4*>(post_add + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += pa[e];
    }
    if (y_f32) *reinterpret_cast<f32x4*>(y_f32 + orow * H + c) = o;
    if (y_bf16) {
      uint2 w = {pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3])};
      *reinterpret_cast<uint2*>(y_bf16 + orow * H + c) = w;
    }
    if (y_q) {
      // MXFP8 image of the bf16 output (the bytes vault_quant_mxfp8 would produce from y_bf16): a block of 32
      // consecutive columns is held by 8 consecutive lanes
      float b[4];
      float amax = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { b[e] = (float)(h16)o[e]; amax = fmaxf(amax, fabsf(b[e])); }
      amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
      amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
      int e8 = 0;
      if (amax > 0.f) {
        e8 = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xff) - 8;
        e8 = e8 < 0 ? 0 : e8;
      }
      const float inv = __builtin_bit_cast(float, (uint32_t)(254 - e8) << 23);
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = fminf(fmaxf(b[e] * inv, -448.f), 448.f);
      int w8 = 0;
      w8 = __builtin_amdgcn_cvt_pk_fp8_f32(b[0], b[1], w8, false);
      w8 = __builtin_amdgcn_cvt_pk_fp8_f32(b[2], b[3], w8, true);
      *reinterpret_cast<int*>(y_q + orow * H + c) = w8;
      if ((lane & 7) == 0) y_scale[orow * (H / 32) + (c >> 5)] = (uint8_t)e8;
    }
    if (y_split3) {   // [hi | lo | hi] A-operand layout of the split-bf16 (precise) GEMM path
      h16 hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split_bf16(o[e], hi[e], lo[e]);
      const uint2 wh = {pack_h16x2((float)hi[0], (float)hi[1]), pack_h16x2((float)hi[2], (float)hi[3])};
      const uint2 wl = {pack_h16x2((float)lo[0], (float)lo[1]), pack_h16x2((float)lo[2], (float)lo[3])};
      h16* d = y_split3 + orow * 3 * H + c;
      *reinterpret_cast<uint2*>(d) = wh;
      *reinterpret_cast<uint2*>(d + H) = wl;
      *reinterpret_cast<uint2*>(d + 2 * H) = wh;
    }
  }
}

// Cross-wave reduction of the three per-lane column sums (d gamma, d beta, the next Linear's bias gradient) of a block and
// one global float atomic per column and block.
template <int VPT, int WAVES>
__device__ __forceinline__ void ln_bwd_fold(const f32x4 (&ag)[VPT], const f32x4 (&ab)[VPT], const f32x4 (&ac)[VPT], float* red,
                                            float* dgamma, float* dbeta, float* dbias, int lane, int wave) {
  constexpr int NB = WAVES >= 8 ? 8 : 4;      // waves that fold into the scratch per round
  constexpr int HH = VPT * 256;               // = H
  if (dgamma == nullptr && dbias == nullptr) return;
  // cross-wave reduction of the three column sums together through a [NB][3][H] scratch (72 KiB at H = 768) that the
  // waves add into NB at a time (LDS float atomics run ~a lane per clock here: 30 us for this - plain read-add-write
  // rounds instead; one quantity at a time through a [4][H] scratch took 12 rounds + 3 passes of atomics: 4-7 us of
  // every launch at small row counts), then one global atomic per column and block.  The global atomics of all blocks
  // land on the same H addresses at the end of the kernel and serialise in the L2: with 1024 blocks that tail cost
  // 14-22 us per launch whatever the row count (tools/ln_bench.py), hence 16-wave blocks, one per CU (<= 256 blocks)
  float* const dsts[3] = {dgamma, dbeta, dbias};
  for (int r = 0; r < WAVES / NB; ++r) {
    if ((wave / NB) == r) {
      float* rw = red + (wave % NB) * (3 * HH);
#pragma unroll
      for (int qn = 0; qn < 3; ++qn) {
        if (dsts[qn] == nullptr) continue;     // uniform
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
          const f32x4 v = qn == 0 ? ag[j] : (qn == 1 ? ab[j] : ac[j]);
          f32x4* q4 = reinterpret_cast<f32x4*>(rw + qn * HH + (lane + 64 * j) * 4);
          *q4 = (r == 0) ? v : (*q4 + v);
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < 3 * HH; c += WAVES * 64) {
    const int qn = c / HH;
    float* dst = dsts[qn];
    if (dst == nullptr) continue;
    float t = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) t += red[b * (3 * HH) + c];
    atomicAdd(dst + (c - qn * HH), t);
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ; dy = dy_bf16 + dy_f32 (either may be
// null).  Outputs dx_f32 = dx + dres (optional) and a bf16 copy (optionally dropout-masked for the
// branch that sits behind a dropout in forward).  dgamma / dbeta: per-block partials + float atomics.
template <int VPT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 16 / WAVES) void ln_bwd_kernel(const h16* __restrict__ dy_bf16, const float* __restrict__ dy_f32,
                                                     RowMap dymap, const float* __restrict__ x, RowMap xmap,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, int rows, int H,
                                                     const float* __restrict__ dres, float* __restrict__ dx_f32,
                                                     h16* __restrict__ dx_bf16, RowMap dxmap, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dbias,
                                                     int rows_per_block, uint32_t drop_thresh, uint32_t drop_seed, uint32_t drop_stream,
                                                     float drop_scale, int drop_on_dy, const h16* __restrict__ dres_bf16) {
  H16_SATURATE();
  constexpr int NB = WAVES >= 8 ? 8 : 4;      // waves that fold into the scratch per round
  constexpr int HH = VPT * 256;               // = H
  extern __shared__ __attribute__((aligned(16))) float red[];   // [NB][3][HH]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gw[VPT], ag[VPT], ab[VPT], ac[VPT];
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    gw[j] = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * j) * 4);
    ag[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ac[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  for (int row = r0 + wave; row < r1; row += WAVES) {
    const size_t xr = xmap(row) * H, dr = dymap(row) * H;
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[VPT], gy[VPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int c = (lane + 64 * j) * 4;
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + xr + c);
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
      if (dy_f32) d = *reinterpret_cast<const f32x4*>(dy_f32 + dr + c);
      if (dy_bf16) {
        const uint2 w = *reinterpret_cast<const uint2*>(dy_bf16 + dr + c);
        const float2 a = unpack_h16x2(w.x), b = unpack_h16x2(w.y);
        d[0] += a.x; d[1] += a.y; d[2] += b.x; d[3] += b.y;
      }
      if (drop_on_dy && drop_thresh != 0u) {   // y = dropout(LN(x)): mask the incoming gradient
#pragma unroll
        for (int e = 0; e < 4; ++e)
          d[e] = dropout_keep(drop_seed, drop_stream, (uint32_t)(dr + c + e), drop_thresh) ? d[e] * drop_scale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[j][e] = (xv[e] - mu) * rs;
        gy[j][e] = d[e] * gw[j][e];
        s1 += gy[j][e];
        s2 += gy[j][e] * xh[j][e];
        ag[j][e] += d[e] * xh[j][e];
        ab[j][e] += d[e];
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    const size_t orow = dxmap(row) * H;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int c = (lane + 64 * j) * 4;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs * (gy[j][e] - s1 - xh[j][e] * s2);
      if (dres) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(dres + orow + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += rv[e];
      }
      if (dres_bf16) {   // residual-gradient stream kept in bf16 (the same tensor is the next Linear's dY operand)
        const uint2 w = *reinterpret_cast<const uint2*>(dres_bf16 + orow + c);
        const float2 a = unpack_h16x2(w.x), b = unpack_h16x2(w.y);
        o[0] += a.x; o[1] += a.y; o[2] += b.x; o[3] += b.y;
      }
      if (dx_f32) *reinterpret_cast<f32x4*>(dx_f32 + orow + c) = o;
      if (dx_bf16) {
        if (drop_thresh != 0u && !drop_on_dy) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = dropout_keep(drop_seed, drop_stream, (uint32_t)(orow + c + e), drop_thresh) ? o[e] * drop_scale : 0.f;
        }
        uint2 w = {pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3])};
        *reinterpret_cast<uint2*>(dx_bf16 + orow + c) = w;
#pragma unroll
        for (int e = 0; e < 4; ++e) ac[j][e] += o[e];   // column sums of the bf16 branch (a Linear's bias gradient)
      }
    }
  }
  ln_bwd_fold<VPT, WAVES>(ag, ab, ac, red, dgamma, dbeta, dbias, lane, wave);
}

// The same backward for the case that makes up the pre-LN ViLT stack's 24 launches per step - 16-bit incoming gradient
// (dy_bf16), 16-bit residual-gradient stream in (dres_bf16) and out (dx_bf16), identity row maps, no dropout, no f32
// stream - as straight-line code: the general kernel's run-time switches cut its row loop into ~40 basic blocks, which
// keeps hipcc from issuing a row's loads together (4.3-4.9 TB/s).  Here a wave requests the nine loads of its NEXT row
// (x: VPT x 16 B, dy / dres: VPT x 8 B per lane) before it touches the current one.  7.5 KB per row: read 6 KB, write 1.5 KB.
template <int VPT>
__global__ __launch_bounds__(512, 1) void ln_bwd_stream_kernel(const h16* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, int rows, int H,
                                                                const h16* __restrict__ dres, h16* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ dbias, int rows_per_block) {
  H16_SATURATE();
  constexpr int WAVES = 8;        // two per SIMD, 256 registers each: two rows of loads in flight per wave beside the current row
  extern __shared__ __attribute__((aligned(16))) float red[];   // [8][3][H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 gw[VPT], ag[VPT], ab[VPT], ac[VPT];
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    gw[j] = *reinterpret_cast<const f32x4*>(gamma + (lane + 64 * j) * 4);
    ag[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ac[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  struct RowRegs {
    f32x4 x[VPT];
    uint2 d[VPT], r[VPT];
    float mu, rs;
  };
  auto request = [&](RowRegs& q, int row) {
    const size_t o = (size_t)row * H;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int c = (lane + 64 * j) * 4;
      q.x[j] = *reinterpret_cast<const f32x4*>(x + o + c);
      q.d[j] = *reinterpret_cast<const uint2*>(dy + o + c);
      q.r[j] = *reinterpret_cast<const uint2*>(dres + o + c);
    }
    q.mu = mean[row]; q.rs = rstd[row];
  };
  auto process = [&](const RowRegs& q, int row) {
    const float mu = q.mu, rs = q.rs;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const float2 a = unpack_h16x2(q.d[j].x), b = unpack_h16x2(q.d[j].y);
      const float d[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (q.x[j][e] - mu) * rs;
        const float gy = d[e] * gw[j][e];
        s1 += gy;
        s2 += gy * xh;
        ag[j][e] += d[e] * xh;
        ab[j][e] += d[e];
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    const size_t o = (size_t)row * H;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      const int c = (lane + 64 * j) * 4;
      const float2 a = unpack_h16x2(q.d[j].x), b = unpack_h16x2(q.d[j].y);
      const float2 ra = unpack_h16x2(q.r[j].x), rb = unpack_h16x2(q.r[j].y);
      const float d[4] = {a.x, a.y, b.x, b.y};
      const float r[4] = {ra.x, ra.y, rb.x, rb.y};
      float out[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (q.x[j][e] - mu) * rs;
        out[e] = rs * (d[e] * gw[j][e] - s1 - xh * s2) + r[e];
        ac[j][e] += out[e];
      }
      uint2 w = {pack_h16x2(out[0], out[1]), pack_h16x2(out[2], out[3])};
      *reinterpret_cast<uint2*>(dx + o + c) = w;
    }
  };
  // software pipeline over the wave's rows (row, row + 8, ...): three register sets rotate, two rows are always requested
  // ahead of the one being processed
  RowRegs qa, qb, qc;
  int row = r0 + wave;
  if (row < r1) request(qa, row);
  if (row + WAVES < r1) request(qb, row + WAVES);
  for (; row < r1; row += 3 * WAVES) {
    if (row + 2 * WAVES < r1) request(qc, row + 2 * WAVES);
    process(qa, row);
    if (row + WAVES >= r1) break;
    if (row + 3 * WAVES < r1) request(qa, row + 3 * WAVES);
    process(qb, row + WAVES);
    if (row + 2 * WAVES >= r1) break;
    if (row + 4 * WAVES < r1) request(qb, row + 4 * WAVES);
    process(qc, row + 2 * WAVES);
  }
  ln_bwd_fold<VPT, WAVES>(ag, ab, ac, red, dgamma, dbeta, dbias, lane, wave);
}

// out[n] += sum over rows < rows of in[row][n]  (bias gradients); N % 256 == 0.  16-wave blocks, 256 row blocks: the float
// atomics of all row blocks land on the same N addresses (thousands of adds per address serialise in the memory-side
// atomic units: 3072 four-wave blocks took 30 us for 73 MB, the adds alone ~25 of them), so few, wide blocks
__global__ __launch_bounds__(1024) void colsum_kernel(const h16* __restrict__ in, int ld, int rows, int rows_per_block,
                                                      float* __restrict__ out, long long batch_in, long long batch_out) {
  __shared__ float red[16][256];
  in += (size_t)blockIdx.z * batch_in;       // batched form: matrix z of a stack, sums into vector z (uniform strides)
  out += (size_t)blockIdx.z * batch_out;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int r = r0 + wave; r < r1; r += 16) {
    const uint2 w = *reinterpret_cast<const uint2*>(in + (size_t)r * ld + c0);
    const float2 x = unpack_h16x2(w.x), y = unpack_h16x2(w.y);
    a0 += x.x; a1 += x.y; a2 += y.x; a3 += y.y;
  }
  red[wave][lane * 4] = a0; red[wave][lane * 4 + 1] = a1; red[wave][lane * 4 + 2] = a2; red[wave][lane * 4 + 3] = a3;
  __syncthreads();
  if (threadIdx.x < 256) {
    const int c = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][c];
    atomicAdd(out + blockIdx.x * 256 + c, t);
  }
}

// column sums of a head-major tensor [planes][hm_rows][64]: block (plane, row block): 1024 threads = 128 rows x 8 lanes of 16
// bytes, two rows in flight per thread
__global__ __launch_bounds__(1024) void colsum_hm_kernel(const h16* __restrict__ in, int rows, int hm_rows, int rows_per_block,
                                                         float* __restrict__ out, long long batch_in, long long batch_out) {
  __shared__ float red[128][65];
  const h16* base = in + (size_t)blockIdx.z * batch_in + (size_t)blockIdx.x * hm_rows * 64;
  const int t = threadIdx.x, r_in = t >> 3, c8 = (t & 7) * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto add = [&](const u32x4& w) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float2 x = unpack_h16x2(w[k]);
      a[2 * k] += x.x; a[2 * k + 1] += x.y;
    }
  };
  int r = r0 + r_in;
  for (; r + 128 < r1; r += 256) {
    const u32x4 w0 = *reinterpret_cast<const u32x4*>(base + (size_t)r * 64 + c8);
    const u32x4 w1 = *reinterpret_cast<const u32x4*>(base + (size_t)(r + 128) * 64 + c8);
    add(w0); add(w1);
  }
  if (r < r1) add(*reinterpret_cast<const u32x4*>(base + (size_t)r * 64 + c8));
#pragma unroll
  for (int k = 0; k < 8; ++k) red[r_in][c8 + k] = a[k];
  __syncthreads();
  if (t < 64) {
    float s_ = 0.f;
#pragma unroll 8
    for (int q = 0; q < 128; ++q) s_ += red[q][t];
    atomicAdd(out + (size_t)blockIdx.z * batch_out + blockIdx.x * 64 + t, s_);
  }
}

}  // namespace

extern "C" int vault_layernorm_fwd(const vault_ln_fwd_args* a, void* stream) {
  if (!a || a->H % 256 || (a->H > 1024 && a->H != 1536) || a->rows <= 0) return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const RowMap xm{a->x_rpg, a->x_gstride, a->x_goff}, ym{a->y_rpg, a->y_gstride, a->y_goff};
  dim3 grid((a->rows + 3) / 4), block(256);
#define LN_FWD(V)                                                                                          \
  hipLaunchKernelGGL(ln_fwd_kernel<V>, grid, block, 0, st, a->x, xm, a->gamma, a->beta, a->eps, a->rows,   \
                     a->H, reinterpret_cast<h16*>(a->y_bf16), a->y_f32, ym, a->post_add, a->mean, a->rstd, \
                     a->drop_thresh, a->drop_seed, a->drop_stream, a->drop_scale, reinterpret_cast<h16*>(a->y_split3), \
                     reinterpret_cast<uint8_t*>(a->y_q), reinterpret_cast<uint8_t*>(a->y_scale))
  switch (a->H / 256) {
    case 1: LN_FWD(1); break;
    case 2: LN_FWD(2); break;
    case 3: LN_FWD(3); break;
    case 4: LN_FWD(4); break;
    case 6: LN_FWD(6); break;
  }
#undef LN_FWD
  return (int)hipGetLastError();
}

extern "C" int vault_layernorm_bwd(const vault_ln_bwd_args* a, void* stream) {
  if (!a || a->H % 256 || (a->H > 1024 && a->H != 1536) || a->rows <= 0) return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const RowMap dym{a->dy_rpg, a->dy_gstride, a->dy_goff}, xm{a->x_rpg, a->x_gstride, a->x_goff},
      dxm{a->dx_rpg, a->dx_gstride, a->dx_goff};
  constexpr int waves = 16;
  const int maxb = 4096 / waves;      // 16 resident waves per CU
  int rpb = (a->rows + maxb - 1) / maxb;
  rpb = ((rpb + waves - 1) / waves) * waves;
  dim3 grid((a->rows + rpb - 1) / rpb), block(waves * 64);
  int lds_bytes = 0;
  // the 16-bit gradient stream of the pre-LN stack (engine.GRAD_STREAM_BF16): straight-line kernel with row prefetch
  const bool ident = a->dy_rpg == 0 && a->x_rpg == 0 && a->dx_rpg == 0;
  if (a->H == 768 && ident && a->dy_bf16 && !a->dy_f32 && !a->dres && a->dres_bf16 && !a->dx_f32 &&
      a->dx_bf16 && a->drop_thresh == 0 && a->dgamma && a->dbeta) {
    constexpr int LDS = 8 * 3 * 768 * 4;
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VAULT_EINVAL;
    if (!done[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_stream_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS) != hipSuccess) return VAULT_EINVAL;
      done[dev] = true;
    }
    int rpb8 = (a->rows + 255) / 256;                 // one 8-wave block per CU
    rpb8 = ((rpb8 + 7) / 8) * 8;
    hipLaunchKernelGGL((ln_bwd_stream_kernel<3>), dim3((a->rows + rpb8 - 1) / rpb8), dim3(512), LDS, st,
                       reinterpret_cast<const h16*>(a->dy_bf16), a->x, a->mean,
                       a->rstd, a->gamma, a->rows, a->H, reinterpret_cast<const h16*>(a->dres_bf16),
                       reinterpret_cast<h16*>(a->dx_bf16), a->dgamma, a->dbeta, a->dbias, rpb8);
    return (int)hipGetLastError();
  }
#define LN_BWD(V) LN_BWD_W(V, 16)
#define LN_BWD_W(V, W)                                                                                              \
  {                                                                                                                 \
  {                                                                                                                 \
    constexpr int LDS = (W >= 8 ? 8 : 4) * 3 * V * 256 * 4;                                                         \
    if (LDS > 64 * 1024) {                                                                                          \
      static bool done[64] = {};                                                                                    \
      int dev = 0;                                                                                                  \
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VAULT_EINVAL;                            \
      if (!done[dev]) {                                                                                             \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<V, W>),                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return VAULT_EINVAL; \
        done[dev] = true;                                                                                           \
      }                                                                                                             \
    }                                                                                                               \
    lds_bytes = LDS;                                                                                                \
  }                                                                                                                 \
  hipLaunchKernelGGL((ln_bwd_kernel<V, W>), grid, block, lds_bytes, st, reinterpret_cast<const h16*>(a->dy_bf16), a->dy_f32, \
                     dym, a->x, xm, a->mean, a->rstd, a->gamma, a->rows, a->H, a->dres, a->dx_f32,               \
                     reinterpret_cast<h16*>(a->dx_bf16), dxm, a->dgamma, a->dbeta, a->dbias, rpb, a->drop_thresh,          \
                     a->drop_seed, a->drop_stream, a->drop_scale, a->drop_on_dy, reinterpret_cast<const h16*>(a->dres_bf16)); \
  }
  switch (a->H / 256) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    case 4: LN_BWD(4); break;
    case 6: LN_BWD(6); break;
  }
#undef LN_BWD
#undef LN_BWD_W
  return (int)hipGetLastError();
}

// out[b][c] += sum over rows of part[b][rows][n]: grid (n / 256, batch, row slices); a thread sums its slice of one column with
// four loads in flight, one float atomic per slice
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ part, int nparts, int n,
                                                              float* __restrict__ out, long long batch_in, long long batch_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const int per = (nparts + (int)gridDim.z - 1) / (int)gridDim.z;
  const int r0 = blockIdx.z * per, r1 = min(nparts, r0 + per);
  const float* p = part + (size_t)blockIdx.y * batch_in + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += p[(size_t)r * n]; a1 += p[(size_t)(r + 1) * n]; a2 += p[(size_t)(r + 2) * n]; a3 += p[(size_t)(r + 3) * n];
  }
  for (; r < r1; ++r) a0 += p[(size_t)r * n];
  if (r1 > r0) atomicAdd(out + (size_t)blockIdx.y * batch_out + c, (a0 + a1) + (a2 + a3));
}

extern "C" int vault_colsum_partials(const float* part, int nparts, int n, float* out, int batch, long long batch_in,
                                     long long batch_out, void* stream) {
  if (!part || !out || nparts <= 0 || n <= 0 || batch <= 0 || batch > 65535) return VAULT_EINVAL;
  const int slices = nparts >= 64 ? 16 : 1;
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((n + 255) / 256, batch, slices), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), part, nparts, n, out, batch_in, batch_out);
  return (int)hipGetLastError();
}

extern "C" int vault_colsum_hm(const void* in_bf16, int rows, int hm_rows, int planes, float* out, int batch, long long batch_in,
                               long long batch_out, void* stream) {
  if (!in_bf16 || !out || rows <= 0 || hm_rows < rows || planes <= 0 || batch <= 0 || batch > 65535 || (batch_in & 3))
    return VAULT_EINVAL;
  const int row_blocks = std::max(4, 256 / (planes * batch));
  int rpb = (rows + row_blocks - 1) / row_blocks;
  rpb = ((rpb + 127) / 128) * 128;
  dim3 grid(planes, (rows + rpb - 1) / rpb, batch);
  hipLaunchKernelGGL(colsum_hm_kernel, grid, dim3(1024), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const h16*>(in_bf16), rows, hm_rows, rpb, out, batch_in, batch_out);
  return (int)hipGetLastError();
}

extern "C" int vault_colsum(const void* in_bf16, int ld, int rows, int N, float* out, void* stream) {
  if (!in_bf16 || !out || N % 256 || rows <= 0) return VAULT_EINVAL;
  const int row_blocks = 256;
  int rpb = (rows + row_blocks - 1) / row_blocks;
  rpb = ((rpb + 15) / 16) * 16;
  dim3 grid(N / 256, (rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(1024), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const h16*>(in_bf16), ld, rows, rpb, out, 0ll, 0ll);
  return (int)hipGetLastError();
}

// `batch` matrices at element stride batch_in, sums into `batch` vectors at float stride batch_out: the QKV bias gradients
// of a group of layers in one launch, next to the group's batched weight gradients (at small batches one matrix is a
// 4 us read behind a 10 us launch + reduction tail)
extern "C" int vault_colsum_batched(const void* in_bf16, int ld, int rows, int N, float* out, int batch, long long batch_in,
                                    long long batch_out, void* stream) {
  if (!in_bf16 || !out || N % 256 || rows <= 0 || batch <= 0 || batch > 65535 || (batch_in & 3)) return VAULT_EINVAL;
  const int row_blocks = std::max(16, 256 / batch);
  int rpb = (rows + row_blocks - 1) / row_blocks;
  rpb = ((rpb + 15) / 16) * 16;
  dim3 grid(N / 256, (rows + rpb - 1) / rpb, batch);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(1024), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const h16*>(in_bf16), ld, rows, rpb, out, batch_in, batch_out);
  return (int)hipGetLastError();
}
