// Embedding-side kernels of the path (all HBM-bound gather/scatter work, 16-byte accesses):
//   position ids, table gather-sum (+ dense source), table scatter-add (backward), patch im2col,
//   the per-patch additive table + CLS rows of the fused [text | patch] sequence, and the backward
//   reduction over the image rows.
//
// Replaces RobertaEmbeddings/BertEmbeddings (HF:models/roberta/modeling_roberta.py:75-155,
// HF:models/bert/modeling_bert.py:69-107), ViLT TextEmbeddings on inputs_embeds
// (HF:models/vilt/modeling_vilt.py:237-269), ViltPatchEmbeddings' unfold (modeling_vilt.py:290-300) and
// the bookkeeping of ViltEmbeddings.visual_embed / forward for full pixel masks (modeling_vilt.py:92-219).
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

// one wave per sample; T <= 64
__global__ void position_ids_kernel(const long long* __restrict__ ids, int* __restrict__ pos, int T, int mode, int pad) {
  const int b = blockIdx.x, t = threadIdx.x;
  int v = 0;
  if (mode == 1) {
    const bool valid = (t < T) && (ids[(size_t)b * T + t] != pad);
    const unsigned long long m = __ballot(valid);
    const int incl = __popcll(m & ((2ull << t) - 1ull));
    v = valid ? incl + pad : pad;
  } else {
    v = t;
  }
  if (t < T) pos[(size_t)b * T + t] = v;
}

struct Gather3 {
  const float* tab[3];
  const void* idx[3];   // int64 (is64=1) or int32 indices per row, or null => use fixed[k]
  int is64[3];
  int fixed[3];         // used when idx null and >= 0; -1 = table absent; -2 = index is (row % period)
  int period;
};

__device__ __forceinline__ long long load_idx(const Gather3& g, int k, int row) {
  if (g.idx[k] != nullptr)
    return g.is64[k] ? reinterpret_cast<const long long*>(g.idx[k])[row]
                     : (long long)reinterpret_cast<const int*>(g.idx[k])[row];
  if (g.fixed[k] == -2) return row % g.period;
  return g.fixed[k];
}

// out[row] = (src ? src[row] : 0) + sum_k tab_k[idx_k[row]]   ; one wave per row, H % 256 == 0
__global__ __launch_bounds__(256) void gather_sum_kernel(const float* __restrict__ src, Gather3 g, float* __restrict__ out,
                                                         int rows, int H) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  long long ix[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) ix[k] = (g.tab[k] != nullptr) ? load_idx(g, k, row) : 0;
  for (int c = lane * 4; c < H; c += 256) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (src) a = *reinterpret_cast<const f32x4*>(src + (size_t)row * H + c);
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (g.tab[k] != nullptr) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(g.tab[k] + (size_t)ix[k] * H + c);
        a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] += t[3];
      }
    *reinterpret_cast<f32x4*>(out + (size_t)row * H + c) = a;
  }
}

struct Scatter3 {
  float* tab[3];
  const void* idx[3];
  int is64[3];
  int fixed[3];
  int period;
};

// tab_k[idx_k[row]] += d[row].  A block owns a run of rows: indexed tables get one float atomic per
// element (256-byte segments per wave-instruction; rows whose mask is 0 carry an exactly-zero gradient
// and are skipped), single-row tables (fixed index: every row hits the same destination) are summed in
// registers first and added once per block.
__global__ __launch_bounds__(256) void scatter_add_kernel(const float* __restrict__ d, Scatter3 g, int rows, int H,
                                                          const float* __restrict__ rowmask, int rows_per_block) {
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  // a thread owns the columns c, c + 256, ... of the block's rows; four rows per trip so that their loads (index, mask,
  // gradient) are in flight together, and the loads of trip t + 1 are issued BEFORE the atomics of trip t: VMEM operations
  // retire in order, so a trip's loads issued behind the previous trip's atomics would wait for those to complete in the
  // memory-side atomic units first (a chain of 8 round trips per block: 87 us for a 31 MB pass whatever the row count)
  constexpr int MAXC = 6;   // H <= 1536
  const int nc = (H + 255 - (int)threadIdx.x) / 256;
  float fsum[3][MAXC];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int q = 0; q < MAXC; ++q) fsum[k][q] = 0.f;
  // Rows of an INDEXED table that carry the index of the block's first row are summed in registers too and added once per
  // block: a token-type table has one or two rows, i.e. every token of the batch lands on the same H addresses; word /
  // position indices rarely repeat the first row's, those keep their atomics.
  long long ix0[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ix0[k] = -1;
    if (g.tab[k] != nullptr && g.idx[k] != nullptr && r0 < rows)
      ix0[k] = g.is64[k] ? reinterpret_cast<const long long*>(g.idx[k])[r0] : (long long)reinterpret_cast<const int*>(g.idx[k])[r0];
  }
  struct Trip {
    float v[4][MAXC];
    long long ix[4][3];
    bool on[4];
  };
  auto load_trip = [&](int rb, Trip& t) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = rb + u;
      t.on[u] = row < r1 && (rowmask == nullptr || rowmask[min(row, r1 - 1)] != 0.f);
      const int rr = min(row, r1 - 1);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        t.ix[u][k] = 0;
        if (g.tab[k] != nullptr && g.idx[k] != nullptr)
          t.ix[u][k] = g.is64[k] ? reinterpret_cast<const long long*>(g.idx[k])[rr]
                                 : (long long)reinterpret_cast<const int*>(g.idx[k])[rr];
        else if (g.fixed[k] == -2)
          t.ix[u][k] = rr % g.period;
      }
#pragma unroll
      for (int q = 0; q < MAXC; ++q) t.v[u][q] = (q < nc) ? d[(size_t)rr * H + threadIdx.x + 256 * q] : 0.f;
    }
  };
  auto add_trip = [&](const Trip& t) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!t.on[u]) continue;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (g.tab[k] == nullptr) continue;
        const bool indexed = (g.idx[k] != nullptr && t.ix[u][k] != ix0[k]) || g.fixed[k] == -2;   // (uniform over the block)
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
          if (q >= nc) continue;
          if (indexed) atomicAdd(g.tab[k] + (size_t)t.ix[u][k] * H + threadIdx.x + 256 * q, t.v[u][q]);
          else fsum[k][q] += t.v[u][q];
        }
      }
    }
  };
  if (r0 < r1) {
    Trip a, b;
    load_trip(r0, a);
    for (int rb = r0; rb < r1; rb += 8) {
      if (rb + 4 < r1) load_trip(rb + 4, b);
      add_trip(a);
      if (rb + 4 >= r1) break;
      if (rb + 8 < r1) load_trip(rb + 8, a);
      add_trip(b);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (g.tab[k] == nullptr) continue;
    long long dst = -1;
    if (g.idx[k] == nullptr && g.fixed[k] >= 0) dst = g.fixed[k];
    else if (g.idx[k] != nullptr) dst = ix0[k];
    if (dst < 0) continue;
#pragma unroll
    for (int q = 0; q < MAXC; ++q)
      if (q < nc && fsum[k][q] != 0.f) atomicAdd(g.tab[k] + (size_t)dst * H + threadIdx.x + 256 * q, fsum[k][q]);
  }
}

// pixel [B][C][IMG][IMG] f32 -> A [B*P (padded)][C*ps*ps] bf16, k = c*ps*ps + py*ps + px, patches row-major
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ pix, h16* __restrict__ out, int B, int Cn,
                                                     int IMG, int ps, long long total_chunks, int split3) {
  H16_SATURATE();
  // one block per output row (patch): the row -> (sample, patch row, patch column) split is scalar work, the
  // per-chunk index math stays in 32 bits (64-bit divisions per 16-byte chunk made this kernel VALU-bound)
  const int grid = IMG / ps;
  const int Kp = Cn * ps * ps, pp = ps * ps;
  const int rows = (int)(total_chunks / (Kp / 8));
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int b = row / (grid * grid), p = row - b * grid * grid;
    const int pr = p / grid, pc = p - pr * grid;
    const float* base = pix + ((size_t)b * Cn * IMG + (size_t)pr * ps) * IMG + pc * ps;
    h16* orow = out + (size_t)row * (split3 ? 3 * Kp : Kp);
    for (int kc = threadIdx.x; kc < Kp / 8; kc += 256) {
      const int k = kc * 8;
      const int c = k / pp, rem = k - c * pp;
      const int py = rem / ps, px = rem - py * ps;
      const float* s = base + ((size_t)c * IMG + py) * IMG + px;
      const f32x4 a = *reinterpret_cast<const f32x4*>(s);
      const f32x4 d = *reinterpret_cast<const f32x4*>(s + 4);
      if (split3) {
        const float xs[8] = {a[0], a[1], a[2], a[3], d[0], d[1], d[2], d[3]};
        h16 hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split_bf16(xs[e], hi[e], lo[e]);
        u32x4 wh = {pack_h16x2((float)hi[0], (float)hi[1]), pack_h16x2((float)hi[2], (float)hi[3]),
                    pack_h16x2((float)hi[4], (float)hi[5]), pack_h16x2((float)hi[6], (float)hi[7])};
        u32x4 wl = {pack_h16x2((float)lo[0], (float)lo[1]), pack_h16x2((float)lo[2], (float)lo[3]),
                    pack_h16x2((float)lo[4], (float)lo[5]), pack_h16x2((float)lo[6], (float)lo[7])};
        *reinterpret_cast<u32x4*>(orow + k) = wh;
        *reinterpret_cast<u32x4*>(orow + Kp + k) = wl;
        *reinterpret_cast<u32x4*>(orow + 2 * Kp + k) = wh;
        continue;
      }
      u32x4 w = {pack_h16x2(a[0], a[1]), pack_h16x2(a[2], a[3]), pack_h16x2(d[0], d[1]), pack_h16x2(d[2], d[3])};
      *reinterpret_cast<u32x4*>(orow + k) = w;
    }
  }
}

// addtab[p][n] = bias[n] + pos[1+p][n] + mtype1[n]  (p < P) ; x[b*S + T][n] = cls[n] + pos[0][n] + mtype1[n]
__global__ __launch_bounds__(256) void image_consts_kernel(const float* __restrict__ bias, const float* __restrict__ pos,
                                                           const float* __restrict__ mtype1, const float* __restrict__ cls,
                                                           float* __restrict__ addtab, float* __restrict__ x, int P, int H,
                                                           int B, int S, int T) {
  const int p = blockIdx.x;  // 0..P ; p == P -> cls rows
  for (int n = threadIdx.x; n < H; n += 256) {
    if (p < P) {
      addtab[(size_t)p * H + n] = bias[n] + pos[(size_t)(1 + p) * H + n] + mtype1[n];
    } else {
      const float v = cls[n] + pos[n] + mtype1[n];
      for (int b = 0; b < B; ++b) x[((size_t)b * S + T) * H + n] = v;
    }
  }
}

// backward over the image rows of dx [B*S][H] (f32): position j in 0..P (0 = CLS):
//   dpos[j] += sum_b dx[b, T+j] ; dmtype1 += (same) ; j==0: dcls += ; j>=1: dbias += and
//   dyp[b*P + j-1] = bf16(dx[b, T+j])   (compact operand for the projection wgrad)
__global__ __launch_bounds__(256) void image_rows_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dpos,
                                                             float* __restrict__ dmtype1, float* __restrict__ dcls,
                                                             float* __restrict__ dbias, h16* __restrict__ dyp, int P, int H,
                                                             int B, int S, int T, int b_per_block) {
  H16_SATURATE();
  const int j = blockIdx.x;
  const int b0 = blockIdx.y * b_per_block, b1 = min(B, b0 + b_per_block);
  // a thread owns the columns n, n + 256, ... (three at H = 768) of position j: their sample loops run side by side, four
  // samples per trip, so that 12 independent loads are in flight (the serial form was latency-bound: 115 us for 171 MB)
  constexpr int MAXC = 6;   // H <= 1536
  float acc[MAXC];
#pragma unroll
  for (int k = 0; k < MAXC; ++k) acc[k] = 0.f;
  const int nc = (H + 255 - (int)threadIdx.x) / 256;      // columns of this thread
  int b = b0;
  for (; b + 4 <= b1; b += 4) {
    float v[4][MAXC];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < MAXC; ++k)
        v[q][k] = (k < nc) ? dx[((size_t)(b + q) * S + T + j) * H + threadIdx.x + 256 * k] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < MAXC; ++k)
        if (k < nc) {
          acc[k] += v[q][k];
          if (j >= 1) dyp[((size_t)(b + q) * P + (j - 1)) * H + threadIdx.x + 256 * k] = (h16)v[q][k];
        }
  }
  for (; b < b1; ++b)
#pragma unroll
    for (int k = 0; k < MAXC; ++k)
      if (k < nc) {
        const float v = dx[((size_t)b * S + T + j) * H + threadIdx.x + 256 * k];
        acc[k] += v;
        if (j >= 1) dyp[((size_t)b * P + (j - 1)) * H + threadIdx.x + 256 * k] = (h16)v;
      }
#pragma unroll
  for (int k = 0; k < MAXC; ++k)
    if (k < nc) {
      const int n = threadIdx.x + 256 * k;
      atomicAdd(dpos + (size_t)j * H + n, acc[k]);
      atomicAdd(dmtype1 + n, acc[k]);
      if (j == 0) atomicAdd(dcls + n, acc[k]); else atomicAdd(dbias + n, acc[k]);
    }
}

__global__ void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, float a, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256ll) dst[i] += a * src[i];
}

// x *= a, 16 bytes per lane (the flat gradient buffer under the fp16 build's power-of-two gradient scale: exact)
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, float a, long long n4) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256ll) {
    f32x4 v = reinterpret_cast<f32x4*>(x)[i];
    reinterpret_cast<f32x4*>(x)[i] = v * a;
  }
}

__global__ void scale_tail_kernel(float* __restrict__ x, float a, int n) {
  if ((int)threadIdx.x < n) x[threadIdx.x] *= a;
}

}  // namespace

extern "C" int vault_position_ids(const int64_t* ids, int* pos, int B, int T, int mode, int pad, void* stream) {
  if (!ids || !pos || T > 64 || T <= 0 || B <= 0) return VAULT_EINVAL;
  hipLaunchKernelGGL(position_ids_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(ids), pos, T, mode, pad);
  return (int)hipGetLastError();
}

extern "C" int vault_gather_sum(const vault_gather_args* a, void* stream) {
  if (!a || !a->out || a->H % 256 || a->rows <= 0) return VAULT_EINVAL;
  Gather3 g;
  for (int k = 0; k < 3; ++k) {
    g.tab[k] = a->tab[k]; g.idx[k] = a->idx[k]; g.is64[k] = a->is64[k]; g.fixed[k] = a->fixed[k];
  }
  g.period = a->period > 0 ? a->period : 1;
  hipLaunchKernelGGL(gather_sum_kernel, dim3((a->rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     a->src, g, a->out, a->rows, a->H);
  return (int)hipGetLastError();
}

extern "C" int vault_scatter_add(const vault_gather_args* a, void* stream) {
  if (!a || !a->src || a->H % 64 || a->H > 1536 || a->rows <= 0) return VAULT_EINVAL;
  Scatter3 g;
  for (int k = 0; k < 3; ++k) {
    g.tab[k] = const_cast<float*>(a->tab[k]); g.idx[k] = a->idx[k]; g.is64[k] = a->is64[k]; g.fixed[k] = a->fixed[k];
  }
  g.period = a->period > 0 ? a->period : 1;
  // rows per block: a block is a chain of trips (loads -> atomics), so few rows per block while the launch is small
  // (tools/scatter_bench.py, 3 tables, H = 768: 2560 rows 87 us at 32 rows per block, 33 us at 8; 10240 rows 90 / 80 / 89 us
  // at 32 / 16 / 8 - there the float atomics themselves, ~0.2 T/s, are the bound)
  const int rpb = a->rows >= 5120 ? 16 : 8;
  hipLaunchKernelGGL(scatter_add_kernel, dim3((a->rows + rpb - 1) / rpb), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a->src, g, a->rows, a->H, a->rowmask, rpb);
  return (int)hipGetLastError();
}

extern "C" int vault_im2col(const float* pix, void* out_bf16, int B, int C, int IMG, int ps, int split3, void* stream) {
  if (!pix || !out_bf16 || ps % 8 || IMG % ps || B <= 0) return VAULT_EINVAL;
  const long long rows = (long long)B * (IMG / ps) * (IMG / ps);
  const long long chunks = rows * (C * ps * ps / 8);
  const int blocks = (int)std::min<long long>(rows, 256 * 64);
  hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pix,
                     reinterpret_cast<h16*>(out_bf16), B, C, IMG, ps, chunks, split3);
  return (int)hipGetLastError();
}

extern "C" int vault_image_consts(const float* bias, const float* pos, const float* mtype1, const float* cls,
                                  float* addtab, float* x, int P, int H, int B, int S, int T, void* stream) {
  if (!bias || !pos || !mtype1 || !cls || !addtab || !x) return VAULT_EINVAL;
  hipLaunchKernelGGL(image_consts_kernel, dim3(P + 1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), bias, pos,
                     mtype1, cls, addtab, x, P, H, B, S, T);
  return (int)hipGetLastError();
}

extern "C" int vault_image_rows_bwd(const float* dx, float* dpos, float* dmtype1, float* dcls, float* dbias,
                                    void* dyp_bf16, int P, int H, int B, int S, int T, void* stream) {
  if (!dx || !dpos || !dmtype1 || !dcls || !dbias || !dyp_bf16 || H > 1536) return VAULT_EINVAL;
  const int bpb = 32;   // samples per block: fewer blocks hammer the shared dmtype1 / dbias / dcls addresses with atomics
  hipLaunchKernelGGL(image_rows_bwd_kernel, dim3(P + 1, (B + bpb - 1) / bpb), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dx, dpos, dmtype1, dcls, dbias,
                     reinterpret_cast<h16*>(dyp_bf16), P, H, B, S, T, bpb);
  return (int)hipGetLastError();
}

extern "C" int vault_axpy_f32(float* dst, const float* src, float a, long long n, void* stream) {
  if (!dst || !src || n <= 0) return VAULT_EINVAL;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dst, src, a, n);
  return (int)hipGetLastError();
}

extern "C" int vault_scale_f32(float* x, float a, long long n, void* stream) {
  if (!x || n <= 0 || (reinterpret_cast<uintptr_t>(x) & 3)) return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // 16-byte pieces over the aligned middle, single floats for a ragged head / tail (small external gradients: [B, 3] logits)
  long long head = ((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / 4;
  if (head > n) head = n;
  const long long n4 = (n - head) / 4, tail = n - head - 4 * n4;
  if (n4 > 0) {
    const int blocks = (int)std::min<long long>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, st, x + head, a, n4);
  }
  if (head > 0) hipLaunchKernelGGL(scale_tail_kernel, dim3(1), dim3(64), 0, st, x, a, (int)head);
  if (tail > 0) hipLaunchKernelGGL(scale_tail_kernel, dim3(1), dim3(64), 0, st, x + head + 4 * n4, a, (int)tail);
  return (int)hipGetLastError();
}

// ---- externally supplied image embeddings (HF ViltEmbeddings.forward with `image_embeds`, modeling_vilt.py:190-207:
// no patch projection, no CLS token, no position table - only the modality type is added; reached from the reference
// through TomViltForTMSC, ref: vault/models/tomvilt/model.py:281-287).
// fwd: out[map(r)] = src[r] + vec   ; map(r) = (r / rpg) * gstride + goff + r % rpg  (rows of the fused sequence)
// bwd: dsrc[r] = dx[map(r)] ; dvec += column sums of those rows
__global__ __launch_bounds__(256) void rows_add_kernel(const float* __restrict__ src, const float* __restrict__ vec,
                                                       float* __restrict__ out, int rows, int H, int rpg, int gstride, int goff) {
  const int h4 = H >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)rows * h4; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / h4), c = (int)(i - (long long)r * h4) * 4;
    const size_t orow = (size_t)(r / rpg) * gstride + goff + (r % rpg);
    const f32x4 s = *reinterpret_cast<const f32x4*>(src + (size_t)r * H + c);
    const f32x4 v = *reinterpret_cast<const f32x4*>(vec + c);
    *reinterpret_cast<f32x4*>(out + orow * H + c) = s + v;
  }
}

__global__ __launch_bounds__(256) void rows_gather_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dsrc,
                                                              float* __restrict__ dvec, int rows, int H, int rpg, int gstride,
                                                              int goff) {
  // block = 64 rows x all columns (thread t: 4 columns at (t % (H/4)) ... strided over H/4 column groups)
  const int h4 = H >> 2;
  const int r0 = blockIdx.x * 64;
  for (int cg = threadIdx.x; cg < h4; cg += 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < min(rows, r0 + 64); ++r) {
      const size_t xrow = (size_t)(r / rpg) * gstride + goff + (r % rpg);
      const f32x4 v = *reinterpret_cast<const f32x4*>(dx + xrow * H + cg * 4);
      *reinterpret_cast<f32x4*>(dsrc + (size_t)r * H + cg * 4) = v;
      acc += v;
    }
    if (dvec != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(dvec + cg * 4 + e, acc[e]);
    }
  }
}

extern "C" int vault_rows_add_f32(const float* src, const float* vec, float* out, int rows, int H, int rpg, int gstride,
                                  int goff, void* stream) {
  if (!src || !vec || !out || rows <= 0 || H <= 0 || (H & 3) || rpg <= 0) return VAULT_EINVAL;
  const long long n = (long long)rows * (H >> 2);
  hipLaunchKernelGGL(rows_add_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 8192)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), src, vec, out, rows, H, rpg, gstride, goff);
  return (int)hipGetLastError();
}

extern "C" int vault_rows_gather_bwd_f32(const float* dx, float* dsrc, float* dvec, int rows, int H, int rpg, int gstride,
                                         int goff, void* stream) {
  if (!dx || !dsrc || rows <= 0 || H <= 0 || (H & 3) || rpg <= 0) return VAULT_EINVAL;
  hipLaunchKernelGGL(rows_gather_bwd_kernel, dim3((rows + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dx,
                     dsrc, dvec, rows, H, rpg, gstride, goff);
  return (int)hipGetLastError();
}
