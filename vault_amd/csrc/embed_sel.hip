// Image embeddings for PADDED batches of differently sized images (pixel_mask != 1, canvas != the
// pre-training grid): the device half of ViltEmbeddings.visual_embed (HF:models/vilt/modeling_vilt.py:92-178).
// The host (vault_amd.spec.select_patches) decides WHICH patch slots of the canvas enter the sequence
// (sel [B][L], the image's own patch rows/cols hw [B][2]); the kernels here
//   * unfold exactly those patches into the projection GEMM's A operand,
//   * add each image's own bilinear resize (align_corners = True, zero outside the image) of the G x G
//     position table to its rows of the sequence, and write the CLS rows,
//   * run the matching backward (table gradient by the transposed interpolation).
// All HBM-bound elementwise/gather work: one 16-byte access per lane, rows of H = 768 floats coalesced.
#include <algorithm>
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

// pixel [B][C][HP][WP] f32 -> A [B*L (padded)][C*ps*ps] bf16 ; row (b, l) = patch slot sel[b*L + l] of the
// (HP/ps) x (WP/ps) grid, k = c*ps*ps + py*ps + px
__global__ __launch_bounds__(256) void im2col_sel_kernel(const float* __restrict__ pix, h16* __restrict__ out,
                                                         const int* __restrict__ sel, int B, int L, int Cn, int HP, int WP,
                                                         int ps, long long total_chunks, int split3) {
  H16_SATURATE();
  const int gw = WP / ps;
  const int Kp = Cn * ps * ps;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total_chunks; i += (long long)gridDim.x * 256ll) {
    const long long row = i / (Kp / 8);
    const int kc = (int)(i - row * (Kp / 8));
    const int b = (int)(row / L);
    const int slot = sel[row];
    const int pr = slot / gw, pc = slot - pr * gw;
    const int k = kc * 8;
    const int c = k / (ps * ps), rem = k - c * ps * ps;
    const int py = rem / ps, px = rem - py * ps;
    const float* s = pix + (((size_t)b * Cn + c) * HP + (size_t)(pr * ps + py)) * WP + pc * ps + px;
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
      h16* o = out + row * 3 * Kp + k;
      *reinterpret_cast<u32x4*>(o) = wh;
      *reinterpret_cast<u32x4*>(o + Kp) = wl;
      *reinterpret_cast<u32x4*>(o + 2 * Kp) = wh;
      continue;
    }
    u32x4 w = {pack_h16x2(a[0], a[1]), pack_h16x2(a[2], a[3]), pack_h16x2(d[0], d[1]), pack_h16x2(d[2], d[3])};
    *reinterpret_cast<u32x4*>(out + row * Kp + k) = w;
  }
}

// addtab[l][n] = bias[n] + mtype1[n] (every row: the position part is per image, added after the GEMM) ;
// x[b*S + T][n] = cls[n] + pos[0][n] + mtype1[n]
__global__ __launch_bounds__(256) void image_sel_consts_kernel(const float* __restrict__ bias, const float* __restrict__ pos,
                                                               const float* __restrict__ mtype1, const float* __restrict__ cls,
                                                               float* __restrict__ addtab, float* __restrict__ x, int L, int H,
                                                               int B, int S, int T) {
  const int l = blockIdx.x;  // 0..L ; l == L -> cls rows
  for (int n = threadIdx.x; n < H; n += 256) {
    if (l < L) {
      addtab[(size_t)l * H + n] = bias[n] + mtype1[n];
    } else {
      const float v = cls[n] + pos[n] + mtype1[n];
      for (int b = 0; b < B; ++b) x[((size_t)b * S + T) * H + n] = v;
    }
  }
}

// Bilinear sample (align_corners = True) of the G x G table at patch (i, j) of an h x w image: the four source
// table rows and their weights; zero weights outside the image (the reference zero-pads the resized table).
struct Lerp4 {
  int r[4];
  float w[4];
};
__device__ __forceinline__ Lerp4 lerp_of(int slot, int gw, int h, int w, int G) {
  Lerp4 q;
  const int i = slot / gw, j = slot - i * gw;
  if (i >= h || j >= w) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { q.r[k] = 0; q.w[k] = 0.f; }
    return q;
  }
  const float sy = h > 1 ? (float)(G - 1) / (float)(h - 1) : 0.f;
  const float sx = w > 1 ? (float)(G - 1) / (float)(w - 1) : 0.f;
  const float fy = sy * (float)i, fx = sx * (float)j;
  const int y0 = min((int)fy, G - 1), x0 = min((int)fx, G - 1);
  const int y1 = min(y0 + 1, G - 1), x1 = min(x0 + 1, G - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  q.r[0] = y0 * G + x0; q.w[0] = (1.f - ly) * (1.f - lx);
  q.r[1] = y0 * G + x1; q.w[1] = (1.f - ly) * lx;
  q.r[2] = y1 * G + x0; q.w[2] = ly * (1.f - lx);
  q.r[3] = y1 * G + x1; q.w[3] = ly * lx;
  return q;
}

// x[b*S + T + 1 + l] += resize_b(pos[1:])[sel[b][l]]      (one block per (l, b); H/4 lanes x float4)
__global__ __launch_bounds__(256) void image_pos_sel_fwd_kernel(float* __restrict__ x, const float* __restrict__ pos,
                                                                const int* __restrict__ sel, const int* __restrict__ hw,
                                                                int L, int S, int T, int H, int gw, int G) {
  const int l = blockIdx.x, b = blockIdx.y;
  const Lerp4 q = lerp_of(sel[b * L + l], gw, hw[2 * b], hw[2 * b + 1], G);
  float* xr = x + ((size_t)b * S + T + 1 + l) * H;
  const float* tab = pos + H;   // row 0 of the table is the CLS position
  for (int n = threadIdx.x * 4; n < H; n += 1024) {
    f32x4 v = *reinterpret_cast<const f32x4*>(xr + n);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (q.w[k] != 0.f) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tab + (size_t)q.r[k] * H + n);
        v += q.w[k] * t;
      }
    }
    *reinterpret_cast<f32x4*>(xr + n) = v;
  }
}

// backward over the image rows of dx [B*S][H]: CLS rows (blockIdx.x == L): dpos[0], dmtype1, dcls += sum_b dx ;
// patch rows l: dyp[b*L + l] = bf16(dx row), dbias, dmtype1 += sum_b dx, dpos[1 + r_k] += w_k dx (transposed
// interpolation).  Masked padding rows carry an exactly zero gradient (no valid query attends to them, nothing
// reads their output), so they need no special case.
__global__ __launch_bounds__(256) void image_sel_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dpos,
                                                            float* __restrict__ dmtype1, float* __restrict__ dcls,
                                                            float* __restrict__ dbias, h16* __restrict__ dyp,
                                                            const int* __restrict__ sel, const int* __restrict__ hw, int L,
                                                            int H, int B, int S, int T, int gw, int G, int b_per_block) {
  H16_SATURATE();
  const int l = blockIdx.x;   // L -> CLS rows
  const int b0 = blockIdx.y * b_per_block, b1 = min(B, b0 + b_per_block);
  for (int n = threadIdx.x; n < H; n += 256) {
    float acc = 0.f;
    for (int b = b0; b < b1; ++b) {
      if (l == L) {
        acc += dx[((size_t)b * S + T) * H + n];
      } else {
        const float v = dx[((size_t)b * S + T + 1 + l) * H + n];
        acc += v;
        dyp[((size_t)b * L + l) * H + n] = (h16)v;
        if (v != 0.f) {
          const Lerp4 q = lerp_of(sel[b * L + l], gw, hw[2 * b], hw[2 * b + 1], G);
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (q.w[k] != 0.f) atomicAdd(dpos + (size_t)(1 + q.r[k]) * H + n, q.w[k] * v);
        }
      }
    }
    atomicAdd(dmtype1 + n, acc);
    if (l == L) {
      atomicAdd(dpos + n, acc);
      atomicAdd(dcls + n, acc);
    } else {
      atomicAdd(dbias + n, acc);
    }
  }
}

}  // namespace

extern "C" int vault_im2col_sel(const float* pix, void* out_bf16, const int* sel, int B, int L, int C, int HP, int WP,
                                int ps, int split3, void* stream) {
  if (!pix || !out_bf16 || !sel || B <= 0 || L <= 0 || ps % 8 || HP % ps || WP % ps || WP % 4) return VAULT_EINVAL;
  const long long rows = (long long)B * L;
  const long long chunks = rows * (C * ps * ps / 8);
  const int blocks = (int)std::min<long long>((chunks + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(im2col_sel_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pix,
                     reinterpret_cast<h16*>(out_bf16), sel, B, L, C, HP, WP, ps, chunks, split3);
  return (int)hipGetLastError();
}

extern "C" int vault_image_sel_consts(const float* bias, const float* pos, const float* mtype1, const float* cls,
                                      float* addtab, float* x, int L, int H, int B, int S, int T, void* stream) {
  if (!bias || !pos || !mtype1 || !cls || !addtab || !x || L <= 0) return VAULT_EINVAL;
  hipLaunchKernelGGL(image_sel_consts_kernel, dim3(L + 1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), bias, pos,
                     mtype1, cls, addtab, x, L, H, B, S, T);
  return (int)hipGetLastError();
}

extern "C" int vault_image_pos_sel_fwd(float* x, const float* pos, const int* sel, const int* hw, int B, int L, int S,
                                       int T, int H, int gw, int G, void* stream) {
  if (!x || !pos || !sel || !hw || B <= 0 || L <= 0 || H % 4 || G <= 0 || gw <= 0) return VAULT_EINVAL;
  hipLaunchKernelGGL(image_pos_sel_fwd_kernel, dim3(L, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, pos, sel,
                     hw, L, S, T, H, gw, G);
  return (int)hipGetLastError();
}

extern "C" int vault_image_sel_bwd(const float* dx, float* dpos, float* dmtype1, float* dcls, float* dbias, void* dyp_bf16,
                                   const int* sel, const int* hw, int B, int L, int S, int T, int H, int gw, int G,
                                   void* stream) {
  if (!dx || !dpos || !dmtype1 || !dcls || !dbias || !dyp_bf16 || !sel || !hw || B <= 0 || L <= 0) return VAULT_EINVAL;
  const int bpb = 32;   // samples per block: fewer blocks hammer the shared dmtype1 / dbias / dcls addresses with atomics
  hipLaunchKernelGGL(image_sel_bwd_kernel, dim3(L + 1, (B + bpb - 1) / bpb), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dx, dpos, dmtype1, dcls, dbias, reinterpret_cast<h16*>(dyp_bf16),
                     sel, hw, L, H, B, S, T, gw, G, bpb);
  return (int)hipGetLastError();
}
