// Shared GEMM epilogue: accumulators (16x16 MFMA C layout) -> per-wave LDS scratch -> full row
// segments per lane -> fused epilogue math -> 16-byte global stores (or 256-byte atomic rows).
#pragma once
#include "common.h"
#include "gemm.h"

template <int TM, int TN, int EPI, int WM_, int WN_>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[TM][TN], const GemmParams& p, char* smem, int m0, int n0,
                                              int wm0, int wn0, int wave, int lane) {
  const int g = lane >> 4, l15 = lane & 15;
  // ---- epilogue: accumulators -> per-wave LDS scratch -> row segments of SEG floats per lane
  __syncthreads();
  constexpr int SEG = TN * 4;                 // floats per lane per row
  constexpr int LDS_LD = TN * 16 + 4;         // padded scratch row (floats)
  float* sc = reinterpret_cast<float*>(smem) + wave * (16 * LDS_LD);
  const int rr = lane >> 2, cs = (lane & 3) * SEG;
  constexpr bool BF16_OUT = (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_DGELU);
  float csum[BF16_OUT ? SEG : 1];
  if constexpr (BF16_OUT) {
#pragma unroll
    for (int c = 0; c < SEG; ++c) csum[c] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[(4 * g + r) * LDS_LD + j * 16 + l15] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    if constexpr (EPI == EPI_F32_ATOMIC && (TN % 4) == 0) {
      // one wave-instruction = one full 256-byte output row segment: the shape float atomics run fastest at
      float* out = reinterpret_cast<float*>(p.out);
      const bool plain = (p.splits == 1 && p.accumulate == 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 16 + r;
        if (m < p.m_valid) {
#pragma unroll
          for (int cc = 0; cc < TN * 16; cc += 64) {
            const float val = sc[r * LDS_LD + cc + lane];
            float* dst = out + (size_t)m * p.ldo + n0 + wn0 + cc + lane;
            if (plain) *dst = val; else atomicAdd(dst, val);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    float v[SEG];
#pragma unroll
    for (int c = 0; c < SEG; c += 4) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(&sc[rr * LDS_LD + cs + c]);
      v[c] = t4[0]; v[c + 1] = t4[1]; v[c + 2] = t4[2]; v[c + 3] = t4[3];
    }
    __builtin_amdgcn_wave_barrier();
    const int m = m0 + wm0 + i * 16 + rr;
    const int n = n0 + wn0 + cs;
    if (m >= p.m_valid) continue;
    size_t orow = (size_t)m;
    if constexpr (EPI == EPI_F32_PATCH) {
      const int grp = m / p.rpg, pi = m - grp * p.rpg;
      orow = (size_t)grp * p.gstride + p.goff + pi;
#pragma unroll
      for (int c = 0; c < SEG; c += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.addtab + (size_t)pi * p.N + n + c);
        v[c] += t4[0]; v[c + 1] += t4[1]; v[c + 2] += t4[2]; v[c + 3] += t4[3];
      }
    } else if constexpr (EPI != EPI_F32_ATOMIC) {
      if (p.bias != nullptr) {
#pragma unroll
        for (int c = 0; c < SEG; c += 4) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.bias + n + c);
          v[c] += t4[0]; v[c + 1] += t4[1]; v[c + 2] += t4[2]; v[c + 3] += t4[3];
        }
      }
    }
    const size_t o = orow * p.ldo + n;
    if constexpr (EPI == EPI_F32_ATOMIC) {
      float* out = reinterpret_cast<float*>(p.out) + o;
      if (p.splits == 1 && p.accumulate == 0) {
#pragma unroll
        for (int c = 0; c < SEG; c += 4)
          *reinterpret_cast<f32x4*>(out + c) = f32x4{v[c], v[c + 1], v[c + 2], v[c + 3]};
      } else {
#pragma unroll
        for (int c = 0; c < SEG; ++c) atomicAdd(out + c, v[c]);
      }
    } else if constexpr (EPI == EPI_F32_RES || EPI == EPI_F32_PATCH) {
      float* out = reinterpret_cast<float*>(p.out) + o;
      if constexpr (EPI == EPI_F32_RES) {
        if (p.drop_thresh != 0u) {
          const float sc_keep = p.drop_scale;
#pragma unroll
          for (int c = 0; c < SEG; ++c)
            v[c] = dropout_keep(p.drop_seed, p.drop_stream, (uint32_t)(o + c), p.drop_thresh) ? v[c] * sc_keep : 0.f;
        }
        if (p.res != nullptr) {
#pragma unroll
          for (int c = 0; c < SEG; c += 4) {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.res + o + c);
            v[c] += t4[0]; v[c + 1] += t4[1]; v[c + 2] += t4[2]; v[c + 3] += t4[3];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < SEG; c += 4)
        *reinterpret_cast<f32x4*>(out + c) = f32x4{v[c], v[c + 1], v[c + 2], v[c + 3]};
    } else {
      // bf16 outputs
      if constexpr (EPI == EPI_BF16_GELU) {
        // out2 (training only) receives gelu'(pre-activation): backward then needs one multiply per
        // element instead of re-evaluating erf/exp (both epilogues are VALU-bound otherwise)
        if (p.out2 != nullptr) {
          bf16* o2 = reinterpret_cast<bf16*>(p.out2) + o;
#pragma unroll
          for (int c = 0; c < SEG; c += 8) {
            float gp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = v[c + e];
              const float cdf = norm_cdf_f(x);
              gp[e] = cdf + x * (0.3989422804014327f * __expf(-0.5f * x * x));
              v[c + e] = x * cdf;
            }
            u32x4 w = {pack_bf16x2(gp[0], gp[1]), pack_bf16x2(gp[2], gp[3]), pack_bf16x2(gp[4], gp[5]),
                       pack_bf16x2(gp[6], gp[7])};
            *reinterpret_cast<u32x4*>(o2 + c) = w;
          }
        } else {
#pragma unroll
          for (int c = 0; c < SEG; ++c) v[c] = gelu_f(v[c]);
        }
      } else if constexpr (EPI == EPI_BF16_DGELU) {
        const bf16* ax = p.aux + o;   // gelu'(u) as stored by the forward epilogue
#pragma unroll
        for (int c = 0; c < SEG; c += 8) {
          const bf16x8 u = *reinterpret_cast<const bf16x8*>(ax + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c + e] *= (float)u[e];
        }
      } else if constexpr (EPI == EPI_BF16_DROPMASK) {
        // dgrad through a dropout that sits behind this Linear's *output* in forward:
        // handled by the caller masking dY; nothing to do here.
      }
      if constexpr (BF16_OUT) {
#pragma unroll
        for (int c = 0; c < SEG; ++c) csum[c] += v[c];
      }
      bf16* out = reinterpret_cast<bf16*>(p.out) + o;
      if (p.split3) {   // precise path: this output is the next GEMM's A operand -> [hi | lo | hi], ldo = 3N
#pragma unroll
        for (int c = 0; c < SEG; c += 8) {
          bf16 hi[8], lo[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) split_bf16(v[c + e], hi[e], lo[e]);
          u32x4 wh = {pack_bf16x2((float)hi[0], (float)hi[1]), pack_bf16x2((float)hi[2], (float)hi[3]),
                      pack_bf16x2((float)hi[4], (float)hi[5]), pack_bf16x2((float)hi[6], (float)hi[7])};
          u32x4 wl = {pack_bf16x2((float)lo[0], (float)lo[1]), pack_bf16x2((float)lo[2], (float)lo[3]),
                      pack_bf16x2((float)lo[4], (float)lo[5]), pack_bf16x2((float)lo[6], (float)lo[7])};
          *reinterpret_cast<u32x4*>(out + c) = wh;
          *reinterpret_cast<u32x4*>(out + p.N + c) = wl;
          *reinterpret_cast<u32x4*>(out + 2 * p.N + c) = wh;
        }
        continue;
      }
#pragma unroll
      for (int c = 0; c < SEG; c += 8) {
        u32x4 w = {pack_bf16x2(v[c], v[c + 1]), pack_bf16x2(v[c + 2], v[c + 3]),
                   pack_bf16x2(v[c + 4], v[c + 5]), pack_bf16x2(v[c + 6], v[c + 7])};
        *reinterpret_cast<u32x4*>(out + c) = w;
      }
    }
  }
  if constexpr (BF16_OUT) {
    if (p.colsum != nullptr) {   // uniform across the block
      // lanes with equal (lane & 3) hold the same columns for different rows: fold the 16 of them,
      // then fold the WM_ waves that share columns through LDS and issue full 256-byte atomic rows
#pragma unroll
      for (int c = 0; c < SEG; ++c) {
        float t = csum[c];
        t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
        csum[c] = t;
      }
      __syncthreads();                                   // every wave is done with its scratch
      float* cs_lds = reinterpret_cast<float*>(smem);    // [waves][TN*16]
      if (lane < 4) {
#pragma unroll
        for (int c = 0; c < SEG; ++c) cs_lds[wave * (TN * 16) + cs + c] = csum[c];
      }
      __syncthreads();
      constexpr int BN_ = WN_ * TN * 16;
      for (int col = threadIdx.x; col < BN_; col += WM_ * WN_ * 64) {
        const int wcn = col / (TN * 16), cin = col - wcn * (TN * 16);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < WM_; ++r) t += cs_lds[(r * WN_ + wcn) * (TN * 16) + cin];
        atomicAdd(p.colsum + n0 + col, t);
      }
    }
  }
}
