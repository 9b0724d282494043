// Shared GEMM epilogue: accumulators (16x16 MFMA C layout) -> per-wave LDS scratch -> row segments
// per lane -> fused epilogue math -> 16-byte global accesses.
//
// Lane mapping after the LDS round trip: a 16-row step is handled in passes of PW columns; in one pass lane
// (rr = lane / 4, seg = lane % 4) owns row rr and, for each of its NI 16-byte accesses k, the GR consecutive
// columns k * 4 * GR + seg * GR ...  (GR = 8 for bf16 outputs, 4 for f32): the four lanes of a row write one
// contiguous 64-byte piece per instruction.
//
// Every global LOAD the epilogue needs is issued ahead of its use: the bias once, the per-row operands
// (residual / gelu' / patch table) PD passes ahead through a small register queue - with one wave per SIMD
// (ring kernel) nothing else would hide their latency, and because VMEM operations retire in order a
// load also waits for every store issued before it: PD passes of distance keep those stores old.
#pragma once
#include "common.h"
#include "gemm.h"

struct GemmEpiNoHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// ASYNC (persistent ring kernel): `smem` is a scratch area OUTSIDE the staging ring, block barriers wait
// for LDS only (never for the global stores in flight), and hook(i) runs once per 16-row step i
// with the whole wave active, before the step's global stores: the kernel issues the next tile's staging
// loads from it and counts on the step's stores being issued after them.
template <int TM, int TN, int EPI, int WM_, int WN_, bool ASYNC = false, typename Hook = GemmEpiNoHook>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[TM][TN], const GemmParams& p, char* smem, int m0, int n0,
                                              int wm0, int wn0, int wave, int lane, Hook hook = Hook(),
                                              int hook0_ops = 0) {
  constexpr bool BF16_OUT = (EPI == EPI_BF16 || EPI == EPI_BF16_GELU || EPI == EPI_BF16_GELU_INF || EPI == EPI_BF16_DGELU);
  constexpr bool ATOMIC = (EPI == EPI_F32_ATOMIC);
  constexpr bool PRE_F32 = (EPI == EPI_F32_RES || EPI == EPI_F32_PATCH);   // per-row f32 operand to prefetch
  constexpr bool PRE_B16 = (EPI == EPI_BF16_DGELU);                         // per-row bf16 operand to prefetch
  constexpr bool HAS_BIAS = !ATOMIC && EPI != EPI_F32_PATCH;
  constexpr int NP = (ASYNC && TN == 8) ? 2 : 1;    // column passes per 16-row step
  constexpr int TNP = TN / NP;                       // 16-column tiles per pass
  constexpr int PW = TNP * 16;                       // pass width (columns)
  constexpr int LD = PW + 4;                         // scratch row stride (floats): LD % 16 == 4 keeps both sides conflict-light
  constexpr int EL = PW / 4;                         // elements per lane per pass
  constexpr int GR = BF16_OUT ? 8 : 4;               // elements per 16-byte global access
#ifndef VAULT_EPI_PD
#define VAULT_EPI_PD 2
#endif
  constexpr int PD = VAULT_EPI_PD * NP;              // prefetch distance (passes) = two 16-row steps
  constexpr int NPASS = TM * NP;
  static_assert(EL % GR == 0 && TN % NP == 0 && LD % 16 == 4, "epilogue tiling");

  if constexpr (ASYNC) lane = lane_id_volatile();   // not a value kept live across the kernel's persistent loop
  const int g = lane >> 4, l15 = lane & 15;
  const int rr = lane >> 2, seg = lane & 3;
  auto block_sync = [] {
    if constexpr (ASYNC) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
  };
  if constexpr (!ASYNC) __syncthreads();   // the scratch aliases the staging buffers
  float* sc = reinterpret_cast<float*>(smem) + wave * (16 * LD);
  float* sc_w = sc + (4 * g) * LD + l15;   // + r * LD + jj * 16
  const float* sc_r = sc + rr * LD + seg * GR;   // + k * 4 * GR + e
  const int nw = n0 + wn0;                 // first column of this wave
  // column (within the pass) of element e of this lane, without the lane part
  auto ecol = [](int e) -> int { return (e / GR) * (4 * GR) + (e % GR); };
  const int lcol = seg * GR;

  float csum[BF16_OUT ? NP * EL : 1];
  if constexpr (BF16_OUT) {
#pragma unroll
    for (int c = 0; c < NP * EL; ++c) csum[c] = 0.f;
  }
  // ---- bias: loaded once (ASYNC: through asm like the per-row operands, retired by pass 0's counted wait)
  float bv[HAS_BIAS ? NP * EL : 1];
  f32x4 bq[(HAS_BIAS && ASYNC) ? NP * EL / 4 : 1];
  bool has_bias = false;
  if constexpr (HAS_BIAS) {
    has_bias = p.bias != nullptr;
    if (has_bias) {
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int e = 0; e < EL; e += 4) {
          const float* src = p.bias + nw + q * PW + lcol;
          if constexpr (ASYNC) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(bq[(q * EL + e) / 4]) : "v"(src), "i"(ecol(e) * 4));
          } else {
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(src + ecol(e));
            bv[q * EL + e] = t4[0]; bv[q * EL + e + 1] = t4[1]; bv[q * EL + e + 2] = t4[2]; bv[q * EL + e + 3] = t4[3];
          }
        }
    } else if constexpr (ASYNC) {
#pragma unroll
      for (int k = 0; k < NP * EL / 4; ++k) bq[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  auto out_row = [&](int m) -> size_t {
    if constexpr (EPI == EPI_F32_PATCH) {
      const int grp = m / p.rpg;
      return (size_t)grp * p.gstride + p.goff + (m - grp * p.rpg);
    } else {
      return (size_t)m;
    }
  };
  // ---- per-row operands, queued PD passes ahead
  struct Pre {
    f32x4 f[PRE_F32 ? EL / 4 : 1];
    h16x8 h[PRE_B16 ? EL / 8 : 1];
  };
  Pre pq[(PRE_F32 || PRE_B16) ? PD : 1];
  // ASYNC: the loads are issued through asm (invisible to hipcc's wait insertion, which would answer the
  // divergent store branches with vmcnt(0) at every use and so wait for every store and staging load in
  // flight); they are unconditional (rows >= m_valid exist in the padded buffers) and retired below by a
  // counted wait that allows for everything certain to have been issued behind them.
  constexpr int NLOAD = PRE_F32 ? EL / 4 : (PRE_B16 ? EL / 8 : 0);   // VMEM loads per pass
  auto prefetch = [&](Pre& d, int i, int q) {
    const int m = m0 + wm0 + i * 16 + rr;
    if constexpr (!ASYNC) {
      if (m >= p.m_valid) return;
    }
    const int nc = nw + q * PW + lcol;
    if constexpr (PRE_F32) {
      const float* src;
      if constexpr (EPI == EPI_F32_RES) {
        if constexpr (!ASYNC) {
          if (p.res == nullptr) return;   // uniform (ASYNC: the launcher guarantees a residual operand)
        }
        src = p.res + (size_t)m * p.ldo + nc;
      } else {
        src = p.addtab + (size_t)(m % p.rpg) * p.N + nc;
      }
#pragma unroll
      for (int k = 0; k < EL / 4; ++k) {
        if constexpr (ASYNC)
          asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(d.f[k]) : "v"(src), "i"(ecol(4 * k) * 4));
        else
          d.f[k] = *reinterpret_cast<const f32x4*>(src + ecol(4 * k));
      }
    } else if constexpr (PRE_B16) {
      const h16* src = p.aux + (size_t)m * p.ldo + nc;
#pragma unroll
      for (int k = 0; k < EL / 8; ++k) {
        if constexpr (ASYNC)
          asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(d.h[k]) : "v"(src), "i"(ecol(8 * k) * 2));
        else
          d.h[k] = *reinterpret_cast<const h16x8*>(src + ecol(8 * k));
      }
    }
  };
  const bool full_tile = m0 + wm0 + TM * 16 <= p.m_valid;            // uniform per wave: every store below is issued
  if constexpr (PRE_F32 || PRE_B16) {
#pragma unroll
    for (int s = 0; s < PD; ++s)
      if (s < NPASS) prefetch(pq[s], s / NP, s % NP);
  }

#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      // ---- accumulators of step i, pass q -> scratch (uniform select of the register group in the rolled form)
#pragma unroll
      for (int ii = 0; ii < TM; ++ii) {
        if (i == ii) {
#pragma unroll
          for (int jj = 0; jj < TNP; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sc_w[r * LD + jj * 16] = acc[ii][q * TNP + jj][r];
            }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if constexpr (ATOMIC && (PW % 64) == 0) {
        if constexpr (ASYNC) {
          if (q == 0) {
            __builtin_amdgcn_sched_barrier(0);
            hook(i);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // one wave-instruction = one full 256-byte output row segment: the shape float atomics run fastest at
        float* out = reinterpret_cast<float*>(p.out);
        const bool plain = (p.splits == 1 && p.accumulate == 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + i * 16 + r;
          if (m < p.m_valid) {
#pragma unroll
            for (int cc = 0; cc < PW; cc += 64) {
              const float val = sc[r * LD + cc + lane];
              float* dst = out + (size_t)m * p.ldo + nw + q * PW + cc + lane;
              if (plain) *dst = val; else atomicAdd(dst, val);
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      float v[EL];
#pragma unroll
      for (int e = 0; e < EL; e += 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(sc_r + ecol(e));
        v[e] = t4[0]; v[e + 1] = t4[1]; v[e + 2] = t4[2]; v[e + 3] = t4[3];
      }
      __builtin_amdgcn_wave_barrier();
      // per-row operands of this pass: queue slot s % PD (static slots, no moves)
      const int s_lin = i * NP + q;
      Pre cur = pq[(PRE_F32 || PRE_B16) ? (s_lin % PD) : 0];
      if constexpr (ASYNC) {   // nothing may move across: the kernel counts the VMEM operations issued after each hook
        if (q == 0) {
          __builtin_amdgcn_sched_barrier(0);
          hook(i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (PRE_F32 || PRE_B16) {
        if constexpr (ASYNC) {
          // retire this pass's operands with a counted wait.  VMEM operations certain to have been issued behind
          // them: the prefetches of the passes in between and, in a fully valid tile, the stores of the passes
          // since their issue (>= SPP per pass); staging loads and second outputs only add to that.
          constexpr int SPP = BF16_OUT ? EL / 8 : EL / 4;
          int nl = (s_lin < PD) ? (PD - 1 - s_lin) : 0;               // pre-loop prefetches behind this one
          for (int t = (s_lin - PD + 1 > 0 ? s_lin - PD + 1 : 0); t < s_lin; ++t) nl += (t + PD < NPASS) ? 1 : 0;
          const int n_part = nl * NLOAD;
          const int n_full_ = n_part + (s_lin < PD ? s_lin : PD) * SPP;
          const int n_full = n_full_ > 63 ? 63 : n_full_;
          {
            if (full_tile) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n_full) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n_part) : "memory");
#pragma unroll
            for (int k = 0; k < (PRE_F32 ? EL / 4 : 1); ++k) asm volatile("" : "+v"(cur.f[k]));
#pragma unroll
            for (int k = 0; k < (PRE_B16 ? EL / 8 : 1); ++k) asm volatile("" : "+v"(cur.h[k]));
          }
        }
        if (s_lin + PD < NPASS) prefetch(pq[s_lin % PD], (s_lin + PD) / NP, (s_lin + PD) % NP);
      }
      if constexpr (HAS_BIAS && ASYNC) {
        if (s_lin == 0) {
          // behind the bias loads: the queued per-row prefetches (their own wait above covers the bias too) or
          // just the staging loads of hook(0)
          if constexpr (!(PRE_F32 || PRE_B16)) {
            if (hook0_ops >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
#pragma unroll
          for (int k = 0; k < NP * EL / 4; ++k) asm volatile("" : "+v"(bq[k]));
        }
      }
      const int m = m0 + wm0 + i * 16 + rr;
      if (m >= p.m_valid) continue;
      const size_t o = out_row(m) * p.ldo + nw + q * PW + lcol;   // + ecol(e)
      if constexpr (ATOMIC) {   // generic fallback
        float* out = reinterpret_cast<float*>(p.out) + o;
#pragma unroll
        for (int e = 0; e < EL; ++e) atomicAdd(out + ecol(e), v[e]);
        continue;
      }
      if constexpr (HAS_BIAS) {
        if constexpr (ASYNC) {
#pragma unroll
          for (int e = 0; e < EL; ++e) v[e] += bq[(q * EL + e) >> 2][e & 3];
        } else if (has_bias) {
#pragma unroll
          for (int e = 0; e < EL; ++e) v[e] += bv[q * EL + e];
        }
      }
      if constexpr (EPI == EPI_F32_RES || EPI == EPI_F32_PATCH) {
        float* out = reinterpret_cast<float*>(p.out) + o;
        if constexpr (EPI == EPI_F32_RES) {
          if (p.drop_thresh != 0u) {
            const float sc_keep = p.drop_scale;
#pragma unroll
            for (int e = 0; e < EL; ++e)
              v[e] = dropout_keep(p.drop_seed, p.drop_stream, (uint32_t)(o + ecol(e)), p.drop_thresh) ? v[e] * sc_keep : 0.f;
          }
          if (ASYNC || p.res != nullptr) {
#pragma unroll
            for (int e = 0; e < EL; ++e) v[e] += cur.f[e >> 2][e & 3];
          }
        } else {
#pragma unroll
          for (int e = 0; e < EL; ++e) v[e] += cur.f[e >> 2][e & 3];
        }
#pragma unroll
        for (int e = 0; e < EL; e += 4)
          *reinterpret_cast<f32x4*>(out + ecol(e)) = f32x4{v[e], v[e + 1], v[e + 2], v[e + 3]};
      } else if constexpr (BF16_OUT) {
        if constexpr (EPI == EPI_BF16_GELU || EPI == EPI_BF16_GELU_INF) {
          // out2 (training only) receives gelu'(pre-activation): backward then needs one multiply per
          // element instead of re-evaluating erf/exp (both epilogues are VALU-bound otherwise).  The ring
          // kernel instantiates the two forms separately (its launcher looks at out2): half the epilogue code.
          if (EPI == EPI_BF16_GELU && (ASYNC || p.out2 != nullptr)) {
            h16* o2 = reinterpret_cast<h16*>(p.out2) + o;
#pragma unroll
            for (int c = 0; c < EL; c += 8) {
              float gp[8];
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                f32x2 y2, d2;
                gelu_fwd_f2(f32x2{v[c + e], v[c + e + 1]}, y2, d2);
                gp[e] = d2[0]; gp[e + 1] = d2[1];
                v[c + e] = y2[0]; v[c + e + 1] = y2[1];
              }
              u32x4 w = {pack_h16x2(gp[0], gp[1]), pack_h16x2(gp[2], gp[3]), pack_h16x2(gp[4], gp[5]),
                         pack_h16x2(gp[6], gp[7])};
              *reinterpret_cast<u32x4*>(o2 + ecol(c)) = w;
            }
          } else {
#pragma unroll
            for (int e = 0; e < EL; e += 2) {
              const f32x2 x2 = {v[e], v[e + 1]};
              const f32x2 y2 = x2 * norm_cdf_f2(x2);
              v[e] = y2[0]; v[e + 1] = y2[1];
            }
          }
        } else if constexpr (EPI == EPI_BF16_DGELU) {
#pragma unroll
          for (int e = 0; e < EL; ++e) v[e] *= (float)cur.h[e >> 3][e & 7];   // aux = gelu' stored by the forward
        }
#pragma unroll
        for (int e = 0; e < EL; ++e) csum[q * EL + e] += v[e];
        h16* out = reinterpret_cast<h16*>(p.out) + o;
        if (p.split3) {   // precise path: this output is the next GEMM's A operand -> [hi | lo | hi], ldo = 3N
#pragma unroll
          for (int c = 0; c < EL; c += 8) {
            h16 hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split_bf16(v[c + e], hi[e], lo[e]);
            u32x4 wh = {pack_h16x2((float)hi[0], (float)hi[1]), pack_h16x2((float)hi[2], (float)hi[3]),
                        pack_h16x2((float)hi[4], (float)hi[5]), pack_h16x2((float)hi[6], (float)hi[7])};
            u32x4 wl = {pack_h16x2((float)lo[0], (float)lo[1]), pack_h16x2((float)lo[2], (float)lo[3]),
                        pack_h16x2((float)lo[4], (float)lo[5]), pack_h16x2((float)lo[6], (float)lo[7])};
            *reinterpret_cast<u32x4*>(out + ecol(c)) = wh;
            *reinterpret_cast<u32x4*>(out + p.N + ecol(c)) = wl;
            *reinterpret_cast<u32x4*>(out + 2 * p.N + ecol(c)) = wh;
          }
          continue;
        }
#pragma unroll
        for (int c = 0; c < EL; c += 8) {
          u32x4 w = {pack_h16x2(v[c], v[c + 1]), pack_h16x2(v[c + 2], v[c + 3]),
                     pack_h16x2(v[c + 4], v[c + 5]), pack_h16x2(v[c + 6], v[c + 7])};
          *reinterpret_cast<u32x4*>(out + ecol(c)) = w;
        }
      }
    }
  }
  if constexpr (BF16_OUT) {
    if (p.colsum != nullptr) {   // uniform across the block
      // lanes with equal (lane & 3) hold the same columns for different rows: fold the 16 of them,
      // then fold the WM_ waves that share columns through LDS and issue full 256-byte atomic rows
#pragma unroll
      for (int c = 0; c < NP * EL; ++c) {
        float t = csum[c];
        // (ds_bpermute on the local lane id: __shfl_xor's own lane id is hoisted to kernel entry and kept live)
#pragma unroll
        for (int msk = 4; msk < 64; msk <<= 1)
          t += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ msk) << 2, __builtin_bit_cast(int, t)));
        csum[c] = t;
      }
      block_sync();                                      // every wave is done with its scratch
      float* cs_lds = reinterpret_cast<float*>(smem);    // [waves][TN*16]
      if (lane < 4) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int e = 0; e < EL; ++e) cs_lds[wave * (TN * 16) + q * PW + lcol + ecol(e)] = csum[q * EL + e];
      }
      block_sync();
      constexpr int BN_ = WN_ * TN * 16;
      for (int col = wave * 64 + lane; col < BN_; col += WM_ * WN_ * 64) {
        const int wcn = col / (TN * 16), cin = col - wcn * (TN * 16);
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < WM_; ++r) t += cs_lds[(r * WN_ + wcn) * (TN * 16) + cin];
        atomicAdd(p.colsum + n0 + col, t);
      }
    }
  }
}
