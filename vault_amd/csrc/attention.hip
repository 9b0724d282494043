// Multi-head self-attention forward/backward for short sequences (S <= 32*NKT keys: the fused
// [text | patch] sequence S = 185 -> NKT = 6, the text-only LM sequence S = 40 -> NKT = 2, padded batches of
// larger images (up to 384 x 640 -> S = 281) -> NKT = 10), d = 64.
//
// One workgroup per (batch, head).  The whole K and V of the head live in LDS (XOR-swizzled 128-byte
// rows); scores are computed transposed (S^T = K Q^T) so that the softmax'd tile is, register for
// register, the A operand of the following P.V MFMA - probabilities never leave registers.  V (and
// in backward K, Q, dO) are consumed k-strided through ds_read_b64_tr_b16 from their row-major images.
//
// Replaces ViltSelfAttention.forward (HF:models/vilt/modeling_vilt.py:322-351: eager softmax with
// an additive finfo.min key mask) and RobertaSelfAttention (HF:models/roberta/modeling_roberta.py:158-250)
// plus their autograd backward.  Masked keys get probability exactly 0 (the reference adds
// finfo.min, which underflows to the same 0 as long as one key is valid - always true: CLS/<s>).
#include "common.h"
#include "../../include/vault_hip.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int swz_row(int row) { return (row >> 1) & 3; }  // 32-byte block XOR key

// stage a [rows<=S][64] bf16 matrix (row stride ld elements) into a swizzled [SK][128 B] LDS image
template <int SK, int NT>
__device__ __forceinline__ void stage_rows(char* dst, const bf16* src, int ld, int S, int tid) {
  // all global loads are issued before the first LDS write: one memory round trip per call instead of
  // one per 16-byte chunk (a rolled load->store loop waits for every load separately)
  constexpr int N = (SK * 8) / NT;
  static_assert((SK * 8) % NT == 0, "chunk count must divide evenly");
  u32x4 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = tid + i * NT, row = c >> 3, pos = c & 7;
    v[i] = u32x4{0u, 0u, 0u, 0u};
    if (row < S) v[i] = *reinterpret_cast<const u32x4*>(src + (size_t)row * ld + pos * 8);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int c = tid + i * NT, row = c >> 3, pos = c & 7;
    *reinterpret_cast<u32x4*>(dst + row * 128 + ((pos ^ (swz_row(row) << 1)) << 4)) = v[i];
  }
}

// row-read fragment (A or B operand, k = d): rows tile*16 + l15, d = 32s + 8g .. +7
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int tile, int s, int g, int l15) {
  const int fx = swz_row(l15) << 1;
  return *LDS_PTR(const bf16x8, img + (tile * 16 + l15) * 128 + (((4 * s + g) ^ fx) << 4));
}

// transposed fragment (B operand, k = row index permuted as kappa(g,j) = 32T + 16(j>>2) + 4g + (j&3),
// col = d = dt*16 + l15) from a row-major [rows][64] image
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int T, int dt, int g, int l15) {
  const int qq = l15 >> 2, pp = l15 & 3;
  const int row = 32 * T + 4 * g + qq;
  const int x = (2 * (g & 1) + (qq >> 1)) & 3;  // == swz_row(row) and == swz_row(row + 16)
  const char* a = img + row * 128 + ((dt ^ x) << 5) + pp * 8;
  return cat_tr(lds_read_tr16(a), lds_read_tr16(a + 16 * 128));
}

__device__ __forceinline__ bf16x8 pack_frag(const f32x4& lo, const f32x4& hi) {
  bf16x8 f;
  f[0] = (bf16)lo[0]; f[1] = (bf16)lo[1]; f[2] = (bf16)lo[2]; f[3] = (bf16)lo[3];
  f[4] = (bf16)hi[0]; f[5] = (bf16)hi[1]; f[6] = (bf16)hi[2]; f[7] = (bf16)hi[3];
  return f;
}

struct AttnDrop {
  uint32_t thresh, seed, stream;
  float scale;
};

template <int NKT, int NWV, int WPE, bool DROP = true>
__global__ __launch_bounds__(NWV * 64, WPE) void attn_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ keymask,
                                                       bf16* __restrict__ ctx, float* __restrict__ lse, int S, int H,
                                                       int heads, float scale, AttnDrop dr, bf16* __restrict__ ctx3) {
  constexpr int SK = NKT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + SK * 128;
  float* mb = reinterpret_cast<float*>(smem + 2 * SK * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.x, b = blockIdx.y;
  const size_t row0 = (size_t)b * S;
  const int ld = 3 * H;
  const bf16* qbase = qkv + row0 * ld + h * 64;
  stage_rows<SK, NWV * 64>(Ks, qbase + H, ld, S, tid);
  stage_rows<SK, NWV * 64>(Vs, qbase + 2 * H, ld, S, tid);
  for (int k = tid; k < SK; k += NWV * 64)
    mb[k] = (k < S && (keymask == nullptr || keymask[(size_t)b * S + k] != 0.f)) ? 0.f : -INFINITY;
  __syncthreads();
  const float sl2 = scale * LOG2E;
  const int nqt = (S + 15) >> 4;
  const uint32_t bh = (uint32_t)(b * heads + h);
  for (int qt = wave; qt < nqt; qt += NWV) {
    const int qrow = min(qt * 16 + l15, S - 1);
    bf16x8 qf[2];
    qf[0] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrow * ld + 8 * g);
    qf[1] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrow * ld + 32 + 8 * g);
    // pass 1: row maximum only (scores are recomputed in pass 2: the matrix pipe is nearly idle in this
    // kernel, while keeping all 12 score tiles live costs 48 registers and the occupancy that hides LDS latency)
    // (the key-mask bias, 0 or -inf per key = accumulator row, is the MFMA's initial accumulator: no add)
    float mx = -INFINITY;
#pragma unroll 3
    for (int kt = 0; kt < 2 * NKT; ++kt) {
      f32x4 a = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, kt, 0, g, l15), qf[0], a, 0, 0, 0);
      a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, kt, 1, g, l15), qf[1], a, 0, 0, 0);
      mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxs = mx * sl2;
    const int q_l = qt * 16 + l15;
    // pass 2: p = exp2(s * sl2 - mxs), row sum, and O += P V with P straight from the accumulator layout
    float sum = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int T = 0; T < NKT; ++T) {
      f32x4 p2[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int kt = 2 * T + hh;
        f32x4 a = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, kt, 0, g, l15), qf[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, kt, 1, g, l15), qf[1], a, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(a[r] * sl2 - mxs);
          sum += pv;
          if (DROP && dr.thresh != 0u) {
            const uint32_t idx = (bh * (uint32_t)S + (uint32_t)q_l) * (uint32_t)S + (uint32_t)(kt * 16 + 4 * g + r);
            pv = dropout_keep(dr.seed, dr.stream, idx, dr.thresh) ? pv * dr.scale : 0.f;
          }
          p2[hh][r] = pv;
        }
      }
      // O^T[d][q] += V^T[d][key] P^T[key][q]: the transposed-read V fragment is the A operand (row = d), the
      // probabilities (accumulator layout: query on the lane) the B operand.  The result has the query on the
      // lane and four consecutive d in its registers: 8-byte stores and a lane-local normalisation.
      const bf16x8 pf = pack_frag(p2[0], p2[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(Vs, T, dt, g, l15), pf, o[dt], 0, 0, 0);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;       // this lane's query row (l15)
    if (q_l < S) {
      if (g == 0) lse[(size_t)bh * S + q_l] = mx * scale + __logf(sum);
      if (ctx3 != nullptr) {   // precise path: [hi | lo | hi] operand of the split-bf16 projection GEMM
        bf16* dst = ctx3 + (row0 + q_l) * 3 * H + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          bf16 hi[4], lo[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) split_bf16(o[dt][r] * inv, hi[r], lo[r]);
          const uint2 wh = {pack_bf16x2((float)hi[0], (float)hi[1]), pack_bf16x2((float)hi[2], (float)hi[3])};
          const uint2 wl = {pack_bf16x2((float)lo[0], (float)lo[1]), pack_bf16x2((float)lo[2], (float)lo[3])};
          *reinterpret_cast<uint2*>(dst + dt * 16) = wh;
          *reinterpret_cast<uint2*>(dst + H + dt * 16) = wl;
          *reinterpret_cast<uint2*>(dst + 2 * H + dt * 16) = wh;
        }
      } else {
        bf16* dst = ctx + (row0 + q_l) * H + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const uint2 w = {pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
          *reinterpret_cast<uint2*>(dst + dt * 16) = w;
        }
      }
    }
  }
}

// Backward.  Phase A (wave = query tile): dQ.  Phase B (wave = key tile): dK, dV.  P is recomputed
// from Q, K and the forward's log-sum-exp; both phases recompute the score tile in the orientation
// whose accumulator is directly the next MFMA's A operand.
template <int NKT, int NWV, int WPE, bool DROP = true>
__global__ __launch_bounds__(NWV * 64, WPE) void attn_bwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ keymask,
                                                       const bf16* __restrict__ ctx, const bf16* __restrict__ dctx,
                                                       const float* __restrict__ lse, bf16* __restrict__ dqkv, int S,
                                                       int H, int heads, float scale, AttnDrop dr) {
  constexpr int SK = NKT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* img0 = smem;               // phase A: K     phase B: Q
  char* img1 = smem + SK * 128;    // phase A: V     phase B: dO
  float* mb = reinterpret_cast<float*>(smem + 2 * SK * 128);
  float* lse_s = mb + SK;          // -lse * log2e ; -inf for q >= S
  float* dl_s = lse_s + SK;        // delta[q]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.x, b = blockIdx.y;
  const size_t row0 = (size_t)b * S;
  const int ld = 3 * H;
  const bf16* qbase = qkv + row0 * ld + h * 64;
  const bf16* obase = ctx + row0 * H + h * 64;
  const bf16* dobase = dctx + row0 * H + h * 64;
  bf16* dqbase = dqkv + row0 * ld + h * 64;
  const uint32_t bh = (uint32_t)(b * heads + h);
  const float sl2 = scale * LOG2E;

  stage_rows<SK, NWV * 64>(img0, qbase + H, ld, S, tid);
  stage_rows<SK, NWV * 64>(img1, qbase + 2 * H, ld, S, tid);
  for (int k = tid; k < SK; k += NWV * 64) {
    mb[k] = (k < S && (keymask == nullptr || keymask[(size_t)b * S + k] != 0.f)) ? 0.f : -INFINITY;
    lse_s[k] = (k < S) ? -lse[(size_t)bh * S + k] * LOG2E : -INFINITY;
  }
  // delta[q] = sum_d dO[q][d] * O[q][d] ; 4 lanes per row
  for (int q = tid >> 2; q < ((SK + NWV * 16 - 1) / (NWV * 16)) * (NWV * 16); q += NWV * 16) {
    float s = 0.f;
    if (q < S) {
      const int d0 = (tid & 3) * 16;
#pragma unroll
      for (int c = 0; c < 16; c += 8) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(obase + (size_t)q * H + d0 + c);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dobase + (size_t)q * H + d0 + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)d[e];
      }
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if ((tid & 3) == 0 && q < SK) dl_s[q] = s;
  }
  __syncthreads();

  const int nqt = (S + 15) >> 4;
  // ---------------- phase A: dQ ----------------
  for (int qt = wave; qt < nqt; qt += NWV) {
    const int q_l = qt * 16 + l15;
    const int qrow = min(q_l, S - 1);
    bf16x8 qf[2], df[2];
    qf[0] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrow * ld + 8 * g);
    qf[1] = *reinterpret_cast<const bf16x8*>(qbase + (size_t)qrow * ld + 32 + 8 * g);
    df[0] = *reinterpret_cast<const bf16x8*>(dobase + (size_t)qrow * H + 8 * g);
    df[1] = *reinterpret_cast<const bf16x8*>(dobase + (size_t)qrow * H + 32 + 8 * g);
    const float nl = lse_s[min(q_l, SK - 1)], dl = dl_s[min(q_l, SK - 1)];
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int T = 0; T < NKT; ++T) {
      f32x4 ds2[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int kt = 2 * T + hh;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(img0, kt, 0, g, l15), qf[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(img0, kt, 1, g, l15), qf[1], a, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(img1, kt, 0, g, l15), df[0], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(img1, kt, 1, g, l15), df[1], dp, 0, 0, 0);
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f((a[r] + m4[r]) * sl2 + nl);
          float dpv = dp[r];
          if (DROP && dr.thresh != 0u) {
            const uint32_t idx = (bh * (uint32_t)S + (uint32_t)q_l) * (uint32_t)S + (uint32_t)(kt * 16 + 4 * g + r);
            dpv = dropout_keep(dr.seed, dr.stream, idx, dr.thresh) ? dpv * dr.scale : 0.f;
          }
          ds2[hh][r] = pv * (dpv - dl) * scale;
        }
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q] (query stays on the lane: 8-byte stores)
      const bf16x8 dsf = pack_frag(ds2[0], ds2[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(img0, T, dt, g, l15), dsf, o[dt], 0, 0, 0);
    }
    if (q_l < S) {
      bf16* dst = dqbase + (size_t)q_l * ld + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint2 w = {pack_bf16x2(o[dt][0], o[dt][1]), pack_bf16x2(o[dt][2], o[dt][3])};
        *reinterpret_cast<uint2*>(dst + dt * 16) = w;
      }
    }
  }
  __syncthreads();
  // ---------------- phase B: dK, dV ----------------
  stage_rows<SK, NWV * 64>(img0, qbase, ld, S, tid);
  stage_rows<SK, NWV * 64>(img1, dobase, H, S, tid);
  __syncthreads();
  const int nkt = (S + 15) >> 4;
  for (int kt = wave; kt < nkt; kt += NWV) {
    const int k_l = kt * 16 + l15;
    const int krow = min(k_l, S - 1);
    bf16x8 kf[2], vf[2];
    kf[0] = *reinterpret_cast<const bf16x8*>(qbase + H + (size_t)krow * ld + 8 * g);
    kf[1] = *reinterpret_cast<const bf16x8*>(qbase + H + (size_t)krow * ld + 32 + 8 * g);
    vf[0] = *reinterpret_cast<const bf16x8*>(qbase + 2 * H + (size_t)krow * ld + 8 * g);
    vf[1] = *reinterpret_cast<const bf16x8*>(qbase + 2 * H + (size_t)krow * ld + 32 + 8 * g);
    const float mk = mb[min(k_l, SK - 1)];
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int T = 0; T < NKT; ++T) {
      f32x4 p2[2], ds2[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int qt = 2 * T + hh;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(img0, qt, 0, g, l15), kf[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(img0, qt, 1, g, l15), kf[1], a, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(img1, qt, 0, g, l15), vf[0], dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(img1, qt, 1, g, l15), vf[1], dp, 0, 0, 0);
        const f32x4 nl4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * g);
        const f32x4 dl4 = *reinterpret_cast<const f32x4*>(dl_s + qt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f((a[r] + mk) * sl2 + nl4[r]);
          float dpv = dp[r];
          if (DROP && dr.thresh != 0u) {
            const uint32_t idx = (bh * (uint32_t)S + (uint32_t)(qt * 16 + 4 * g + r)) * (uint32_t)S + (uint32_t)k_l;
            const bool keep = dropout_keep(dr.seed, dr.stream, idx, dr.thresh);
            dpv = keep ? dpv * dr.scale : 0.f;
            ds2[hh][r] = pv * (dpv - dl4[r]) * scale;
            pv = keep ? pv * dr.scale : 0.f;
          } else {
            ds2[hh][r] = pv * (dpv - dl4[r]) * scale;
          }
          p2[hh][r] = pv;
        }
      }
      const bf16x8 pf = pack_frag(p2[0], p2[1]);
      const bf16x8 dsf = pack_frag(ds2[0], ds2[1]);
      // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]  (key on the lane)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(img1, T, dt, g, l15), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(img0, T, dt, g, l15), dsf, dk[dt], 0, 0, 0);
      }
    }
    if (k_l < S) {
      bf16* dstk = dqbase + H + (size_t)k_l * ld + 4 * g;
      bf16* dstv = dqbase + 2 * H + (size_t)k_l * ld + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const uint2 wk = {pack_bf16x2(dk[dt][0], dk[dt][1]), pack_bf16x2(dk[dt][2], dk[dt][3])};
        const uint2 wv = {pack_bf16x2(dv[dt][0], dv[dt][1]), pack_bf16x2(dv[dt][2], dv[dt][3])};
        *reinterpret_cast<uint2*>(dstk + dt * 16) = wk;
        *reinterpret_cast<uint2*>(dstv + dt * 16) = wv;
      }
    }
  }
}

// Backward, resident form (64 < S <= 192): Q, K, V and dO of a (batch, head) are ALL held in LDS (4 x 24 KiB), so every
// operand is read from HBM exactly once (the two-phase kernel above reads Q, K, V twice and dO three times: 662 MB
// per launch measured against 364 MB algorithmic at B = 256), and nothing is re-staged between the dQ and the
// dK / dV passes (no barrier between them).  One workgroup of NWV waves per CU (a wave owns 2 NKT / NWV 16-row tiles
// in both passes and works on them together, sharing every fragment read) walks the (batch, head) items
// persistently; the operands of the NEXT item are fetched into registers (two 16-byte chunks per thread and matrix
// at 12 waves) before the current item's passes and written to LDS after them, so the loads of item i + 1 are in
// flight under the arithmetic of item i.  delta = rowsum(dO * O) comes from the same registers.  The item barriers
// wait for LDS only (`s_waitcnt lgkmcnt(0)`; `__syncthreads()` would also wait for the stores just issued).
// Measured at B = 256, S = 185 (tools/attn_bench.py, same box): 233 us against 296 us for the two-phase kernel;
// four waves x three tiles (512 registers per lane, a third of the LDS reads) 287 us: the loop is bound by the LDS
// round trips in front of the transposed-fragment MFMAs, which the compiler neither hoists (register cap 168 at
// three waves per SIMD) nor overlaps within one wave.
template <int NKT, int NWV, bool DROP = true>
__global__ __launch_bounds__(NWV * 64, 1) void attn_bwd_res_kernel(const bf16* __restrict__ qkv, const float* __restrict__ keymask,
                                                         const bf16* __restrict__ ctx, const bf16* __restrict__ dctx,
                                                         const float* __restrict__ lse, bf16* __restrict__ dqkv, int S,
                                                         int H, int heads, int items, float scale, AttnDrop dr) {
  constexpr int SK = NKT * 32;
  constexpr int NT = NWV * 64;
  constexpr int NC = (SK * 8) / NT;   // 16-byte chunks per thread and matrix
  constexpr int TPW = (2 * NKT) / NWV;   // 16-row tiles per wave, processed together
  static_assert((2 * NKT) % NWV == 0, "tiles must divide evenly over the waves");
  static_assert((SK * 8) % NT == 0 && NT % 8 == 0 && SK <= NT, "chunk / row bookkeeping");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* Ks = smem + SK * 128;
  char* Vs = smem + 2 * SK * 128;
  char* Ds = smem + 3 * SK * 128;   // dO
  float* mb = reinterpret_cast<float*>(smem + 4 * SK * 128);
  float* lse_s = mb + SK;          // -lse * log2e ; -inf for q >= S
  float* dl_s = lse_s + SK;        // delta[q]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int ld = 3 * H;
  const float sl2 = scale * LOG2E;

  u32x4 rq[NC], rk[NC], rv[NC], rd[NC], ro[NC];
  float rl = 0.f, rm = 0.f;
  auto fetch = [&](int item) {
    const int b = item / heads, h = item - b * heads;
    const size_t row0 = (size_t)b * S;
    const bf16* qb = qkv + row0 * ld + h * 64;
    const bf16* ob = ctx + row0 * H + h * 64;
    const bf16* db = dctx + row0 * H + h * 64;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = tid + i * NT, row = c >> 3, pos = c & 7;
      rq[i] = rk[i] = rv[i] = rd[i] = ro[i] = u32x4{0u, 0u, 0u, 0u};
      if (row < S) {
        const bf16* r = qb + (size_t)row * ld + pos * 8;
        rq[i] = *reinterpret_cast<const u32x4*>(r);
        rk[i] = *reinterpret_cast<const u32x4*>(r + H);
        rv[i] = *reinterpret_cast<const u32x4*>(r + 2 * H);
        rd[i] = *reinterpret_cast<const u32x4*>(db + (size_t)row * H + pos * 8);
        ro[i] = *reinterpret_cast<const u32x4*>(ob + (size_t)row * H + pos * 8);
      }
    }
    if (tid < SK) {
      rl = (tid < S) ? lse[(size_t)item * S + tid] : INFINITY;
      rm = (tid < S && keymask != nullptr) ? keymask[(size_t)b * S + tid] : 1.f;
    }
  };

  int item = blockIdx.x;
  if (item < items) fetch(item);
  for (; item < items; item += gridDim.x) {
    // ---- registers -> LDS images, delta, key bias, -lse ----
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = tid + i * NT, row = c >> 3, pos = c & 7;
      const int off = row * 128 + ((pos ^ (swz_row(row) << 1)) << 4);
      *reinterpret_cast<u32x4*>(Qs + off) = rq[i];
      *reinterpret_cast<u32x4*>(Ks + off) = rk[i];
      *reinterpret_cast<u32x4*>(Vs + off) = rv[i];
      *reinterpret_cast<u32x4*>(Ds + off) = rd[i];
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float2 a = unpack_bf16x2(ro[i][w]), d = unpack_bf16x2(rd[i][w]);
        s += a.x * d.x + a.y * d.y;
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (pos == 0) dl_s[row] = s;   // rows >= S hold zeros: delta = 0
    }
    if (tid < SK) {
      mb[tid] = (tid < S && rm != 0.f) ? 0.f : -INFINITY;
      lse_s[tid] = -rl * LOG2E;      // -inf for rows >= S
    }
    const int b = item / heads, h = item - b * heads;
    const uint32_t bh = (uint32_t)item;
    bf16* dqbase = dqkv + (size_t)b * S * ld + h * 64;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (no vmcnt wait: the previous item's stores stay in flight)
    if (item + (int)gridDim.x < items) fetch(item + gridDim.x);

    // ---------------- dQ: wave = TPW query tiles at once (they share every K / V fragment read) ----------------
    {
      bf16x8 qf[TPW][2], df[TPW][2];
      float nl[TPW], dl[TPW];
      f32x4 o[TPW][4];
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int qt = wave * TPW + j;
        qf[j][0] = frag_rows(Qs, qt, 0, g, l15); qf[j][1] = frag_rows(Qs, qt, 1, g, l15);
        df[j][0] = frag_rows(Ds, qt, 0, g, l15); df[j][1] = frag_rows(Ds, qt, 1, g, l15);
        nl[j] = lse_s[qt * 16 + l15]; dl[j] = dl_s[qt * 16 + l15];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll 1
      for (int T = 0; T < NKT; ++T) {
        f32x4 ds2[TPW][2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int kt = 2 * T + hh;
          const bf16x8 k0 = frag_rows(Ks, kt, 0, g, l15), k1 = frag_rows(Ks, kt, 1, g, l15);
          const bf16x8 v0 = frag_rows(Vs, kt, 0, g, l15), v1 = frag_rows(Vs, kt, 1, g, l15);
          const f32x4 m4 = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[j][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[j][1], a, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, df[j][0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, df[j][1], dp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = __builtin_amdgcn_exp2f((a[r] + m4[r]) * sl2 + nl[j]);
              float dpv = dp[r];
              if (DROP && dr.thresh != 0u) {
                const uint32_t q_l = (uint32_t)((wave * TPW + j) * 16 + l15);
                const uint32_t idx = (bh * (uint32_t)S + q_l) * (uint32_t)S + (uint32_t)(kt * 16 + 4 * g + r);
                dpv = dropout_keep(dr.seed, dr.stream, idx, dr.thresh) ? dpv * dr.scale : 0.f;
              }
              ds2[j][hh][r] = pv * (dpv - dl[j]) * scale;
            }
          }
        }
        bf16x8 dsf[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) dsf[j] = pack_frag(ds2[j][0], ds2[j][1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 kt_ = frag_tr(Ks, T, dt, g, l15);
#pragma unroll
          for (int j = 0; j < TPW; ++j) o[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_, dsf[j], o[j][dt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int q_l = (wave * TPW + j) * 16 + l15;
        if (q_l < S) {
          bf16* dst = dqbase + (size_t)q_l * ld + 4 * g;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const uint2 w = {pack_bf16x2(o[j][dt][0], o[j][dt][1]), pack_bf16x2(o[j][dt][2], o[j][dt][3])};
            *reinterpret_cast<uint2*>(dst + dt * 16) = w;
          }
        }
      }
    }
    // ---------------- dK, dV: wave = TPW key tiles at once (they share every Q / dO fragment read) ----------------
    {
      bf16x8 kf[TPW][2], vf[TPW][2];
      float mk[TPW];
      f32x4 dk[TPW][4], dv[TPW][4];
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int kt = wave * TPW + j;
        kf[j][0] = frag_rows(Ks, kt, 0, g, l15); kf[j][1] = frag_rows(Ks, kt, 1, g, l15);
        vf[j][0] = frag_rows(Vs, kt, 0, g, l15); vf[j][1] = frag_rows(Vs, kt, 1, g, l15);
        mk[j] = mb[kt * 16 + l15];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dk[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
          dv[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll 1
      for (int T = 0; T < NKT; ++T) {
        f32x4 p2[TPW][2], ds2[TPW][2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int qt = 2 * T + hh;
          const bf16x8 q0 = frag_rows(Qs, qt, 0, g, l15), q1 = frag_rows(Qs, qt, 1, g, l15);
          const bf16x8 d0 = frag_rows(Ds, qt, 0, g, l15), d1 = frag_rows(Ds, qt, 1, g, l15);
          const f32x4 nl4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * g);
          const f32x4 dl4 = *reinterpret_cast<const f32x4*>(dl_s + qt * 16 + 4 * g);
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, kf[j][0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, kf[j][1], a, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, vf[j][0], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1, vf[j][1], dp, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pv = __builtin_amdgcn_exp2f((a[r] + mk[j]) * sl2 + nl4[r]);
              float dpv = dp[r];
              if (DROP && dr.thresh != 0u) {
                const uint32_t k_l = (uint32_t)((wave * TPW + j) * 16 + l15);
                const uint32_t idx = (bh * (uint32_t)S + (uint32_t)(qt * 16 + 4 * g + r)) * (uint32_t)S + k_l;
                const bool keep = dropout_keep(dr.seed, dr.stream, idx, dr.thresh);
                dpv = keep ? dpv * dr.scale : 0.f;
                ds2[j][hh][r] = pv * (dpv - dl4[r]) * scale;
                pv = keep ? pv * dr.scale : 0.f;
              } else {
                ds2[j][hh][r] = pv * (dpv - dl4[r]) * scale;
              }
              p2[j][hh][r] = pv;
            }
          }
        }
        bf16x8 pf[TPW], dsf[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          pf[j] = pack_frag(p2[j][0], p2[j][1]);
          dsf[j] = pack_frag(ds2[j][0], ds2[j][1]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 dt_ = frag_tr(Ds, T, dt, g, l15), qt_ = frag_tr(Qs, T, dt, g, l15);
#pragma unroll
          for (int j = 0; j < TPW; ++j) {
            dv[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt_, pf[j], dv[j][dt], 0, 0, 0);
            dk[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_, dsf[j], dk[j][dt], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int k_l = (wave * TPW + j) * 16 + l15;
        if (k_l < S) {
          bf16* dstk = dqbase + H + (size_t)k_l * ld + 4 * g;
          bf16* dstv = dqbase + 2 * H + (size_t)k_l * ld + 4 * g;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const uint2 wk = {pack_bf16x2(dk[j][dt][0], dk[j][dt][1]), pack_bf16x2(dk[j][dt][2], dk[j][dt][3])};
            const uint2 wv = {pack_bf16x2(dv[j][dt][0], dv[j][dt][1]), pack_bf16x2(dv[j][dt][2], dv[j][dt][3])};
            *reinterpret_cast<uint2*>(dstk + dt * 16) = wk;
            *reinterpret_cast<uint2*>(dstv + dt * 16) = wv;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the images before the next item overwrites them
  }
}

template <int NKT>
constexpr int attn_res_lds_bytes() { return NKT * 32 * 128 * 4 + NKT * 32 * 4 * 3; }

template <int NKT>
constexpr int attn_lds_bytes() { return NKT * 32 * 128 * 2 + NKT * 32 * 4 * 3; }

}  // namespace

extern "C" int vault_attention_fwd(const vault_attn_args* a, void* stream) {
  if (!a || !a->qkv || (!a->ctx && !a->ctx_split3) || !a->lse || a->S <= 0 || a->B <= 0 || a->H != a->heads * 64)
    return VAULT_EINVAL;
  if (a->S > 320) return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const AttnDrop dr{a->drop_thresh, a->drop_seed, a->drop_stream, a->drop_scale};
  dim3 grid(a->heads, a->B), block(256);
  const float scale = 0.125f;  // 1/sqrt(64)
#define FWD_V(NK, NW, WP, DR)                                                                                          \
    hipLaunchKernelGGL((attn_fwd_kernel<NK, NW, WP, DR>), grid, dim3(NW * 64), attn_lds_bytes<NK>(), st,                \
                       reinterpret_cast<const bf16*>(a->qkv), a->keymask, reinterpret_cast<bf16*>(a->ctx), a->lse, a->S, \
                       a->H, a->heads, scale, dr, reinterpret_cast<bf16*>(a->ctx_split3))
  // (dropout is a template switch: a per-element run-time test splits the loop body into basic blocks that the
  //  instruction scheduler cannot move MFMAs and LDS reads across)
  const bool drop = a->drop_thresh != 0u;
  if (a->S <= 64) {
    if (drop) FWD_V(2, 4, 3, true); else FWD_V(2, 4, 3, false);
  } else if (a->S <= 192) {
    if (drop) FWD_V(6, 12, 6, true); else FWD_V(6, 12, 6, false);
  } else {   // long sequences of padded, larger images: K/V image 80 KiB -> one block per CU
    auto kern = attn_fwd_kernel<10, 4, 1>;
    static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         attn_lds_bytes<10>());
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), attn_lds_bytes<10>(), st, reinterpret_cast<const bf16*>(a->qkv),
                       a->keymask, reinterpret_cast<bf16*>(a->ctx), a->lse, a->S, a->H, a->heads, scale, dr,
                       reinterpret_cast<bf16*>(a->ctx_split3));
  }
#undef FWD_V
  return (int)hipGetLastError();
}

extern "C" int vault_attention_bwd(const vault_attn_args* a, void* stream) {
  if (!a || !a->qkv || !a->ctx || !a->lse || !a->dctx || !a->dqkv || a->S <= 0 || a->B <= 0 ||
      a->H != a->heads * 64)
    return VAULT_EINVAL;
  if (a->S > 320) return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const AttnDrop dr{a->drop_thresh, a->drop_seed, a->drop_stream, a->drop_scale};
  dim3 grid(a->heads, a->B), block(256);
  const float scale = 0.125f;
  const bool drop = a->drop_thresh != 0u;
#define OLD_V(NK, DR)                                                                                                         \
    hipLaunchKernelGGL((attn_bwd_kernel<NK, 4, 3, DR>), grid, dim3(256), attn_lds_bytes<NK>(), st, reinterpret_cast<const bf16*>(a->qkv), \
                       a->keymask, reinterpret_cast<const bf16*>(a->ctx), reinterpret_cast<const bf16*>(a->dctx), a->lse,     \
                       reinterpret_cast<bf16*>(a->dqkv), a->S, a->H, a->heads, scale, dr)
#define RES_V(DR)                                                                                                             \
    {                                                                                                                         \
      auto kern = attn_bwd_res_kernel<6, 12, DR>;   /* 98.3 KiB of LDS: one 12-wave workgroup per CU, persistent */           \
      static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];                                                                                          \
      if (!attr_done) {                                                                                                       \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                           attn_res_lds_bytes<6>());                                                          \
        if (e != hipSuccess) return (int)e;                                                                                   \
        attr_done = true;                                                                                                     \
      }                                                                                                                       \
      hipLaunchKernelGGL(kern, dim3(items < 256 ? items : 256), dim3(768), attn_res_lds_bytes<6>(), st,                       \
                         reinterpret_cast<const bf16*>(a->qkv), a->keymask, reinterpret_cast<const bf16*>(a->ctx),            \
                         reinterpret_cast<const bf16*>(a->dctx), a->lse, reinterpret_cast<bf16*>(a->dqkv), a->S, a->H,        \
                         a->heads, items, scale, dr);                                                                         \
    }
  if (a->S <= 64) {
    if (drop) OLD_V(2, true); else OLD_V(2, false);
  } else if (a->S <= 192) {
    const int items = a->B * a->heads;
    if (drop) RES_V(true) else RES_V(false)
  } else {
    auto kern = attn_bwd_kernel<10, 4, 1>;
    static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         attn_lds_bytes<10>());
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), attn_lds_bytes<10>(), st, reinterpret_cast<const bf16*>(a->qkv),
                       a->keymask, reinterpret_cast<const bf16*>(a->ctx), reinterpret_cast<const bf16*>(a->dctx), a->lse,
                       reinterpret_cast<bf16*>(a->dqkv), a->S, a->H, a->heads, scale, dr);
  }
#undef OLD_V
#undef RES_V
  return (int)hipGetLastError();
}
