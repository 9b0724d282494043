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
// Waves of the one-workgroup-per-CU kernels for 192 < S <= 320 (80 KiB K / V image; 320 * 8 chunks / (64 waves) must be whole).
// tools/attn_bench.py 64 281, same box: backward 252.6 us with 4 waves, 153.4 with 8, 134.3 with 10 (18 query / key tiles per
// item: 4 waves walk them in 5 rounds); the forward is fastest with 4 (106.8 us; 116.6 with 8, 158 with 10).
// 192 < S <= 288: waves per workgroup / launch-bounds waves per SIMD (2304 chunks / (64 waves) must be whole)
// (tools/attn_bench.py 64 281, same box: forward 73.5 us with 4 waves, 78 with 6, 50.2 with 12; backward 157 us with 6 waves,
//  133 with 9, 121 with 12 - against 106.8 / 134.3 us for the 320-row kernels below, one block per CU)
#ifndef ATTN_L9_FWD_W
#define ATTN_L9_FWD_W 12
#define ATTN_L9_FWD_WPE 6     // (two 12-wave blocks per CU: 46.9 against 50.3 us with one)
#endif
#ifndef ATTN_L9_BWD_W
#define ATTN_L9_BWD_W 12
#define ATTN_L9_BWD_WPE 3
#endif
#ifndef ATTN_LONG_WAVES_FWD
#define ATTN_LONG_WAVES_FWD 4
#endif
#ifndef ATTN_LONG_WAVES_BWD
#define ATTN_LONG_WAVES_BWD 10
#endif
#ifndef ATTN_FWD_WL
#define ATTN_FWD_WL 1     // forward: permuted d rows (16-byte pieces) + whole-line stores (0: 8-byte pieces, development A/B)
#endif
#ifndef ATTN_WL
#define ATTN_WL 1         // single-pass backward: whole-line output stores (0: the half-line form, development A/B)
#endif
#ifndef ATTN_ABLATE
#define ATTN_ABLATE 0     // development (resident backward kernels): 1 = memory traffic only, 2 = arithmetic only (every block on item 0..heads-1)
#endif

__device__ __forceinline__ int swz_row(int row) { return (row >> 1) & 3; }  // 32-byte block XOR key

// Layout of qkv / dqkv.  Row-major (hm_rows == 0): [tokens][3H], q | k | v, head h at columns 64 h: a (batch, head) item's rows
// are 128-byte segments at a 6 H byte stride.  Head-major (hm_rows = padded token rows): [3][heads][hm_rows][64] - an item's S
// rows are contiguous, its stores and loads stream (vault_attn_args.qkv_hm; the QKV GEMM's epilogue writes it, the QKV data /
// weight gradient GEMMs read it: vault_gemm_args.out_hm / a_hm).
struct QkvLayout {
  int ld;        // elements between consecutive token rows
  int hs;        // elements between consecutive heads
  int pl;        // elements between q, k and v of one head
  __device__ __forceinline__ QkvLayout(int H, int heads, int hm_rows)
      : ld(hm_rows ? 64 : 3 * H), hs(hm_rows ? hm_rows * 64 : 64), pl(hm_rows ? heads * hm_rows * 64 : H) {}
};

// stage a [rows<=S][64] bf16 matrix (row stride ld elements) into a swizzled [SK][128 B] LDS image
template <int SK, int NT>
__device__ __forceinline__ void stage_rows(char* dst, const h16* src, int ld, int S, int tid) {
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
__device__ __forceinline__ h16x8 frag_rows(const char* img, int tile, int s, int g, int l15) {
  const int fx = swz_row(l15) << 1;
  return *LDS_PTR(const h16x8, img + (tile * 16 + l15) * 128 + (((4 * s + g) ^ fx) << 4));
}

// transposed fragment (B operand, k = row index permuted as kappa(g,j) = 32T + 16(j>>2) + 4g + (j&3),
// col = d = dt*16 + l15) from a row-major [rows][64] image
__device__ __forceinline__ h16x8 frag_tr(const char* img, int T, int dt, int g, int l15) {
  const int qq = l15 >> 2, pp = l15 & 3;
  const int row = 32 * T + 4 * g + qq;
  const int x = (2 * (g & 1) + (qq >> 1)) & 3;  // == swz_row(row) and == swz_row(row + 16)
  const char* a = img + row * 128 + ((dt ^ x) << 5) + pp * 8;
  return cat_tr(lds_read_tr16(a), lds_read_tr16(a + 16 * 128));
}

// transposed fragment with the d columns PERMUTED so that an output lane ends up with 8 consecutive d: MFMA tile
// t = 2 half + odd takes d = 32 half + 8 (i >> 2) + 4 odd + (i & 3) for its row i.  The accumulator lane (g, l15) of tiles
// (2 half, 2 half + 1) then holds d = 32 half + 8 g + 0..7 of row l15: one 16-byte store instead of two 8-byte ones, 64
// contiguous bytes per row and instruction (the output rows are 128 bytes per head).  Costs a 2-way bank conflict on
// these reads (two of the four rows of a lane group share a 64-byte half of the swizzled image).
__device__ __forceinline__ h16x8 frag_tr8(const char* img, int T, int half, int odd, int g, int l15) {
  const int qq = l15 >> 2, pp = l15 & 3;
  const int row = 32 * T + 4 * g + qq;
  const int x = (2 * (g & 1) + (qq >> 1)) & 3;  // == swz_row(row) and == swz_row(row + 16)
  const char* a = img + row * 128 + (((4 * half + pp) ^ (x << 1)) << 4) + odd * 8;
  return cat_tr(lds_read_tr16(a), lds_read_tr16(a + 16 * 128));
}

__device__ __forceinline__ h16x8 pack_frag(const f32x4& lo, const f32x4& hi) {
  h16x8 f;
  f[0] = (h16)lo[0]; f[1] = (h16)lo[1]; f[2] = (h16)lo[2]; f[3] = (h16)lo[3];
  f[4] = (h16)hi[0]; f[5] = (h16)hi[1]; f[6] = (h16)hi[2]; f[7] = (h16)hi[3];
  return f;
}

// Output tile of 16 rows x 64 d (one head's 128-byte slice per row) whose accumulators came from frag_tr8 operands: tiles
// (2 hf, 2 hf + 1) of lane (g, l15) hold d = 32 hf + 8 g + 0..7 of row l15, i.e. the 16-byte chunks g and 4 + g of the row.
// Lanes l15 and l15 ^ 8 swap one chunk each (DPP row_ror:8) so that the two store instructions write rows 0-7 and rows 8-15
// of the tile as whole 128-byte lines.  dst: element (row 0, first column of the head); rows >= S are not stored.
__device__ __forceinline__ void store_tile_lines(h16* dst, int ld, int tile_row0, int S, const f32x4 (&o)[4], float mul,
                                                 int g, int l15) {
  u32x4 w[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
    w[hf] = u32x4{pack_h16x2(o[2 * hf][0] * mul, o[2 * hf][1] * mul), pack_h16x2(o[2 * hf][2] * mul, o[2 * hf][3] * mul),
                  pack_h16x2(o[2 * hf + 1][0] * mul, o[2 * hf + 1][1] * mul), pack_h16x2(o[2 * hf + 1][2] * mul, o[2 * hf + 1][3] * mul)};
  const bool lo8 = l15 < 8;
  u32x4 wa, wb;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t xr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[0][k], 0x128, 0xF, 0xF, true);   // row_ror:8
    const uint32_t yr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[1][k], 0x128, 0xF, 0xF, true);
    wa[k] = lo8 ? w[0][k] : yr;
    wb[k] = lo8 ? xr : w[1][k];
  }
  const int ra = tile_row0 + (l15 & 7);
  h16* d = dst + (size_t)ra * ld + (l15 >> 3) * 32 + 8 * g;
  if (ra < S) *reinterpret_cast<u32x4*>(d) = wa;
  if (ra + 8 < S) *reinterpret_cast<u32x4*>(d + 8 * (size_t)ld) = wb;
}

struct AttnDrop {
  uint32_t thresh, seed, stream;
  float scale;
};

template <int NKT, int NWV, int WPE, bool DROP = true>
__global__ __launch_bounds__(NWV * 64, WPE) void attn_fwd_kernel(const h16* __restrict__ qkv, const float* __restrict__ keymask,
                                                       h16* __restrict__ ctx, float* __restrict__ lse, int S, int H,
                                                       int heads, float scale, AttnDrop dr, h16* __restrict__ ctx3, int hm_rows) {
  H16_SATURATE();
  constexpr int SK = NKT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + SK * 128;
  float* mb = reinterpret_cast<float*>(smem + 2 * SK * 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.x, b = blockIdx.y;
  const size_t row0 = (size_t)b * S;
  const QkvLayout lay(H, heads, hm_rows);
  const int ld = lay.ld;
  const h16* qbase = qkv + row0 * ld + (size_t)h * lay.hs;
  // the query fragments of this wave's first tile are requested BEFORE the K / V staging: their round trip runs under it
  // (behind the staging barrier it would be a second exposed memory latency per workgroup)
  h16x8 qf[2];
  {
    const int qrow = min(wave * 16 + l15, S - 1);
    qf[0] = *reinterpret_cast<const h16x8*>(qbase + (size_t)qrow * ld + 8 * g);
    qf[1] = *reinterpret_cast<const h16x8*>(qbase + (size_t)qrow * ld + 32 + 8 * g);
  }
  stage_rows<SK, NWV * 64>(Ks, qbase + lay.pl, ld, S, tid);
  stage_rows<SK, NWV * 64>(Vs, qbase + 2 * (size_t)lay.pl, ld, S, tid);
  for (int k = tid; k < SK; k += NWV * 64)
    mb[k] = (k < S && (keymask == nullptr || keymask[(size_t)b * S + k] != 0.f)) ? 0.f : -INFINITY;
  __syncthreads();
  const float sl2 = scale * LOG2E;
  const int nqt = (S + 15) >> 4;
  const uint32_t bh = (uint32_t)(b * heads + h);
  for (int qt = wave; qt < nqt; qt += NWV) {
    if (qt != wave) {
      const int qrow = min(qt * 16 + l15, S - 1);
      qf[0] = *reinterpret_cast<const h16x8*>(qbase + (size_t)qrow * ld + 8 * g);
      qf[1] = *reinterpret_cast<const h16x8*>(qbase + (size_t)qrow * ld + 32 + 8 * g);
    }
    // pass 1: row maximum only (scores are recomputed in pass 2: the matrix pipe is nearly idle in this
    // kernel, while keeping all 12 score tiles live costs 48 registers and the occupancy that hides LDS latency)
    // (the key-mask bias, 0 or -inf per key = accumulator row, is the MFMA's initial accumulator: no add)
    float mx = -INFINITY;
#pragma unroll 3
    for (int kt = 0; kt < 2 * NKT; ++kt) {
      f32x4 a = *reinterpret_cast<const f32x4*>(mb + kt * 16 + 4 * g);
      a = mfma16(frag_rows(Ks, kt, 0, g, l15), qf[0], a);
      a = mfma16(frag_rows(Ks, kt, 1, g, l15), qf[1], a);
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
        a = mfma16(frag_rows(Ks, kt, 0, g, l15), qf[0], a);
        a = mfma16(frag_rows(Ks, kt, 1, g, l15), qf[1], a);
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
      // lane (lane-local normalisation); with the d rows of the A operand permuted (frag_tr8) the tiles (2 hf, 2 hf + 1)
      // hold d = 32 hf + 8 g + 0..7 of the lane's row: 16-byte pieces, written as whole 128-byte lines (below).
      const h16x8 pf = pack_frag(p2[0], p2[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#if ATTN_FWD_WL
        o[dt] = mfma16(frag_tr8(Vs, T, dt >> 1, dt & 1, g, l15), pf, o[dt]);
#else
        o[dt] = mfma16(frag_tr(Vs, T, dt, g, l15), pf, o[dt]);
#endif
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;       // this lane's query row (l15)
#if ATTN_FWD_WL
    if (q_l < S && g == 0) lse[(size_t)bh * S + q_l] = mx * scale + __logf(sum);
    if (ctx3 != nullptr) {   // precise path: [hi | lo | hi] operand of the split-bf16 projection GEMM (16-byte pieces)
      if (q_l < S) {
        h16* dst = ctx3 + (row0 + q_l) * 3 * H + h * 64 + 8 * g;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          uint32_t wh[4], wl[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            h16 h0, l0, h1, l1;
            split_bf16(o[2 * hf + (k >> 1)][2 * (k & 1)] * inv, h0, l0);
            split_bf16(o[2 * hf + (k >> 1)][2 * (k & 1) + 1] * inv, h1, l1);
            wh[k] = pack_h16x2((float)h0, (float)h1);
            wl[k] = pack_h16x2((float)l0, (float)l1);
          }
          const u32x4 vh = {wh[0], wh[1], wh[2], wh[3]}, vl = {wl[0], wl[1], wl[2], wl[3]};
          *reinterpret_cast<u32x4*>(dst + 32 * hf) = vh;
          *reinterpret_cast<u32x4*>(dst + H + 32 * hf) = vl;
          *reinterpret_cast<u32x4*>(dst + 2 * H + 32 * hf) = vh;
        }
      }
    } else {
      // whole-line stores: lanes l15 and l15 ^ 8 swap one 16-byte chunk each (DPP row_ror:8, see the single-pass backward):
      // instruction A writes rows 0-7 of the tile, instruction B rows 8-15, 128 contiguous bytes per row
      u32x4 w[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
        w[hf] = u32x4{pack_h16x2(o[2 * hf][0] * inv, o[2 * hf][1] * inv), pack_h16x2(o[2 * hf][2] * inv, o[2 * hf][3] * inv),
                      pack_h16x2(o[2 * hf + 1][0] * inv, o[2 * hf + 1][1] * inv), pack_h16x2(o[2 * hf + 1][2] * inv, o[2 * hf + 1][3] * inv)};
      const bool lo8 = l15 < 8;
      u32x4 wa, wb;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t xr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[0][k], 0x128, 0xF, 0xF, true);   // row_ror:8
        const uint32_t yr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[1][k], 0x128, 0xF, 0xF, true);
        wa[k] = lo8 ? w[0][k] : yr;
        wb[k] = lo8 ? xr : w[1][k];
      }
      const int ra = qt * 16 + (l15 & 7);
      h16* dst = ctx + (row0 + ra) * H + h * 64 + (l15 >> 3) * 32 + 8 * g;
      if (ra < S) *reinterpret_cast<u32x4*>(dst) = wa;
      if (ra + 8 < S) *reinterpret_cast<u32x4*>(dst + 8 * (size_t)H) = wb;
    }
#else
    if (q_l < S) {
      if (g == 0) lse[(size_t)bh * S + q_l] = mx * scale + __logf(sum);
      if (ctx3 != nullptr) {   // precise path: [hi | lo | hi] operand of the split-bf16 projection GEMM
        h16* dst = ctx3 + (row0 + q_l) * 3 * H + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          h16 hi[4], lo[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) split_bf16(o[dt][r] * inv, hi[r], lo[r]);
          const uint2 wh = {pack_h16x2((float)hi[0], (float)hi[1]), pack_h16x2((float)hi[2], (float)hi[3])};
          const uint2 wl = {pack_h16x2((float)lo[0], (float)lo[1]), pack_h16x2((float)lo[2], (float)lo[3])};
          *reinterpret_cast<uint2*>(dst + dt * 16) = wh;
          *reinterpret_cast<uint2*>(dst + H + dt * 16) = wl;
          *reinterpret_cast<uint2*>(dst + 2 * H + dt * 16) = wh;
        }
      } else {
        h16* dst = ctx + (row0 + q_l) * H + h * 64 + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const uint2 w = {pack_h16x2(o[dt][0] * inv, o[dt][1] * inv), pack_h16x2(o[dt][2] * inv, o[dt][3] * inv)};
          *reinterpret_cast<uint2*>(dst + dt * 16) = w;
        }
      }
    }
#endif
  }
}

// Backward.  Phase A (wave = query tile): dQ.  Phase B (wave = key tile): dK, dV.  P is recomputed
// from Q, K and the forward's log-sum-exp; both phases recompute the score tile in the orientation
// whose accumulator is directly the next MFMA's A operand.
template <int NKT, int NWV, int WPE, bool DROP = true>
__global__ __launch_bounds__(NWV * 64, WPE) void attn_bwd_kernel(const h16* __restrict__ qkv, const float* __restrict__ keymask,
                                                       const h16* __restrict__ ctx, const h16* __restrict__ dctx,
                                                       const float* __restrict__ lse, h16* __restrict__ dqkv, int S,
                                                       int H, int heads, float scale, AttnDrop dr) {
  H16_SATURATE();
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
  const h16* qbase = qkv + row0 * ld + h * 64;
  const h16* obase = ctx + row0 * H + h * 64;
  const h16* dobase = dctx + row0 * H + h * 64;
  h16* dqbase = dqkv + row0 * ld + h * 64;
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
        const h16x8 a = *reinterpret_cast<const h16x8*>(obase + (size_t)q * H + d0 + c);
        const h16x8 d = *reinterpret_cast<const h16x8*>(dobase + (size_t)q * H + d0 + c);
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
    h16x8 qf[2], df[2];
    qf[0] = *reinterpret_cast<const h16x8*>(qbase + (size_t)qrow * ld + 8 * g);
    qf[1] = *reinterpret_cast<const h16x8*>(qbase + (size_t)qrow * ld + 32 + 8 * g);
    df[0] = *reinterpret_cast<const h16x8*>(dobase + (size_t)qrow * H + 8 * g);
    df[1] = *reinterpret_cast<const h16x8*>(dobase + (size_t)qrow * H + 32 + 8 * g);
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
        a = mfma16(frag_rows(img0, kt, 0, g, l15), qf[0], a);
        a = mfma16(frag_rows(img0, kt, 1, g, l15), qf[1], a);
        dp = mfma16(frag_rows(img1, kt, 0, g, l15), df[0], dp);
        dp = mfma16(frag_rows(img1, kt, 1, g, l15), df[1], dp);
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
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q] (query stays on the lane; d rows permuted: 16-byte chunks, whole-line stores)
      const h16x8 dsf = pack_frag(ds2[0], ds2[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = mfma16(frag_tr8(img0, T, dt >> 1, dt & 1, g, l15), dsf, o[dt]);
    }
    store_tile_lines(dqbase, ld, qt * 16, S, o, 1.0f, g, l15);
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
    h16x8 kf[2], vf[2];
    kf[0] = *reinterpret_cast<const h16x8*>(qbase + H + (size_t)krow * ld + 8 * g);
    kf[1] = *reinterpret_cast<const h16x8*>(qbase + H + (size_t)krow * ld + 32 + 8 * g);
    vf[0] = *reinterpret_cast<const h16x8*>(qbase + 2 * H + (size_t)krow * ld + 8 * g);
    vf[1] = *reinterpret_cast<const h16x8*>(qbase + 2 * H + (size_t)krow * ld + 32 + 8 * g);
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
        a = mfma16(frag_rows(img0, qt, 0, g, l15), kf[0], a);
        a = mfma16(frag_rows(img0, qt, 1, g, l15), kf[1], a);
        dp = mfma16(frag_rows(img1, qt, 0, g, l15), vf[0], dp);
        dp = mfma16(frag_rows(img1, qt, 1, g, l15), vf[1], dp);
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
      const h16x8 pf = pack_frag(p2[0], p2[1]);
      const h16x8 dsf = pack_frag(ds2[0], ds2[1]);
      // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]  (key on the lane)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = mfma16(frag_tr8(img1, T, dt >> 1, dt & 1, g, l15), pf, dv[dt]);
        dk[dt] = mfma16(frag_tr8(img0, T, dt >> 1, dt & 1, g, l15), dsf, dk[dt]);
      }
    }
    store_tile_lines(dqbase + H, ld, kt * 16, S, dk, 1.0f, g, l15);
    store_tile_lines(dqbase + 2 * H, ld, kt * 16, S, dv, 1.0f, g, l15);
  }
}

// Backward, single-pass resident form (round 3; S <= 192).  The two-pass resident kernel it replaced (deleted in round 5; a
// (batch, head) item's Q, K, V, dO all in LDS, one pass per orientation) recomputed the scores, the
// probabilities and dP twice and was bound by its own instruction stream: with the HBM traffic of a
// launch served from L2 it still takes 180 us at B = 256, with the arithmetic removed 143 us (tools/attn_bench.py on the
// ATTN_ABLATE builds) - the exp / dS arithmetic of BOTH passes saturates the vector issue of a SIMD (3 waves x ~100 VALU
// per 32 keys against 28 MFMAs).  Here every score is computed ONCE:
//   phase 1 (wave = one 16-key tile): S^T, P, dP, dS for all queries; dK, dV accumulate in registers; dS goes to LDS as a
//            row-major [key][query] bf16 image (416-byte rows: conflict-free for the 8-byte stores and the transposed reads);
//   phase 2 (wave = one 16-query tile): dQ = dS K from that image and the K image, both read k-strided
//            (ds_read_b64_tr_b16): 4 MFMAs per 32 keys and no vector arithmetic.
// V never enters LDS (a wave needs only its own 16 rows, as MFMA fragments: loaded straight into registers with the
// prefetch of the next item); Q, K, dO images + dS + row statistics = 152 KiB.  The mask is the initial accumulator of the
// score MFMA (0 / -inf per key), the 1/sqrt(d) factor is applied to dQ / dK once per output element.  The operands of the
// next item are prefetched into registers under the current item's arithmetic; its Q / dO images and statistics are
// written at the start of phase 2 (phase 2 reads only K and dS), its K image after phase 2.
#ifndef ATTN_ST_NT
#define ATTN_ST_NT 0      // development: 1 = non-temporal output stores
#endif
#ifndef ATTN_BWD_KT
#define ATTN_BWD_KT 1     // 16-row tiles per wave of the S <= 192 backward: 1 = 12 waves, 2 = 6 waves, 3 = 4 waves (round 6: measured
                          // slower, profiles/r06_dev_attn_bwd_tiles_per_wave.txt - phase 1 is bound by vector issue and latency, not by LDS bytes)
#endif
#ifndef ATTN_P1_UNROLL
#define ATTN_P1_UNROLL 1  // development: unroll factor of the phase-1 loop over the 32-query steps
#endif
#define ATTN_STR_(X) #X
#define ATTN_STR(X) ATTN_STR_(X)
#if ATTN_WL && defined(ATTN_LOADS_FIRST) && ATTN_LOADS_FIRST
#error "ATTN_LOADS_FIRST counts the half-line form's stores: build it with -DATTN_WL=0"
#endif
#ifndef ATTN_LOADS_FIRST
#define ATTN_LOADS_FIRST 0   // development: 1 = the next items' loads are issued in front of the dK / dV stores
#endif
#if ATTN_ST_NT
#define ATTN_STORE16(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<u32x4*>(ptr))
#else
#define ATTN_STORE16(ptr, val) (*reinterpret_cast<u32x4*>(ptr) = (val))
#endif
// bytes per row of the dS^T image: SK queries x 2 B + 32 of padding - 104 dwords at SK = 192, 40 at SK = 64, both = 40 mod 64:
// the 8 rows a transposed read touches per cycle then start 40 r mod 64 = 0, 40, 16, 56, 32, 8, 48, 24 dwords apart
// (8 disjoint runs of 8 banks), and the 8-byte stores of 16 consecutive keys are at most 2-way
template <int SK> constexpr int ds_ld() { return SK * 2 + 32; }

// Global loads and LDS-DMA of the single-pass kernel through inline asm: scalar base + 32-bit lane offset (no 64-bit
// address registers: the kernel lives at the 168-register edge of three waves per SIMD), and - the point - hipcc keeps no
// record of them, so it never answers an unrelated use with `s_waitcnt vmcnt(0)`: `vmcnt` counts loads, stores and DMA
// in issue order, and a conservative wait in the wrong place serialises an item's memory traffic with its arithmetic
// (the two-pass kernel: scratch reloads and exec-masked loads make hipcc wait for the whole prefetch right after
// issuing it).  Every wait on these loads is explicit and counted below.
__device__ __forceinline__ void attn_glds16(const char* sbase, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void attn_gload16(u32x4& dst, const char* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void attn_gload4(float& dst, const char* sbase, uint32_t voff) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// Column sums of a 16-row x 64-column accumulator tile (t[dt][r] of lane (g, l15): row l15, column 32 (dt >> 1) + 8 g +
// 4 (dt & 1) + r - the d order of frag_tr8) over the wave's 16 rows, added to 64 floats in LDS: the rows of a lane group are
// the 16 lanes of a DPP row (no LDS traffic for the reduction), lane e of a group then owns value e, and ONE LDS float add
// per wave writes 64 distinct addresses.  The QKV bias gradient (column sums of dq / dk / dv over the tokens) without a pass
// over dqkv: vault_attn_args.bias_partials.
__device__ __forceinline__ void attn_tile_colsum(const f32x4 (&t)[4], float mul, float* dst64, int g, int l15) {
  // transpose-reduce over the 16 lanes of a DPP row: at the step of lane bit b a lane keeps the values whose index has its own
  // bit b and adds the partner's share of them (partner = the lane that differs in bit b and agrees in the bits already done:
  // quad_perm for bits 0 / 1, a rotation by 4 / 8 for bits 2 / 3) - 15 DPP adds + 30 selects instead of 16 four-step trees;
  // lane e ends with the total of value e = 4 dt + r
#define ATTN_DPP(X, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, X), CTRL, 0xF, 0xF, true))
  float a[8], b[4], c[2];
  const bool b0 = l15 & 1, b1 = l15 & 2, b2 = l15 & 4, b3 = l15 & 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float lo = t[j >> 1][2 * (j & 1)], hi = t[j >> 1][2 * (j & 1) + 1];
    const float keep = b0 ? hi : lo, snd = b0 ? lo : hi;
    a[j] = keep + ATTN_DPP(snd, 0xB1);      // quad_perm [1,0,3,2]
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float keep = b1 ? a[2 * j + 1] : a[2 * j], snd = b1 ? a[2 * j] : a[2 * j + 1];
    b[j] = keep + ATTN_DPP(snd, 0x4E);      // quad_perm [2,3,0,1]
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float keep = b2 ? b[2 * j + 1] : b[2 * j], snd = b2 ? b[2 * j] : b[2 * j + 1];
    c[j] = keep + ATTN_DPP(snd, 0x124);     // row_ror:4
  }
  const float keep = b3 ? c[1] : c[0], snd = b3 ? c[0] : c[1];
  const float sel = keep + ATTN_DPP(snd, 0x128);   // row_ror:8
#undef ATTN_DPP
  const int d = 32 * (l15 >> 3) + 8 * g + 4 * ((l15 >> 2) & 1) + (l15 & 3);
  __builtin_amdgcn_ds_faddf(LDS_PTR(float, dst64 + d), sel * mul, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
}

// a pointer that is uniform by construction, pinned to scalar registers (asm "s" operands)
__device__ __forceinline__ const char* attn_uniform(const void* p) {
  const uint64_t u = (uint64_t)p;
  return reinterpret_cast<const char*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u));
}

// `s_waitcnt vmcnt(N)` for a compile-time N
template <int N>
__device__ __forceinline__ void attn_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KT (round 6) = 16-row tiles per wave: wave w owns key tiles w KT .. w KT + KT - 1 in phase 1 and the same query tiles in phase 2.
// Phase 1 is bound by LDS reads (profiles/r06_dev_attn_bwd_ablation.txt: 65-70 us of the launch, nothing hidden behind the
// memory pipeline): every wave reads ALL of Q and dO (row fragments for S^T / dP, transposed ones for dV / dK) whatever keys it
// owns - 12 waves of one tile read them 12 times, 6 waves of two tiles 6 times, each fragment feeding two MFMAs from registers.
// Every output element is computed by the same sequence of operations as with KT = 1 (bit-identical dq / dk / dv).
template <int NKT, int NWV, bool DROP = true, int WPE = 1, int KT = 1>
__global__ __launch_bounds__(NWV * 64, WPE) void attn_bwd_one_kernel(const h16* __restrict__ qkv, const float* __restrict__ keymask,
                                                         const h16* __restrict__ ctx, const h16* __restrict__ dctx,
                                                         const float* __restrict__ lse, h16* __restrict__ dqkv, int S,
                                                         int H, int heads, int items, float scale, AttnDrop dr, int hm_rows,
                                                         float* __restrict__ cs_out, int cs_thirds) {
  H16_SATURATE();
  constexpr int SK = NKT * 32;
  constexpr int DS_LD = ds_ld<SK>();
  constexpr int NT = NWV * 64;
  constexpr int NC = (SK * 8) / NT;   // 16-byte chunks per thread and matrix
  constexpr int LQ = 3 * NC + 2;      // loads of fetch_q_do per wave
  static_assert(2 * NKT == NWV * KT, "KT 16-row tiles per wave");
  static_assert((SK * 8) % NT == 0 && SK <= NT && NC >= 2 && NC <= 6 && KT >= 1 && KT <= 3, "chunk / row bookkeeping (NC 1 KiB DMA pieces per wave)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* Ks = smem + SK * 128;
  char* Ds = smem + 2 * SK * 128;   // dO
  char* dSs = smem + 3 * SK * 128;  // dS^T [key][query], DS_LD bytes per row
  float* mb = reinterpret_cast<float*>(smem + 3 * SK * 128 + SK * DS_LD);
  float* lse_s = mb + SK;          // -lse * log2e ; -inf for q >= S
  float* dl_s = lse_s + SK;        // delta[q]
  // this block's share of the QKV bias gradient (cs_out != null): cs_thirds x H column sums of dq (| dk | dv) over the rows of
  // its items, written to row blockIdx.x of cs_out at exit (a small kernel adds the blocks' rows: vault_colsum_partials)
  float* cs_lds = dl_s + SK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int tile0 = wave * KT;     // first 16-row tile of this wave
  const QkvLayout lay(H, heads, hm_rows);
  const int ld = lay.ld;
  const float sl2 = scale * LOG2E;
  const uint32_t ks_lds = (uint32_t)(size_t)LDS_PTR(char, Ks);
  const int G = (int)gridDim.x;
  const int S_st = (ATTN_ABLATE == 5 ? -S : S);
  // store instructions this wave issues per output matrix (wave-uniform: the counted waits below rely on it)
#if ATTN_WL   // (per tile: instruction A if the tile has rows, instruction B if it has more than 8)
  int nst_ = 0;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) nst_ += ((tile0 + kt) * 16 < S_st ? 1 : 0) + ((tile0 + kt) * 16 + 8 < S_st ? 1 : 0);
#else         // (two half-line stores per tile with rows)
  int nst_ = 0;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) nst_ += ((tile0 + kt) * 16 < S_st ? 2 : 0);
#endif
  const int nst = __builtin_amdgcn_readfirstlane(nst_);

  u32x4 rq[NC], rd[NC], ro[NC];    // Q, dO, O chunks of the item after next (rows >= S: row S - 1 again, see below)
  u32x4 rkf[KT][2], rvf[KT][2];    // this wave's own K / V tiles of the next item as MFMA fragments
  float rl = 0.f, rm = 1.f;
  // Rows >= S are never zero-filled: they re-read row S - 1.  Finite stand-ins are enough - a query row >= S has -lse = -inf,
  // so its probabilities and dS are exactly 0; a key row >= S has the -inf mask as initial accumulator - and unconditional
  // loads keep the number of memory operations per wave fixed, which the counted waits below rely on.
  uint32_t off_q[NC], off_o[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = tid + i * NT, row = min(c >> 3, S - 1), pos = c & 7;
    off_q[i] = (uint32_t)row * (uint32_t)(ld * 2) + (uint32_t)pos * 16u;
    off_o[i] = (uint32_t)row * (uint32_t)(H * 2) + (uint32_t)pos * 16u;
  }
  uint32_t off_frag[KT];           // own key rows, d = 8 g (+ 32 s)
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
    off_frag[kt] = (uint32_t)min((tile0 + kt) * 16 + l15, S - 1) * (uint32_t)(ld * 2) + (uint32_t)g * 16u;
  const uint32_t tile_step = 16u * (uint32_t)(ld * 2);                                                           // bytes between the wave's tiles
#if !ATTN_WL
  const uint32_t off_out = (uint32_t)(tile0 * 16 + l15) * (uint32_t)(ld * 2) + (uint32_t)g * 16u;             // own row, d = 8 g (+ 32 hf)
#endif
#if ATTN_WL
  // Whole-line stores: a lane holds the 16-byte chunks g (hf = 0) and 4 + g (hf = 1) of its row's 128-byte head slice, so a
  // store of one hf writes 16 half lines.  Lanes l15 and l15 ^ 8 swap one chunk each (DPP row_ror:8): instruction A then
  // writes rows 0-7 of the tile as whole lines (lanes l15 < 8: chunk g, lanes l15 >= 8: chunk 4 + g of row l15 - 8),
  // instruction B rows 8-15.  (Measured beforehand with the addresses alone: 5-8 % of the kernel.)
  const int row_a = tile0 * 16 + (l15 & 7);                                                                   // B: + 8 ; tile kt: + 16 kt
  const uint32_t off_wl = (uint32_t)row_a * (uint32_t)(ld * 2) + (uint32_t)(l15 >> 3) * 64u + (uint32_t)g * 16u;
  const uint32_t off_wl_b = 8u * (uint32_t)(ld * 2);
  auto wl_pair = [&](const u32x4& x, const u32x4& y, u32x4& a, u32x4& b) {     // x: chunk g, y: chunk 4 + g of the own row
    const bool lo8 = l15 < 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t xr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x[k], 0x128, 0xF, 0xF, true);   // row_ror:8
      const uint32_t yr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y[k], 0x128, 0xF, 0xF, true);
      a[k] = lo8 ? x[k] : yr;
      b[k] = lo8 ? xr : y[k];
    }
  };
#endif
  const uint32_t off_stat = (uint32_t)min(tid, S - 1) * 4u;
  auto item_bh = [&](int item, int& b, int& h) {
#if ATTN_ABLATE == 2 || ATTN_ABLATE == 6
    item = item % heads;
#endif
    b = item / heads; h = item - b * heads;
  };
  auto fetch_q_do = [&](int item) {      // LQ = 3 NC + 2 loads per wave, unconditional
    int b, h;
    item_bh(item, b, h);
    const size_t row0 = (size_t)b * S;
    const char* qb = attn_uniform(qkv + row0 * ld + (size_t)h * lay.hs);
    const char* ob = attn_uniform(ctx + row0 * H + h * 64);
    const char* db = attn_uniform(dctx + row0 * H + h * 64);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      attn_gload16(rq[i], qb, off_q[i]);
      attn_gload16(rd[i], db, off_o[i]);
      attn_gload16(ro[i], ob, off_o[i]);
    }
    attn_gload4(rl, attn_uniform(lse + (size_t)item * S), off_stat);
    attn_gload4(rm, attn_uniform(keymask != nullptr ? keymask + (size_t)b * S : lse + (size_t)item * S), off_stat);
  };
  auto fetch_kv_frags = [&](int item) {  // 4 KT loads per wave, unconditional
    int b, h;
    item_bh(item, b, h);
    const char* kb = attn_uniform(qkv + (size_t)b * S * ld + (size_t)h * lay.hs + lay.pl);
    const char* vb = attn_uniform(qkv + (size_t)b * S * ld + (size_t)h * lay.hs + 2 * (size_t)lay.pl);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        attn_gload16(rkf[kt][s_], kb, off_frag[kt] + 64u * s_);
        attn_gload16(rvf[kt][s_], vb, off_frag[kt] + 64u * s_);
      }
  };
  // the K image of an item by LDS-DMA: NC 1 KiB pieces (8 rows each) per wave, XOR swizzle on the source chunk; rows >= S are
  // never written (they stay zero from the start of the kernel)
  auto dma_k = [&](int item) {
    int b, h;
    item_bh(item, b, h);
    const char* kb = attn_uniform(qkv + (size_t)b * S * ld + (size_t)h * lay.hs + lay.pl);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int pc = wave * NC + i, row = pc * 8 + (lane >> 3), slot = lane & 7;
      const uint32_t voff = (uint32_t)row * (uint32_t)(ld * 2) + (uint32_t)((slot ^ (swz_row(row) << 1)) << 4);
      if (row < S) attn_glds16(kb, voff, (uint32_t)__builtin_amdgcn_readfirstlane((int)(ks_lds + pc * 1024)));
    }
  };
  // registers -> the images phase 2 does not read: Q, dO, delta, key bias, -lse (call only behind a wait for the loads)
  auto write_q_do = [&]() {
#pragma unroll
    for (int i = 0; i < NC; ++i) asm volatile("" : "+v"(rq[i]), "+v"(rd[i]), "+v"(ro[i]));
    asm volatile("" : "+v"(rl), "+v"(rm));
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = tid + i * NT, row = c >> 3, pos = c & 7;
      const int off = row * 128 + ((pos ^ (swz_row(row) << 1)) << 4);
      *reinterpret_cast<u32x4*>(Qs + off) = rq[i];
      *reinterpret_cast<u32x4*>(Ds + off) = rd[i];
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float2 a = unpack_h16x2(ro[i][w]), d = unpack_h16x2(rd[i][w]);
        s += a.x * d.x + a.y * d.y;
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (pos == 0) dl_s[row] = s;
    }
    if (tid < SK) {
      mb[tid] = (tid < S && (keymask == nullptr || rm != 0.f)) ? 0.f : -INFINITY;
      lse_s[tid] = (tid < S) ? -rl * LOG2E : -INFINITY;
    }
  };

  int item = blockIdx.x;
  if (item >= items) return;
  if (cs_out != nullptr)
    for (int i = tid; i < cs_thirds * H; i += NT) cs_lds[i] = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) *reinterpret_cast<u32x4*>(Ks + (tid + i * NT) * 16) = u32x4{0u, 0u, 0u, 0u};
  fetch_q_do(item);
  fetch_kv_frags(item);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // loads landed; the zeros are in place before the first DMA
  dma_k(item);
  write_q_do();
  if (item + G < items) fetch_q_do(item + G);
  for (; item < items; item += G) {
    h16x8 kf0[KT], kf1[KT], vf0[KT], vf1[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("" : "+v"(rkf[kt][0]), "+v"(rkf[kt][1]), "+v"(rvf[kt][0]), "+v"(rvf[kt][1]));   // (landed: waited for in front of the DMA / above)
      kf0[kt] = __builtin_bit_cast(h16x8, rkf[kt][0]); kf1[kt] = __builtin_bit_cast(h16x8, rkf[kt][1]);
      vf0[kt] = __builtin_bit_cast(h16x8, rvf[kt][0]); vf1[kt] = __builtin_bit_cast(h16x8, rvf[kt][1]);
    }
    int b, h;
    item_bh(item, b, h);
#if ATTN_ABLATE == 2
    b = 0;
#elif ATTN_ABLATE == 6
    b = item / heads; h = item - b * heads;
#endif
    const uint32_t bh = (uint32_t)item;
    char* dqbase = reinterpret_cast<char*>(dqkv + (size_t)b * S * ld + (size_t)h * lay.hs);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // Q / dO images + statistics of this item complete
    const bool more = item + G < items, more2 = item + 2 * G < items;

    // ---------------- phase 1: wave = key tiles tile0 ..; dK, dV in registers, dS^T -> LDS ----------------
    float mk[KT];
    f32x4 dk[KT][4], dv[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      mk[kt] = mb[(tile0 + kt) * 16 + l15];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dk[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    {
      char* dsrow = dSs + (tile0 * 16 + l15) * DS_LD + 8 * g;      // tile kt: + 16 kt DS_LD
_Pragma(ATTN_STR(unroll ATTN_P1_UNROLL))
      for (int T = 0; T < ((ATTN_ABLATE == 1 || ATTN_ABLATE == 4) ? 0 : NKT); ++T) {
        f32x4 p2[KT][2], ds2[KT][2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int qt = 2 * T + hh;
          const h16x8 q0 = frag_rows(Qs, qt, 0, g, l15), q1 = frag_rows(Qs, qt, 1, g, l15);
          const h16x8 d0 = frag_rows(Ds, qt, 0, g, l15), d1 = frag_rows(Ds, qt, 1, g, l15);
          const f32x4 nl4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * g);
          const f32x4 dl4 = *reinterpret_cast<const f32x4*>(dl_s + qt * 16 + 4 * g);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            f32x4 a = {mk[kt], mk[kt], mk[kt], mk[kt]}, dp = {0.f, 0.f, 0.f, 0.f};     // (the key mask is the initial accumulator)
            a = mfma16(q0, kf0[kt], a);
            a = mfma16(q1, kf1[kt], a);
            dp = mfma16(d0, vf0[kt], dp);
            dp = mfma16(d1, vf1[kt], dp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], sl2, nl4[r]));
              float dpv = dp[r];
              if (DROP && dr.thresh != 0u) {
                const uint32_t k_l = (uint32_t)((tile0 + kt) * 16 + l15);
                const uint32_t idx = (bh * (uint32_t)S + (uint32_t)(qt * 16 + 4 * g + r)) * (uint32_t)S + k_l;
                const bool keep = dropout_keep(dr.seed, dr.stream, idx, dr.thresh);
                dpv = keep ? dpv * dr.scale : 0.f;
                ds2[kt][hh][r] = pv * (dpv - dl4[r]);
                pv = keep ? pv * dr.scale : 0.f;
              } else {
                ds2[kt][hh][r] = pv * (dpv - dl4[r]);
              }
              p2[kt][hh][r] = pv;
            }
          }
        }
        h16x8 pf[KT], dsf[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          pf[kt] = pack_frag(p2[kt][0], p2[kt][1]);
          dsf[kt] = pack_frag(ds2[kt][0], ds2[kt][1]);
          // dS^T[key = this lane's][queries 32 T + 16 hh + 4 g + 0..3]: two 8-byte stores
          const u32x4 dsw = __builtin_bit_cast(u32x4, dsf[kt]);
          *reinterpret_cast<uint2*>(dsrow + kt * 16 * DS_LD + T * 64) = uint2{dsw[0], dsw[1]};
          *reinterpret_cast<uint2*>(dsrow + kt * 16 * DS_LD + T * 64 + 32) = uint2{dsw[2], dsw[3]};
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const h16x8 trd = frag_tr8(Ds, T, dt >> 1, dt & 1, g, l15), trq = frag_tr8(Qs, T, dt >> 1, dt & 1, g, l15);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            dv[kt][dt] = mfma16(trd, pf[kt], dv[kt][dt]);
            dk[kt][dt] = mfma16(trq, dsf[kt], dk[kt][dt]);
          }
        }
      }
    }
    // dS^T complete, every wave done with Q, dO and the statistics.  vmcnt(0): what is outstanding here - the K image of
    // this item (DMA issued an item ago), the Q / dO / O registers of the next item (requested an item ago) and old stores -
    // is needed right behind the barrier; phase 1 issued no memory operation
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (cs_out != nullptr && cs_thirds == 3) {      // key / value bias gradient: column sums of this wave's dk (scaled) / dv tiles
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        attn_tile_colsum(dk[kt], scale, cs_lds + H + h * 64, g, l15);
        attn_tile_colsum(dv[kt], 1.0f, cs_lds + 2 * H + h * 64, g, l15);
      }
    }
    if (more) write_q_do();                                           // images of the next item: phase 2 reads only K and dS^T
#if ATTN_LOADS_FIRST
    if (more) fetch_kv_frags(item + G);
    if (more2) fetch_q_do(item + 2 * G);
#endif
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#if ATTN_WL
      char* dstk = dqbase + 2 * (size_t)lay.pl;        // (byte offsets: K part at + pl elements, V part at + 2 pl elements)
      char* dstv = dqbase + 4 * (size_t)lay.pl;
      u32x4 wk[2], wv[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {    // tiles (2 hf, 2 hf + 1): d = 32 hf + 8 g + 0..7 of this lane's row
        const f32x4 ka = dk[kt][2 * hf], kb2 = dk[kt][2 * hf + 1], va = dv[kt][2 * hf], vb2 = dv[kt][2 * hf + 1];
        wk[hf] = u32x4{pack_h16x2(ka[0] * scale, ka[1] * scale), pack_h16x2(ka[2] * scale, ka[3] * scale),
                       pack_h16x2(kb2[0] * scale, kb2[1] * scale), pack_h16x2(kb2[2] * scale, kb2[3] * scale)};
        wv[hf] = u32x4{pack_h16x2(va[0], va[1]), pack_h16x2(va[2], va[3]), pack_h16x2(vb2[0], vb2[1]), pack_h16x2(vb2[2], vb2[3])};
      }
      u32x4 ka_, kb_, va_, vb_;
      wl_pair(wk[0], wk[1], ka_, kb_);
      wl_pair(wv[0], wv[1], va_, vb_);
      const uint32_t ow = off_wl + (uint32_t)kt * tile_step;
      if (row_a + 16 * kt < S_st) {
        ATTN_STORE16(dstk + ow, ka_);
        ATTN_STORE16(dstv + ow, va_);
      }
      if (row_a + 16 * kt + 8 < S_st) {
        ATTN_STORE16(dstk + (ow + off_wl_b), kb_);
        ATTN_STORE16(dstv + (ow + off_wl_b), vb_);
      }
#else
      if ((tile0 + kt) * 16 + l15 < S_st) {
        char* dstk = dqbase + 2 * (size_t)lay.pl;        // (byte offsets: K part at + pl elements, V part at + 2 pl elements)
        char* dstv = dqbase + 4 * (size_t)lay.pl;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {    // tiles (2 hf, 2 hf + 1): d = 32 hf + 8 g + 0..7 of this lane's row
          const f32x4 ka = dk[kt][2 * hf], kb2 = dk[kt][2 * hf + 1], va = dv[kt][2 * hf], vb2 = dv[kt][2 * hf + 1];
          const u32x4 wk = {pack_h16x2(ka[0] * scale, ka[1] * scale), pack_h16x2(ka[2] * scale, ka[3] * scale),
                            pack_h16x2(kb2[0] * scale, kb2[1] * scale), pack_h16x2(kb2[2] * scale, kb2[3] * scale)};
          const u32x4 wv = {pack_h16x2(va[0], va[1]), pack_h16x2(va[2], va[3]), pack_h16x2(vb2[0], vb2[1]), pack_h16x2(vb2[2], vb2[3])};
          ATTN_STORE16(dstk + (off_out + (uint32_t)kt * tile_step + 64u * hf), wk);
          ATTN_STORE16(dstv + (off_out + (uint32_t)kt * tile_step + 64u * hf), wv);
        }
      }
#endif
    }
    // requests of the following items, issued here so that they are in flight for a whole item: the K / V fragments of the
    // next item first (4 KT loads), then the Q / dO / O chunks + statistics of the item after next (LQ loads)
#if !ATTN_LOADS_FIRST
    if (more) fetch_kv_frags(item + G);
    if (more2) fetch_q_do(item + 2 * G);
#endif
    // ---------------- phase 2: wave = query tiles tile0 ..; dQ = dS K ----------------
    {
      f32x4 o[KT][4];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* dsb = dSs + (4 * g + (l15 >> 2)) * DS_LD + tile0 * 32 + (l15 & 3) * 8;     // query tile kt: + 32 kt
#pragma unroll
      for (int T = 0; T < ((ATTN_ABLATE == 1 || ATTN_ABLATE == 3) ? 0 : NKT); ++T) {
        const char* a = dsb + T * 32 * DS_LD;
        h16x8 dsB[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) dsB[kt] = cat_tr(lds_read_tr16(a + kt * 32), lds_read_tr16(a + kt * 32 + 16 * DS_LD));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const h16x8 trk = frag_tr8(Ks, T, dt >> 1, dt & 1, g, l15);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) o[kt][dt] = mfma16(trk, dsB[kt], o[kt][dt]);
        }
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        // query bias gradient: column sums of this wave's dq tile (rows q >= S are zero: their probabilities are)
        if (cs_out != nullptr) attn_tile_colsum(o[kt], scale, cs_lds + h * 64, g, l15);
#if ATTN_WL
        u32x4 wq[2], qa_, qb_;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const f32x4 qa = o[kt][2 * hf], qb2 = o[kt][2 * hf + 1];
          wq[hf] = u32x4{pack_h16x2(qa[0] * scale, qa[1] * scale), pack_h16x2(qa[2] * scale, qa[3] * scale),
                         pack_h16x2(qb2[0] * scale, qb2[1] * scale), pack_h16x2(qb2[2] * scale, qb2[3] * scale)};
        }
        wl_pair(wq[0], wq[1], qa_, qb_);
        const uint32_t ow = off_wl + (uint32_t)kt * tile_step;
        if (row_a + 16 * kt < S_st) ATTN_STORE16(dqbase + ow, qa_);
        if (row_a + 16 * kt + 8 < S_st) ATTN_STORE16(dqbase + (ow + off_wl_b), qb_);
#else
        if ((tile0 + kt) * 16 + l15 < S_st) {
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const f32x4 qa = o[kt][2 * hf], qb2 = o[kt][2 * hf + 1];
            const u32x4 w = {pack_h16x2(qa[0] * scale, qa[1] * scale), pack_h16x2(qa[2] * scale, qa[3] * scale),
                             pack_h16x2(qb2[0] * scale, qb2[1] * scale), pack_h16x2(qb2[2] * scale, qb2[3] * scale)};
            ATTN_STORE16(dqbase + (off_out + (uint32_t)kt * tile_step + 64u * hf), w);
          }
        }
#endif
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with K and dS^T
    if (more) {
      // the next item's K / V fragments must have landed; behind them in the (in-order) counter: LQ loads of the item after
      // next if requested, and this wave's dQ stores (nst of them: wave-uniform) - so the count is exact
#if ATTN_LOADS_FIRST   // (+ the dK / dV stores: twice as many again)
      const int behind = 3 * nst;
#else
      const int behind = nst;
#endif
      static_assert(2 * KT * (ATTN_LOADS_FIRST ? 3 : 1) <= 18, "cases below");
#define ATTN_WAIT_CASE(N) case N: if (more2) attn_wait_vm<LQ + N>(); else attn_wait_vm<N>(); break;
      switch (behind) {
        ATTN_WAIT_CASE(0) ATTN_WAIT_CASE(1) ATTN_WAIT_CASE(2) ATTN_WAIT_CASE(3) ATTN_WAIT_CASE(4) ATTN_WAIT_CASE(5) ATTN_WAIT_CASE(6)
        ATTN_WAIT_CASE(7) ATTN_WAIT_CASE(8) ATTN_WAIT_CASE(9) ATTN_WAIT_CASE(10) ATTN_WAIT_CASE(11) ATTN_WAIT_CASE(12)
        ATTN_WAIT_CASE(13) ATTN_WAIT_CASE(14) ATTN_WAIT_CASE(15) ATTN_WAIT_CASE(16) ATTN_WAIT_CASE(17) ATTN_WAIT_CASE(18)
        default: attn_wait_vm<0>(); break;
      }
#undef ATTN_WAIT_CASE
      dma_k(item + G);
    }
  }
  if (cs_out != nullptr) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float* dst = cs_out + (size_t)blockIdx.x * (size_t)(cs_thirds * H);
    for (int i = tid; i < cs_thirds * H; i += NT) dst[i] = cs_lds[i];
  }
}

template <int NKT>
constexpr int attn_one_lds_bytes() { return NKT * 32 * 128 * 3 + NKT * 32 * ds_ld<NKT * 32>() + NKT * 32 * 4 * 3; }

template <int NKT>
constexpr int attn_lds_bytes() { return NKT * 32 * 128 * 2 + NKT * 32 * 4 * 3; }

}  // namespace

extern "C" int vault_attention_fwd(const vault_attn_args* a, void* stream) {
  if (!a || !a->qkv || (!a->ctx && !a->ctx_split3) || !a->lse || a->S <= 0 || a->B <= 0 || a->H != a->heads * 64)
    return VAULT_EINVAL;
  if (a->S > 320) return VAULT_EINVAL;
  if (a->qkv_hm != 0 && (a->qkv_hm < a->B * a->S || a->ctx_split3 != nullptr)) return VAULT_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const AttnDrop dr{a->drop_thresh, a->drop_seed, a->drop_stream, a->drop_scale};
  dim3 grid(a->heads, a->B), block(256);
  const float scale = 0.125f;  // 1/sqrt(64)
#define FWD_V(NK, NW, WP, DR)                                                                                          \
    hipLaunchKernelGGL((attn_fwd_kernel<NK, NW, WP, DR>), grid, dim3(NW * 64), attn_lds_bytes<NK>(), st,                \
                       reinterpret_cast<const h16*>(a->qkv), a->keymask, reinterpret_cast<h16*>(a->ctx), a->lse, a->S, \
                       a->H, a->heads, scale, dr, reinterpret_cast<h16*>(a->ctx_split3), a->qkv_hm)
  // (dropout is a template switch: a per-element run-time test splits the loop body into basic blocks that the
  //  instruction scheduler cannot move MFMAs and LDS reads across)
  const bool drop = a->drop_thresh != 0u;
  if (a->S <= 64) {
    if (drop) FWD_V(2, 4, 3, true); else FWD_V(2, 4, 3, false);
  } else if (a->S <= 192) {
    if (drop) FWD_V(6, 12, 6, true); else FWD_V(6, 12, 6, false);
  } else if (a->S <= 288) {   // padded batches up to the HF processor's 384 x 640 canvas (281 tokens): 72 KiB K / V image, two blocks per CU
    auto kern = attn_fwd_kernel<9, ATTN_L9_FWD_W, ATTN_L9_FWD_WPE>;
    static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         attn_lds_bytes<9>());
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(ATTN_L9_FWD_W * 64), attn_lds_bytes<9>(), st, reinterpret_cast<const h16*>(a->qkv),
                       a->keymask, reinterpret_cast<h16*>(a->ctx), a->lse, a->S, a->H, a->heads, scale, dr,
                       reinterpret_cast<h16*>(a->ctx_split3), a->qkv_hm);
  } else {   // long sequences of padded, larger images: K/V image 80 KiB -> one block per CU
    auto kern = attn_fwd_kernel<10, ATTN_LONG_WAVES_FWD, 1>;
    static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         attn_lds_bytes<10>());
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(ATTN_LONG_WAVES_FWD * 64), attn_lds_bytes<10>(), st, reinterpret_cast<const h16*>(a->qkv),
                       a->keymask, reinterpret_cast<h16*>(a->ctx), a->lse, a->S, a->H, a->heads, scale, dr,
                       reinterpret_cast<h16*>(a->ctx_split3), a->qkv_hm);
  }
#undef FWD_V
  return (int)hipGetLastError();
}

// rows of `bias_partials` a backward launch with these arguments writes (= its grid), 0 when the shape has no such form
extern "C" int vault_attention_bwd_partials(const vault_attn_args* a) {
  if (!a || a->S <= 0 || a->B <= 0 || a->heads <= 0 || a->H != a->heads * 64 || a->S > 192) return 0;
  if (a->bias_thirds != 1 && a->bias_thirds != 3) return 0;
  const int items = a->B * a->heads;
  if (a->S <= 64) return a->bias_thirds * a->H * 4 <= 16384 ? (items < 768 ? items : 768) : 0;
  return (a->bias_thirds == 1 && a->H <= 1024) ? (items < 256 ? items : 256) : 0;
}

extern "C" int vault_attention_bwd(const vault_attn_args* a, void* stream) {
  if (!a || !a->qkv || !a->ctx || !a->lse || !a->dctx || !a->dqkv || a->S <= 0 || a->B <= 0 ||
      a->H != a->heads * 64)
    return VAULT_EINVAL;
  if (a->S > 320) return VAULT_EINVAL;
  // head-major qkv / dqkv: the single-pass kernels (S <= 192), not the split3 (precise) output
  if (a->qkv_hm != 0 && (a->qkv_hm < a->B * a->S || a->S > 192)) return VAULT_EINVAL;
  // bias partials: the single-pass kernels; the query third alone, or (S <= 64: LDS) all three
  if (a->bias_partials != nullptr && vault_attention_bwd_partials(a) == 0) return VAULT_EINVAL;
  const int cs_bytes = a->bias_partials ? a->bias_thirds * a->H * 4 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const AttnDrop dr{a->drop_thresh, a->drop_seed, a->drop_stream, a->drop_scale};
  dim3 grid(a->heads, a->B), block(256);
  const float scale = 0.125f;
  const bool drop = a->drop_thresh != 0u;
#define ONE_V(DR)                                                                                                             \
    {                                                                                                                         \
      /* 152 KiB of LDS (+ 4 KiB of bias partials): one workgroup per CU, persistent; ATTN_BWD_KT key tiles per wave */ \
      auto kern = attn_bwd_one_kernel<6, 12 / ATTN_BWD_KT, DR, (ATTN_BWD_KT == 1 ? 1 : (ATTN_BWD_KT == 2 ? 2 : 1)), ATTN_BWD_KT>;   \
      static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];                                                                                          \
      if (!attr_done) {                                                                                                       \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                           attn_one_lds_bytes<6>() + 4096);                                                   \
        if (e != hipSuccess) return (int)e;                                                                                   \
        attr_done = true;                                                                                                     \
      }                                                                                                                       \
      hipLaunchKernelGGL(kern, dim3(items < 256 ? items : 256), dim3(768 / ATTN_BWD_KT), attn_one_lds_bytes<6>() + cs_bytes, st, \
                         reinterpret_cast<const h16*>(a->qkv), a->keymask, reinterpret_cast<const h16*>(a->ctx),            \
                         reinterpret_cast<const h16*>(a->dctx), a->lse, reinterpret_cast<h16*>(a->dqkv), a->S, a->H,        \
                         a->heads, items, scale, dr, a->qkv_hm, a->bias_partials, a->bias_thirds);                            \
    }
  if (a->S <= 64) {
    // text-only sequences (the LM stack): the single-pass kernel with four waves per workgroup (one 16-key tile each) and
    // three workgroups per CU (35 KiB of LDS, <= 168 registers), persistent over the (batch, head) items
    const int items = a->B * a->heads;
    const int grid = items < 768 ? items : 768;
    if (drop) hipLaunchKernelGGL((attn_bwd_one_kernel<2, 4, true, 3>), dim3(grid), dim3(256), attn_one_lds_bytes<2>() + cs_bytes, st,
                                 reinterpret_cast<const h16*>(a->qkv), a->keymask, reinterpret_cast<const h16*>(a->ctx),
                                 reinterpret_cast<const h16*>(a->dctx), a->lse, reinterpret_cast<h16*>(a->dqkv), a->S, a->H,
                                 a->heads, items, scale, dr, a->qkv_hm, a->bias_partials, a->bias_thirds);
    else hipLaunchKernelGGL((attn_bwd_one_kernel<2, 4, false, 3>), dim3(grid), dim3(256), attn_one_lds_bytes<2>() + cs_bytes, st,
                            reinterpret_cast<const h16*>(a->qkv), a->keymask, reinterpret_cast<const h16*>(a->ctx),
                            reinterpret_cast<const h16*>(a->dctx), a->lse, reinterpret_cast<h16*>(a->dqkv), a->S, a->H,
                            a->heads, items, scale, dr, a->qkv_hm, a->bias_partials, a->bias_thirds);
  } else if (a->S <= 192) {
    const int items = a->B * a->heads;
    if (drop) ONE_V(true) else ONE_V(false)
  } else if (a->S <= 288) {
    auto kern = attn_bwd_kernel<9, ATTN_L9_BWD_W, ATTN_L9_BWD_WPE>;
    static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         attn_lds_bytes<9>());
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(ATTN_L9_BWD_W * 64), attn_lds_bytes<9>(), st, reinterpret_cast<const h16*>(a->qkv),
                       a->keymask, reinterpret_cast<const h16*>(a->ctx), reinterpret_cast<const h16*>(a->dctx), a->lse,
                       reinterpret_cast<h16*>(a->dqkv), a->S, a->H, a->heads, scale, dr);
  } else {
    auto kern = attn_bwd_kernel<10, ATTN_LONG_WAVES_BWD, 1>;
    static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         attn_lds_bytes<10>());
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(ATTN_LONG_WAVES_BWD * 64), attn_lds_bytes<10>(), st, reinterpret_cast<const h16*>(a->qkv),
                       a->keymask, reinterpret_cast<const h16*>(a->ctx), reinterpret_cast<const h16*>(a->dctx), a->lse,
                       reinterpret_cast<h16*>(a->dqkv), a->S, a->H, a->heads, scale, dr);
  }
#undef ONE_V
  return (int)hipGetLastError();
}
