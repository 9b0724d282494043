// bf16 MFMA GEMM for gfx950 with fused epilogues: the work-horse of the BERT->ViLT hot path.
//
//   C[M,N] (+)= A . B        fp32 accumulate, v_mfma_f32_16x16x32_bf16
//
// Operand storage modes (so forward, dgrad and wgrad all run without transposed copies):
//   A_MODE 0: A stored [M][K] (K contiguous)   -> fragments by ds_read_b128
//   A_MODE 1: A stored [K][M] (M contiguous)   -> fragments by ds_read_b64_tr_b16
//   B_MODE 0: B stored [N][K] (K contiguous; an HF Linear weight [out,in])
//   B_MODE 1: B stored [K][N] (N contiguous)
//     forward  Y = X W^T        : A0 (X [M,K])      B0 (W [N,K])
//     dgrad    dX = dY W        : A0 (dY [M,Nout])  B1 (W [Nout,Kin] = [K][N])
//     wgrad    dW = dY^T X      : A1 (dY [Mtok,Nout] = [K][M])  B1 (X [Mtok,Kin] = [K][N])
//
// Structure: BMxBNx64 tiles, global->LDS by 16-byte global_load_lds (double buffered, one
// barrier per K tile), XOR-swizzled LDS images (swizzle applied on the per-lane SOURCE address
// and on the read address; the LDS destination of a global_load_lds is lane-linear),
// XCD-aware block->tile mapping, LDS-staged epilogue so that every global store/load of the
// epilogue is a full 128-byte row segment.
//
// Replaces (reference, via HuggingFace/ATen): every nn.Linear of
// HF:models/vilt/modeling_vilt.py:303-414 and HF:models/roberta/modeling_roberta.py:222-398, the
// Conv2d patch projection (modeling_vilt.py:290-300) and their autograd backward.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "gemm.h"
#include "gemm_epi.h"

#ifndef GEMM_NS4_MIN_K
#define GEMM_NS4_MIN_K 768    // shortest contraction that takes the four-stage 128 x 128 form (rounds 3-5: 1536; round 6: K = 768 at <= 256
                              // tiles - the LM stack's N = 3072 / 2304 Linears at B <= 32 - 16.1 -> 15.5, 13.2 -> 12.9, 12.6 -> 12.2 us, +0.2-0.3 % of the
                              // B = 32 step: profiles/r06_dev_four_stage_k768.txt)
#endif

namespace {

constexpr int BK = 64;

template <int EPI> struct EpiTraits;

// 16-byte LDS-DMA through inline asm for the deep-pipeline form (NS > 2): hipcc keeps no record of it, so it does not put
// `s_waitcnt vmcnt(0)` in front of the barrier / the fragment reads (which would drain every stage in flight); the loop
// waits with counted `vmcnt` instead.  gptr: per-lane global address, lds: uniform LDS byte address (lane l lands at + 16 l).
__device__ __forceinline__ void gemm_glds16_asm(const void* gptr, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gptr), "s"(lds) : "memory", "m0");
}

// NS = 2: double buffered, two blocks of 128 x 128 per CU.  NS = 4 (round 3): four stages, three K tiles in flight, one block
// per CU - for the launches whose K loop runs (nearly) alone on its CU: a block of the double-buffered form waits a full
// L2 / MALL round trip per K tile (1.3 us per 64-deep step at M = 2560, K = 3072: 10 % of the matrix rate of its tile),
// three tiles in flight divide that by three.
template <int BM, int BN, int WM, int WN, int A_MODE, int B_MODE, int EPI, int NS = 2>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmParams p_in) {
  H16_SATURATE();
  GemmParams p = p_in;
  int split_z = blockIdx.z;
  if constexpr (EPI == EPI_F32_ATOMIC) {
    if (p.batch > 1) {   // batched weight gradients: z = problem * splits + split
      const int bz = blockIdx.z / p.splits;
      split_z = blockIdx.z - bz * p.splits;
      p.A += (size_t)bz * p.batch_a;
      p.B += (size_t)bz * p.batch_b;
      p.out = reinterpret_cast<float*>(p.out) + (size_t)bz * p.batch_o;
    }
  }
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NIA = A_BYTES / 1024 / NW;  // global_load_lds wave-instructions per wave per tile
  constexpr int NIB = B_BYTES / 1024 / NW;
  static_assert(NIA >= 1 && NIB >= 1, "tile too small for the wave count");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WN) * (BM / WM);
  const int wn0 = (wave % WN) * (BN / WN);

  // ---- block -> tile (XCD-aware: blocks with equal id%8 share an L2; give each XCD a
  //      contiguous run of tiles so the A row panel and the whole of B stay L2-resident)
  //      Persistent form: the grid holds at most `resident` blocks; each walks the raster with stride
  //      gridDim.x, so the epilogue's stores of one tile drain under the next tile's main loop.
  const int total_tiles = (p.M / BM) * (p.N / BN);
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
  int tile_m, tile_n;
  gemm_tile_of_block(total_tiles, tile, p.M / BM, p.N / BN, p.gn, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- split-K range
  const int nk_total = p.K / BK;
  const int per = (nk_total + p.splits - 1) / p.splits;
  const int kt0 = split_z * per;
  const int kt1 = min(nk_total, kt0 + per);
  const int nk = kt1 - kt0;

  // ---- per-lane staging sources
  const h16* a_src[NIA];
  const h16* b_src[NIB];
  if constexpr (A_MODE == 0) {
    const int r8 = lane >> 3, pos = lane & 7;
    const int c = pos ^ (((r8 >> 1) & 3) << 1);
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int row = 8 * (wave * NIA + i) + r8;
      a_src[i] = p.A + (size_t)(m0 + row) * p.lda + (size_t)kt0 * BK + c * 8;
    }
  } else {
    constexpr int CPR = BM / 8;          // 16-byte chunks per k-row
    constexpr int RPI = 64 / CPR;        // k-rows per wave-instruction
    const int kin = lane / CPR, pos16 = lane % CPR;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int krow = (wave * NIA + i) * RPI + kin;
      const int h = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int col = (((pos16 >> 1) ^ h) << 4) + ((pos16 & 1) << 3);
      a_src[i] = p.A + (size_t)(kt0 * BK + krow) * p.lda + m0 + col;
    }
  }
  if constexpr (B_MODE == 0) {
    const int r8 = lane >> 3, pos = lane & 7;
    const int c = pos ^ (((r8 >> 1) & 3) << 1);
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int row = 8 * (wave * NIB + i) + r8;
      b_src[i] = p.B + (size_t)(n0 + row) * p.ldb + (size_t)kt0 * BK + c * 8;
    }
  } else {
    constexpr int CPR = BN / 8;
    constexpr int RPI = 64 / CPR;
    const int kin = lane / CPR, pos16 = lane % CPR;
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int krow = (wave * NIB + i) * RPI + kin;
      const int h = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int col = (((pos16 >> 1) ^ h) << 4) + ((pos16 & 1) << 3);
      b_src[i] = p.B + (size_t)(kt0 * BK + krow) * p.ldb + n0 + col;
    }
  }
  const size_t a_step = (A_MODE == 0) ? (size_t)BK : (size_t)BK * p.lda;
  const size_t b_step = (B_MODE == 0) ? (size_t)BK : (size_t)BK * p.ldb;

  const uint32_t lds0 = (uint32_t)(size_t)LDS_PTR(char, smem);
  auto stage = [&](int buf, int t) {
    if constexpr (NS == 2) {
      char* sa = smem + buf * STAGE + wave * NIA * 1024;
      char* sb = smem + buf * STAGE + A_BYTES + wave * NIB * 1024;
#pragma unroll
      for (int i = 0; i < NIA; ++i) glds16(a_src[i] + (size_t)t * a_step, sa + i * 1024);
#pragma unroll
      for (int i = 0; i < NIB; ++i) glds16(b_src[i] + (size_t)t * b_step, sb + i * 1024);
    } else {
      const uint32_t sa = lds0 + (uint32_t)(buf * STAGE + wave * NIA * 1024);
      const uint32_t sb = lds0 + (uint32_t)(buf * STAGE + A_BYTES + wave * NIB * 1024);
#pragma unroll
      for (int i = 0; i < NIA; ++i) gemm_glds16_asm(a_src[i] + (size_t)t * a_step, (uint32_t)__builtin_amdgcn_readfirstlane((int)(sa + i * 1024)));
#pragma unroll
      for (int i = 0; i < NIB; ++i) gemm_glds16_asm(b_src[i] + (size_t)t * b_step, (uint32_t)__builtin_amdgcn_readfirstlane((int)(sb + i * 1024)));
    }
  };

  // ---- per-lane fragment read offsets (bytes inside the A / B image of a stage)
  const int g = lane >> 4, l15 = lane & 15;
  int a_off[2], b_off[2];  // [k-step] for mode 0 ; [half] for mode 1
  if constexpr (A_MODE == 0) {
    const int fx = ((l15 >> 1) & 3) << 1;
    a_off[0] = (wm0 + l15) * 128 + ((g ^ fx) << 4);
    a_off[1] = (wm0 + l15) * 128 + (((4 + g) ^ fx) << 4);
  } else {
    const int q = l15 >> 2, pp = l15 & 3, h = q | ((g & 1) << 2);
    // address of (krow = 8g + q [+4], col block cb): krow*BM*2 + ((cb ^ h) * 32) + pp*8 ; cb added later
    a_off[0] = (8 * g + q) * (BM * 2) + pp * 8;
    a_off[1] = h;  // swizzle key
  }
  if constexpr (B_MODE == 0) {
    const int fx = ((l15 >> 1) & 3) << 1;
    b_off[0] = (wn0 + l15) * 128 + ((g ^ fx) << 4);
    b_off[1] = (wn0 + l15) * 128 + (((4 + g) ^ fx) << 4);
  } else {
    const int q = l15 >> 2, pp = l15 & 3, h = q | ((g & 1) << 2);
    b_off[0] = (8 * g + q) * (BN * 2) + pp * 8;
    b_off[1] = h;
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (NS == 2) {
    if (nk > 0) stage(0, 0);
  } else {
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
      if (s_ < nk) stage(s_, s_);
  }
  for (int t = 0; t < nk; ++t) {
    if constexpr (NS == 2) {
      __syncthreads();  // tile t has landed (vmcnt(0) precedes the barrier) and buffer (t+1)&1 is free
      if (t + 1 < nk) stage((t + 1) & 1, t + 1);
    } else {
      // tile t has landed when at most the DMA pieces of the younger tiles in flight (min(nk - 1 - t, NS - 2) tiles of
      // NIA + NIB pieces per wave) are outstanding; the barrier also frees buffer (t - 1) % NS for tile t + NS - 1
      static_assert(NS == 4 && NIA + NIB <= 16, "counted waits below are written for two tiles in flight");
      const int younger = min(nk - 1 - t, NS - 2);
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (NIA + NIB)) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIA + NIB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (t + NS - 1 < nk) stage((t + NS - 1) % NS, t + NS - 1);
    }
    const char* As = smem + (t % NS) * STAGE;
    const char* Bs = As + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if constexpr (A_MODE == 0) {
          af[i] = *LDS_PTR(const h16x8, As + a_off[s] + i * 16 * 128);
        } else {
          const int cb = (wm0 >> 4) + i;
          const char* base = As + a_off[0] + s * 32 * (BM * 2) + ((cb ^ a_off[1]) << 5);
          af[i] = cat_tr(lds_read_tr16(base), lds_read_tr16(base + 4 * (BM * 2)));
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (B_MODE == 0) {
          bfr[j] = *LDS_PTR(const h16x8, Bs + b_off[s] + j * 16 * 128);
        } else {
          const int cb = (wn0 >> 4) + j;
          const char* base = Bs + b_off[0] + s * 32 * (BN * 2) + ((cb ^ b_off[1]) << 5);
          bfr[j] = cat_tr(lds_read_tr16(base), lds_read_tr16(base + 4 * (BN * 2)));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = mfma16(af[i], bfr[j], acc[i][j]);
    }
  }

  gemm_epilogue<TM, TN, EPI, WM, WN>(acc, p, smem, m0, n0, wm0, wn0, wave, lane);
  __syncthreads();   // epilogue scratch (LDS) is free again before the next tile stages into it
  }
}

template <int BM, int BN, int WM, int WN, int A_MODE, int B_MODE, int EPI, int NS = 2>
int launch_cfg(const GemmParams& p, hipStream_t st) {
  if (p.M % BM || p.N % BN || p.K % BK) return VAULT_EINVAL;
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int LDS = NS * STAGE;
  auto kern = gemm_kernel<BM, BN, WM, WN, A_MODE, B_MODE, EPI, NS>;
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VAULT_EINVAL;
  if (!attr_done[dev]) {   // (function attributes are per device)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_done[dev] = true;
  }
  GemmParams q = p;
  // raster groups of 4 n-tiles for short contractions (the group's B panel, 4 x BN x K bf16 <= 1.5 MiB, stays
  // in the XCD's L2): in-process A/B on M = 47360, K = 768: QKV 197 -> 183 us, FFN-in 252 -> 234 us, dgrad
  // FFN-out 290 -> 257 us (tools/gn_ab.py); no gain for K = 3072 or for the ring kernel -> plain raster there
  q.gn = (p.gn > 0) ? std::min(p.gn, p.N / BN) : (p.K <= 1024 ? std::min(p.N / BN, 4) : p.N / BN);
  // resident blocks: 256 CUs x (blocks that fit: LDS-limited, 160 KiB per CU)
  const int resident = 256 * std::max(1, (160 * 1024) / LDS);
  const int total = (p.M / BM) * (p.N / BN);
  const int nbatch = (EPI == EPI_F32_ATOMIC && p.batch > 1) ? p.batch : 1;
  dim3 grid((p.persist & 2) == 0 ? std::min(total, resident) : total, 1, p.splits * nbatch);   // persistent unless (persist & 2)
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), LDS, st, q);
  return (int)hipGetLastError();
}

template <int A_MODE, int B_MODE, int EPI>
int launch_modes(const GemmParams& p, int cfg, hipStream_t st) {
  switch (cfg) {
    case 0: {
      // four-stage form for forward / data-gradient launches where one block per CU covers the launch and the K loop is long: M = 2560 (tools/gemm_bench.py, same box) FFN-in
      // dgrad 50.0 -> 34.1 us, QKV dgrad 38.6 -> 26.3, FFN-out forward 36.6 -> 34.7; with more tiles than CUs two
      // double-buffered blocks per CU are faster (QKV forward 20.3 against 23.3 us), and a single block's K loop stays bound
      // by its CU's L2 -> LDS rate (32 KiB per 2.1 MFLOP step)
      if constexpr (A_MODE == 0 && EPI != EPI_F32_ATOMIC) {
        if (p.splits <= 1 && (long)(p.M / 128) * (p.N / 128) <= 256 && p.K >= GEMM_NS4_MIN_K)
          return launch_cfg<128, 128, 2, 2, A_MODE, B_MODE, EPI, 4>(p, st);
      }
      return launch_cfg<128, 128, 2, 2, A_MODE, B_MODE, EPI>(p, st);
    }
    case 1: return launch_cfg<256, 128, 4, 2, A_MODE, B_MODE, EPI>(p, st);
    case 2: return launch_cfg<256, 256, 2, 4, A_MODE, B_MODE, EPI>(p, st);
    case 7:   // 64 x 128 tiles, four stages: twice the blocks of cfg 0 for the launches that leave half of the CUs idle
      if constexpr (A_MODE == 0 && EPI != EPI_F32_ATOMIC) return launch_cfg<64, 128, 2, 2, A_MODE, B_MODE, EPI, 4>(p, st);
      return VAULT_EINVAL;
    default: return VAULT_EINVAL;
  }
}

}  // namespace

int vault_gemm256_launch(const GemmParams& p, int a_mode, int b_mode, int epi, int ntq, hipStream_t st);
bool vault_gemm8w_supports(const GemmParams& p, int a_mode, int b_mode, int epi, int ntw);
int vault_gemm8w_launch(const GemmParams& p, int epi, int ntw, hipStream_t st);

// Split count of the in-launch split-K reduction (gemm256.hip SK: 192-wide ring tiles, (0,1) EPI_BF16 / (0,0) EPI_F32_RES) for a
// launch whose tiles fill less than a round of the chip: 1 = do not split.  Measured (tools/splitk_bench.py,
// profiles/r06_dev_splitk_bench.txt; N = 768, K = 3072 / 2304, us): a K tile of a 256 x 192 item takes a CU ~1.0; the hand-off
// costs ~10 at two splits (every split stores a 192 KiB slab at the same moment: 37 MB at 96 tiles - the launch's own operand
// bytes once more - and the reducers read half of it back) and ~3.5 more per further slab the reducer reads.  So two splits pay
// where the un-split alternatives are one half-empty round - 88 / 96..128 tiles (M = 5,632..8,192: ViLT at B = 31..44): 46-51 -> 42-45
// (K = 3072), 40 -> 37 (K = 2304) - and nowhere else: below, the 64 x 128 / 128 x 128 kernels (more CUs staging at once) are at
// 17-32 against 31-38; above, 128-wide ring tiles fill the round (43-46 against 66-84).
static int gemm_sk_splits(const GemmParams& p, int a_mode, int b_mode, int epi) {
  if (p.sk_ws == nullptr || a_mode != 0 || p.batch > 1 || p.split3 || (p.M & 255) || p.N % 192 || (p.K & 63) || p.out_hm) return 1;
  if (!((b_mode == 1 && epi == EPI_BF16) || (b_mode == 0 && epi == EPI_F32_RES && p.res != nullptr))) return 1;
  const long tiles = (long)(p.M >> 8) * (p.N / 192);
  const int nk = p.K >> 6;
  // (the f32-residual forward gains 4 % at 96 tiles and loses against the 128 x 128 kernel at 80 - the LM stack at B = 128, measured
  //  in the step of BASELINE config 4: -5 % on its LM block; the data gradients gain 7-17 % at 96)
  if (tiles < (epi == EPI_F32_RES ? 96 : 88) || tiles > 128 || nk < 32) return 1;
  if (p.sk_bytes < GEMM_SK_COUNTER_BYTES + tiles * 2 * (256LL * 192 * 4)) return 1;
  return 2;
}

// argument checks + kernel choice: the resolved cfg (0..8), or -VAULT_EINVAL
int vault_gemm_resolve(GemmParams& p, int a_mode, int b_mode, int epi, int cfg) {
  if (p.splits < 1) p.splits = 1;
  if (p.A == nullptr || p.B == nullptr || p.out == nullptr) return -VAULT_EINVAL;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return -VAULT_EINVAL;
  if (p.out_q != nullptr || p.out_scale != nullptr) return -VAULT_EINVAL;   // (the MXFP8 output image: vault_gemm_mxfp8 only)
  if ((p.lda & 7) || (p.ldb & 7) || (p.ldo & 7)) return -VAULT_EINVAL;
  if (p.batch > 1) {   // batched weight gradients: double-buffered kernel, atomic epilogue only
    if (epi != EPI_F32_ATOMIC || cfg == 4 || (p.batch_a & 7) || (p.batch_b & 7) || (p.batch_o & 3) ||
        (long long)p.batch * p.splits > 65535)
      return -VAULT_EINVAL;
    if (cfg < 0) cfg = (p.M % 256 == 0 && p.N % 256 == 0) ? 3 : ((p.M % 128 == 0 && p.N % 128 == 0) ? 0 : -1);
    if (cfg < 0) return -VAULT_EINVAL;
  }
  const bool auto_cfg = cfg < 0;
  if (cfg < 0) {
    // default kernel/tile choice (measured on MI355X at M = 47360, tools/gemm_bench.py + tools/k_sweep.py):
    //   256x256 persistent ring kernel from K = 512 up and for all wgrads (A stored [K][M]): its epilogue
    //   overlaps the next tile, the double-buffered kernel only wins for very short contractions;
    //   the 256x192 form of the ring kernel when that fills the last round of CUs better
    //   (N = 768: 740 vs 555 tiles, N = 2304: 2220 vs 1665); 256x128 / 128x128 otherwise
    auto eff = [](long tiles) { return (double)tiles / (double)(((tiles + 255) / 256) * 256); };
    if (p.M % 256 == 0 && p.N % 256 == 0) {
      // (the ring kernel's (1,1) form exists for the weight-gradient epilogue only)
      cfg = ((p.K >= 512 && a_mode == 0) || (a_mode == 1 && epi == EPI_F32_ATOMIC)) ? 3 : 2;
      if (p.N % 192 == 0 && a_mode == 0 && epi != EPI_F32_ATOMIC && p.splits == 1 && p.K >= 512 &&
          eff((long)(p.M / 256) * (p.N / 192)) > 1.02 * eff((long)(p.M / 256) * (p.N / 256)))
        cfg = 4;
    } else {
      cfg = (p.M % 256 == 0 && p.N % 128 == 0 && epi != EPI_F32_ATOMIC) ? 1 : 0;
    }
  }
  // bf16-output Linears with short contractions (forward-form operands: QKV, FFN-in (GELU), and - on the transposed
  // weight shadow - the FFN-out (gelu' product) and attention-out data gradients): the 8-wave kernel whose epilogue
  // stores from registers (tools/gemm_bench.py at M = 47360, same box: QKV 168 against 193-198 us, FFN-in 299 against
  // 355-388, gelu'-product dgrad 282 against 294-317, N = K = 768 dgrad 55 against ~90).  Not the f32-residual forms
  // (bound by their HBM bytes: the ring kernel's 192-wide form is as fast) and not K = 3072 (ring: better main loop).
  // Few row tiles (LM stack at batch <= 128, both stacks at batch 8): the 256-row kernels leave most CUs idle and a
  // block's K loop runs alone on its CU - 128x128 tiles spread the same work over 4x the blocks (tools/gemm_bench.py,
  // M = 2560: attention-out 13.4 against 21.0 us, FFN-out 36.3 against 53.3, FFN-in 23.4 against 26.6, gelu'-product
  // dgrad 20.5 against 31; M = 5120: FFN-in 41.2 against 45.8, FFN-out 40.5 against 56.3; from M = 10240 the big tiles win)
  bool small = false;
  if (auto_cfg && p.batch <= 1 && epi != EPI_F32_ATOMIC && a_mode == 0 && p.M % 256 == 0 && p.N % 128 == 0) {
    const long c192 = (long)(p.M / 256) * ((p.N + 191) / 192), c256 = (long)(p.M / 256) * ((p.N + 255) / 256);
    small = c192 < 128 || (p.N >= 3072 && c256 <= 256);
    if (small) cfg = 0;
    // ... and 64 x 128 tiles (four stages) while those still fit one block per CU: a block's K loop is bound by its CU's
    // L2 -> LDS rate (~20 B/clk), i.e. by the bytes a K tile stages - 24 instead of 32 KiB - and twice the CUs work
    // (tools/small_tile_bench.py, M = 2560, N = 768: FFN-in dgrad 34.1 -> 23.9 us, QKV dgrad 27.5 -> 18.9, FFN-out forward
    // 34.1 -> 24.3, attention-out forward 12.8 -> 9.6, its dgrad 11.5 -> 8.4; the same ratios down to M = 512; from 257 blocks
    // on the second round costs more: M = 3072 32.5 -> 41.0)
    if (small && p.splits <= 1 && (long)(p.M / 64) * (p.N / 128) <= 256) cfg = 7;
  }
  // VAULT_GEMM8W=0 (read once per process) keeps the 8-wave kernel out of the automatic choice: the one kernel-choice switch
  // kept, because the engine's 8-bit gelu' plan must FOLLOW the choice (tests/test_gpu_train.py::
  // test_8bit_gelu_prime_follows_the_kernel_choice)
  static const bool use8w = [] { const char* e = getenv("VAULT_GEMM8W"); return !(e && e[0] == '0'); }();
  // (also in data-parallel steps - persist bit 0 -: its static tile walk under an RCCL kernel that holds 8-64 CUs costs what the
  //  ring kernel's dynamic scheduler costs there, 235-250 against 207-240 us for the QKV shape, and nothing when the GPU is
  //  not shared, 165 against 196-239 us: tools/contention_test.py)
  if (auto_cfg && !small && use8w && a_mode == 0 && b_mode == 0 && p.K <= 1024 && p.M >= 2048 &&
      (epi == EPI_BF16 || epi == EPI_BF16_GELU || epi == EPI_BF16_DGELU)) {
    auto eff8 = [](long tiles) { return (double)tiles / (double)(((tiles + 255) / 256) * 256); };
    const bool ok4 = vault_gemm8w_supports(p, a_mode, b_mode, epi, 4), ok3 = vault_gemm8w_supports(p, a_mode, b_mode, epi, 3);
    // (256-wide tiles unless the 192-wide ones fill the last round of CUs much better: N = 768; at N = 2304 - 6.5
    //  against 8.7 rounds - the wider tile's fewer staging bytes per FLOP still win: 168 against 188 us)
    if (ok4 && (!ok3 || eff8((long)(p.M / 256) * (p.N / 256)) >= 0.9 * eff8((long)(p.M / 256) * (p.N / 192)))) cfg = 5;
    else if (ok3) cfg = 6;
  }
  // split-K with the reduction inside the launch (cfg 4 + splits, round 6) where the caller gave a workspace and the model above
  // says the launch fills the chip better that way: the N = 768 Linears with K = 3072 / 2304 at small batches
  if (auto_cfg && p.splits == 1 && (cfg == 0 || cfg == 1 || cfg == 4 || cfg == 7 || cfg == 8)) {
    const int sp = gemm_sk_splits(p, a_mode, b_mode, epi);
    if (sp > 1) { cfg = 4; p.splits = sp; }
  }
  // 256 x 128 tiles of the ring kernel (cfg 8) where the 192-wide ones are a single partial round that the 128-wide ones
  // still cover in one: N = 768 at 32..42 row tiles (the LM stack at per-GPU batch 256: 160 -> 240 tiles on 256 CUs;
  // tools/tile128_bench.py, M = 10240: attention-out 27.9 -> 23.9 us, FFN-out 62.6 -> 51.6, FFN-in dgrad 59.9 -> 48.1, QKV
  // dgrad 45.5 -> 37.7; one row tile more and the second round costs 45 %)
  if (auto_cfg && cfg == 4 && a_mode == 0 && p.splits == 1 && p.batch <= 1 && p.N % 128 == 0 &&
      ((b_mode == 0 && epi == EPI_F32_RES && p.res != nullptr) || (b_mode == 1 && epi == EPI_BF16 && p.colsum == nullptr))) {
    const long c192 = (long)(p.M / 256) * (p.N / 192), c128 = (long)(p.M / 256) * (p.N / 128);
    if (c192 < 256 && c128 <= 256) cfg = 8;
  }
  // the ring kernel's residual epilogue always loads its residual operand: without one use the simple kernel
  if ((cfg == 3 || cfg == 4) && epi == EPI_F32_RES && p.res == nullptr) cfg = (p.N % 256 == 0) ? 2 : 1;
  if (cfg == 5 || cfg == 6) {   // 8-wave kernel with register-direct epilogue (gemm8w.hip), 256- / 192-wide tiles:
    const int ntw = (cfg == 5) ? 4 : 3;   // forward-form operands only
    if (!vault_gemm8w_supports(p, a_mode, b_mode, epi, ntw)) return -VAULT_EINVAL;
  } else if (p.aux_u8) {
    return -VAULT_EINVAL;       // the 8-bit gelu' exists in the 8-wave kernel's tile order only
  }
  // split-K of a non-accumulating epilogue: the ring kernel's 192-wide SK form with a workspace, nothing else
  if (p.splits > 1 && epi != EPI_F32_ATOMIC &&
      !(cfg == 4 && p.sk_ws != nullptr && a_mode == 0 && p.batch <= 1 && !p.split3 && !p.out_hm &&
        ((b_mode == 1 && epi == EPI_BF16) || (b_mode == 0 && epi == EPI_F32_RES && p.res != nullptr))))
    return -VAULT_EINVAL;
  if (cfg == 8 && (p.M % 256 || p.N % 128 || p.K % 64 || p.splits > 1 || p.batch > 1 || a_mode != 0 ||
                   !((b_mode == 0 && epi == EPI_F32_RES && p.res != nullptr) || (b_mode == 1 && epi == EPI_BF16))))
    return -VAULT_EINVAL;
  // head-major tensors (GemmParams::out_hm / a_hm): written by the 8-wave kernel's plain 16-bit epilogue, read as the ring
  // kernel's row-major-side A operand; anything else refuses
  if (p.out_hm && !((cfg == 5 || cfg == 6) && epi == EPI_BF16 && p.out_hm >= p.M && p.N % 64 == 0 && p.split3 == 0 &&
                    (long long)(p.N / 64) * p.out_hm * 128 < (1ll << 32)))
    return -VAULT_EINVAL;
  if (p.a_hm && !((cfg == 3 || cfg == 4 || cfg == 8) && a_mode == 0 && p.a_hm >= p.M)) return -VAULT_EINVAL;
  if (cfg < 0 || cfg > 8) return -VAULT_EINVAL;
  if (cfg == 7 && (a_mode != 0 || epi == EPI_F32_ATOMIC || p.splits > 1 || p.batch > 1)) return -VAULT_EINVAL;
  return cfg;
}

int vault_gemm_launch(const GemmParams& p_in, int a_mode, int b_mode, int epi, int cfg_in, hipStream_t st) {
  GemmParams p = p_in;
  const int cfg = vault_gemm_resolve(p, a_mode, b_mode, epi, cfg_in);
  if (cfg < 0) return -cfg;
  // (Round 6, VERDICT r05 item 1a, measured and NOT kept for the 8-wave kernel: its launches cut into the row panels of the full
  //  rounds + the remaining row panels on 128 x 128 tiles.  A round of the double-buffered kernel's small tiles takes as long as a
  //  round of 256 x 192 register-direct tiles at K = 768 (19.4 against 16 us at 348 / 256 tiles): FFN-in at 24 row panels 47.5 cut
  //  against 45.0 un-cut, attention-out dgrad at 93 panels 45 against 33 - profiles/r06_dev_tail_rows.txt.  The ring kernel's
  //  192-wide launches below do gain: their tail runs on the same kernel's 128-wide tiles.)
  if (cfg == 5 || cfg == 6) return vault_gemm8w_launch(p, epi, cfg == 5 ? 4 : 3, st);
  if (cfg == 4 && cfg_in < 0 && p.splits == 1 && a_mode == 0 && p.batch <= 1 && p.N % 128 == 0 && !p.split3 &&
      ((b_mode == 0 && epi == EPI_F32_RES && p.res != nullptr && p.drop_thresh == 0u) ||      // (dropout masks are keyed by the
       (b_mode == 1 && epi == EPI_BF16 && p.colsum == nullptr))) {                              //  element offset from `out`)
    // Tail of the last round (round 6) for the ring kernel's 192-wide launches (N = 768): the tiles come in
    // rounds of 256 and a mostly empty last round costs a whole one.  Whole rounds run on 256 x 192 tiles, the remaining ROW PANELS
    // on 256 x 128 tiles (cfg 8) where those are at most one round: two launches of one kernel family, the same arithmetic per
    // element (bit-identical), no kernel change.  tools/splitk_bench.py, us: 139 row panels (B = 192; 2 rounds + 66 tiles) 172 / 142 /
    // 103 cut against 197 / 168 / 123 un-cut (FFN-out forward / FFN-in dgrad / QKV dgrad); 93 panels (B = 128) 132 / 102 / 80
    // against 132 / 105 / 83.  Not with dropout in the residual epilogue (masks are keyed by the element offset from `out`).
    const int tn = p.N / 192, tn8 = p.N / 128, tm = p.M >> 8;
    const long tiles = (long)tm * tn, full = tiles / 256, rem = tiles - full * 256;
    const int R = (int)((full * 256) / tn), tail = tm - R;
    if (full >= 1 && rem > 0 && R >= 1 && tail >= 1 && (long)tail * tn8 <= 256) {
      const int r0 = R << 8;
      GemmParams q = p;
      q.M = r0; q.m_valid = std::min(p.m_valid, r0);
      int rc = vault_gemm256_launch(q, a_mode, b_mode, epi, 3, st);
      if (rc != 0 || p.m_valid <= r0) return rc;
      q = p;
      q.M = p.M - r0; q.m_valid = p.m_valid - r0;
      q.A = p.a_hm ? p.A + (size_t)r0 * 64 : p.A + (size_t)r0 * p.lda;      // (head-major A: rows inside every plane)
      if (epi == EPI_F32_RES) {
        q.out = reinterpret_cast<float*>(p.out) + (size_t)r0 * p.ldo;
        q.res = p.res + (size_t)r0 * p.ldo;
      } else {
        q.out = reinterpret_cast<h16*>(p.out) + (size_t)r0 * p.ldo;
      }
      return vault_gemm256_launch(q, a_mode, b_mode, epi, 2, st);
    }
  }
  if (cfg == 3) return vault_gemm256_launch(p, a_mode, b_mode, epi, 4, st);
  if (cfg == 4) return vault_gemm256_launch(p, a_mode, b_mode, epi, 3, st);
  if (cfg == 8) return vault_gemm256_launch(p, a_mode, b_mode, epi, 2, st);
  const int key = a_mode * 2 + b_mode;
#define VAULT_DISPATCH(AM, BMD)                                                             \
  switch (epi) {                                                                            \
    case EPI_BF16: return launch_modes<AM, BMD, EPI_BF16>(p, cfg, st);                      \
    case EPI_BF16_GELU: return launch_modes<AM, BMD, EPI_BF16_GELU>(p, cfg, st);            \
    case EPI_BF16_DGELU: return launch_modes<AM, BMD, EPI_BF16_DGELU>(p, cfg, st);          \
    case EPI_F32_RES: return launch_modes<AM, BMD, EPI_F32_RES>(p, cfg, st);                \
    case EPI_F32_PATCH: return launch_modes<AM, BMD, EPI_F32_PATCH>(p, cfg, st);            \
    case EPI_F32_ATOMIC: return launch_modes<AM, BMD, EPI_F32_ATOMIC>(p, cfg, st);          \
    default: return VAULT_EINVAL;                                                           \
  }
  switch (key) {
    case 0: VAULT_DISPATCH(0, 0)
    case 1: VAULT_DISPATCH(0, 1)
    case 3: VAULT_DISPATCH(1, 1)
    default: return VAULT_EINVAL;
  }
#undef VAULT_DISPATCH
}
