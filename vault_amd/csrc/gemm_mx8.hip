// MXFP8 (OCP microscaling: e4m3 elements, one E8M0 scale per 32 consecutive k) forward GEMM for gfx950 and the
// quantiser that produces its operands - BASELINE config "fp8 MFMA forward, bf16 backward" (SURVEY 8d config 5).
//
//   C[M,N] = A[M,K] . W[N,K]^T     A, W: e4m3 bytes, K contiguous; As [M][K/32], Ws [N][K/32]: E8M0 bytes
//   v_mfma_scale_f32_16x16x128_f8f6f4: the block scales are applied by the matrix pipe (2x the bf16 rate)
//
// Only the forward orientation exists (backward stays bf16 on the saved bf16 activations).  Structure as
// gemm.hip: 256 x 256 x 128 tiles (a tile row is 128 BYTES, so the LDS images, the 16-byte global_load_lds
// staging and the epilogue are those of the bf16 kernel), double buffered, persistent grid, 8 waves (2 x 4).
// Operand map of the instruction (found and checked with exact integer data, tests/test_gpu_mx8.py): lane
// (l15 = lane % 16, g = lane / 16) supplies row l15 and, in its eight registers, k = 16 g .. 16 g + 15 followed by
// k = 64 + 16 g .. 64 + 16 g + 15 of the 128-deep step (two stacked 64-deep halves, NOT 32 consecutive k), while its
// scale register carries the E8M0 byte of the 32 CONSECUTIVE k of block g (k = 32 g .. 32 g + 31).  The fragment is
// therefore the 16-byte chunks g and 4 + g of the row image: the two reads of the bf16 kernel's k-steps, same
// XOR swizzle, conflict-free.
#include <algorithm>
#include "common.h"
#include "gemm.h"
#include "gemm_epi.h"
#include "../../include/vault_hip.h"

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BKB = 128;   // k per tile = bytes per tile row

// swizzle key of an image row: 16-byte chunk c of row r is stored at position c ^ key(r)   (as gemm.hip)
__device__ __forceinline__ int mx8_key(int row) { return ((row >> 1) & 3) << 1; }

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_mx8_kernel(const GemmParams p, const uint8_t* __restrict__ a_scale,
                                                               const uint8_t* __restrict__ b_scale, int lds_a, int lds_b) {
  H16_SATURATE();
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16;
  constexpr int TN = BN / WN / 16;
  constexpr int A_BYTES = BM * BKB;
  constexpr int B_BYTES = BN * BKB;
  constexpr int SC_BYTES = (BM + BN) * 4;          // one dword (4 block scales) per row and tile
  constexpr int STAGE = A_BYTES + B_BYTES + SC_BYTES;
  constexpr int NIA = A_BYTES / 1024 / NW;
  constexpr int NIB = B_BYTES / 1024 / NW;
  static_assert(NIA >= 1 && NIB >= 1 && (BM + BN) % 64 == 0, "tile too small for the wave count");
  constexpr int NSC = (BM + BN) / 64;              // 4-byte global_load_lds wave-instructions for the scales of a tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WN) * (BM / WM);
  const int wn0 = (wave % WN) * (BN / WN);
  const uint8_t* Aq = reinterpret_cast<const uint8_t*>(p.A);
  const uint8_t* Bq = reinterpret_cast<const uint8_t*>(p.B);

  const int total_tiles = (p.M / BM) * (p.N / BN);
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    int tile_m, tile_n;
    gemm_tile_of_block(total_tiles, tile, p.M / BM, p.N / BN, p.gn, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk = p.K / BKB;

    // ---- per-lane staging sources (bytes)
    const uint8_t* a_src[NIA];
    const uint8_t* b_src[NIB];
    {
      const int r8 = lane >> 3, pos = lane & 7;
      const int c = pos ^ mx8_key(r8);
#pragma unroll
      for (int i = 0; i < NIA; ++i) a_src[i] = Aq + (size_t)(m0 + 8 * (wave * NIA + i) + r8) * p.lda + c * 16;
#pragma unroll
      for (int i = 0; i < NIB; ++i) b_src[i] = Bq + (size_t)(n0 + 8 * (wave * NIB + i) + r8) * p.ldb + c * 16;
    }
    auto stage = [&](int buf, int t) {
      char* sa = smem + buf * STAGE + wave * NIA * 1024;
      char* sb = smem + buf * STAGE + A_BYTES + wave * NIB * 1024;
#pragma unroll
      for (int i = 0; i < NIA; ++i) glds16(a_src[i] + (size_t)t * BKB, sa + i * 1024);
#pragma unroll
      for (int i = 0; i < NIB; ++i) glds16(b_src[i] + (size_t)t * BKB, sb + i * 1024);
      // scales: rows 0..BM-1 of A then 0..BN-1 of W, one dword each (K/32 is a multiple of 4: K % 128 == 0)
      char* ss = smem + buf * STAGE + A_BYTES + B_BYTES;
      for (int i = wave; i < NSC; i += NW) {
        const int row = i * 64 + lane;
        const uint8_t* src = row < BM ? a_scale + (size_t)(m0 + row) * lds_a + t * 4
                                      : b_scale + (size_t)(n0 + row - BM) * lds_b + t * 4;
        __builtin_amdgcn_global_load_lds(GLB_PTR(void, src), LDS_PTR(void, ss + i * 256), 4, 0, 0);
      }
    };

    // ---- per-lane fragment read offsets
    const int g = lane >> 4, l15 = lane & 15;
    const int ck = (g ^ mx8_key(l15)) << 4;            // chunk g: k = 16 g .. of the tile
    const int ck1 = ((4 + g) ^ mx8_key(l15)) << 4;     // chunk 4 + g: k = 64 + 16 g ..
    const int a_row = (wm0 + l15) * 128, b_row = (wn0 + l15) * 128;
    const int sh = 8 * g;                              // this lane's block scale inside the row's dword

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) stage(0, 0);
    for (int t = 0; t < nk; ++t) {
      __syncthreads();  // tile t has landed (vmcnt(0) precedes the barrier) and buffer (t+1)&1 is free
      if (t + 1 < nk) stage((t + 1) & 1, t + 1);
      const char* As = smem + (t & 1) * STAGE;
      const char* Bs = As + A_BYTES;
      const char* Ss = Bs + B_BYTES;
      i32x8 af[TM], bfr[TN];
      int sa[TM], sb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const i32x4 lo = *LDS_PTR(const i32x4, As + a_row + i * 16 * 128 + ck);
        const i32x4 hi = *LDS_PTR(const i32x4, As + a_row + i * 16 * 128 + ck1);
        af[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        sa[i] = (*LDS_PTR(const int, Ss + (wm0 + i * 16 + l15) * 4) >> sh) & 0xff;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const i32x4 lo = *LDS_PTR(const i32x4, Bs + b_row + j * 16 * 128 + ck);
        const i32x4 hi = *LDS_PTR(const i32x4, Bs + b_row + j * 16 * 128 + ck1);
        bfr[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        sb[j] = (*LDS_PTR(const int, Ss + (BM + wn0 + j * 16 + l15) * 4) >> sh) & 0xff;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bfr[j], acc[i][j], 0, 0, 0, sa[i], 0, sb[j]);
    }

    gemm_epilogue<TM, TN, EPI, WM, WN>(acc, p, smem, m0, n0, wm0, wn0, wave, lane);
    __syncthreads();   // epilogue scratch (LDS) is free again before the next tile stages into it
  }
}

template <int EPI>
int launch_mx8(const GemmParams& p, const uint8_t* a_scale, const uint8_t* b_scale, hipStream_t st) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
  if (p.M % BM || p.N % BN || p.K % BKB) return VAULT_EINVAL;
  constexpr int LDS = 2 * ((BM + BN) * BKB + (BM + BN) * 4);
  auto kern = gemm_mx8_kernel<BM, BN, WM, WN, EPI>;
  static bool attr_done_dev[64] = {}; int attr_dev = 0; (void)hipGetDevice(&attr_dev); bool& attr_done = attr_done_dev[(attr_dev >= 0 && attr_dev < 64) ? attr_dev : 0];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  GemmParams q = p;
  q.splits = 1;
  q.gn = (p.gn > 0) ? std::min(p.gn, p.N / BN) : (p.K <= 1024 ? std::min(p.N / BN, 4) : p.N / BN);
  const int total = (p.M / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(std::min(total, 256)), dim3(WM * WN * 64), LDS, st, q, a_scale, b_scale, p.K / 32, p.K / 32);
  return (int)hipGetLastError();
}

// ---- quantiser: 4 lanes share a block of 32 (8 elements = 16 bytes of bf16 each); the arithmetic is common.h mx8_quant8

__global__ __launch_bounds__(256) void quant_mx8_kernel(const h16* __restrict__ src, int64_t rows, int K, int ld,
                                                        uint8_t* __restrict__ q, uint8_t* __restrict__ scale) {
  const int64_t nchunk = rows * (K / 8);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nchunk; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / (K / 8);
    const int c8 = (int)(idx - row * (K / 8));
    const h16x8 v = *reinterpret_cast<const h16x8*>(src + row * ld + c8 * 8);
    float x[8];
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[e] = (float)v[e]; amax = fmaxf(amax, fabsf(x[e])); }
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    int e8;
    const uint2 w = mx8_quant8(x, amax, e8);
    *reinterpret_cast<uint2*>(q + row * K + c8 * 8) = w;
    if ((c8 & 3) == 0) scale[row * (K / 32) + (c8 >> 2)] = (uint8_t)e8;
  }
}

}  // namespace

extern "C" int vault_quant_mxfp8(const void* src_bf16, long long rows, int K, int ld_src, void* dst_q, void* dst_scale,
                                 void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (src_bf16 == nullptr || dst_q == nullptr || dst_scale == nullptr) return VAULT_EINVAL;
  if (rows <= 0 || K <= 0 || (K & 31) || (ld_src & 7) || ld_src < K) return VAULT_EINVAL;
  const long long nchunk = rows * (K / 8);
  const int blocks = (int)std::min<long long>((nchunk + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(quant_mx8_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const h16*>(src_bf16),
                     (int64_t)rows, K, ld_src, reinterpret_cast<uint8_t*>(dst_q), reinterpret_cast<uint8_t*>(dst_scale));
  return (int)hipGetLastError();
}

bool vault_gemm8w_mx_supports(const GemmParams& p, int epi, int ntw);
int vault_gemm8w_mx_launch(const GemmParams& p, int epi, int ntw, hipStream_t st);

// p.cfg semantics of the C entry point (vault_gemm_args.cfg): -1 = automatic (the 8-wave kernel's MXFP8 form where it takes
// the shape, 256-wide tiles unless only the 192-wide ones divide N), 0 = the simple double-buffered kernel of this file,
// 5 / 6 = the 8-wave form with 256- / 192-wide tiles (EINVAL where it does not take the call)
// argument checks + kernel choice: 0 (simple kernel), 5 / 6 (8-wave form; q = the parameters it runs on), or -VAULT_EINVAL.
// The scale pointers only have to be non-null (a plan does not dereference them).
int vault_gemm_mx8_resolve(const GemmParams& p, const void* a_scale, const void* b_scale, int epi, int cfg, GemmParams& q) {
  if (p.A == nullptr || p.B == nullptr || p.out == nullptr || a_scale == nullptr || b_scale == nullptr) return -VAULT_EINVAL;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.lda != p.K || p.ldb != p.K || (p.ldo & 7)) return -VAULT_EINVAL;
  if (cfg != -1 && cfg != 0 && cfg != 5 && cfg != 6) return -VAULT_EINVAL;
  q = p;
  q.a_scale = reinterpret_cast<const uint8_t*>(a_scale); q.b_scale = reinterpret_cast<const uint8_t*>(b_scale);
  q.lds_a = p.K / 32; q.lds_b = p.K / 32;
  if (q.m_valid <= 0) q.m_valid = q.M;
  if (cfg != 0) {
    // automatic tile width: 256 unless only 192 divides N or the 192-wide tiles fill the last round of CUs much better
    // (N = 768 at 185 row tiles: 740 tiles = 2.9 rounds at 3/4 of the cost against 555 = 2.2 rounds, as vault_gemm chooses);
    // the output image (out_q) and a requested cfg fix it
    auto eff = [](long tiles) { return (double)tiles / (double)(((tiles + 255) / 256) * 256); };
    int ntw = (cfg == 6) ? 3 : ((cfg == 5 || p.N % 256 == 0) ? 4 : 3);
    if (cfg < 0 && ntw == 4 && p.N % 192 == 0 && p.out_q == nullptr && !p.aux_u8 &&
        vault_gemm8w_mx_supports(q, epi, 3) &&
        0.9 * eff((long)(p.M / 256) * (p.N / 192)) > eff((long)(p.M / 256) * (p.N / 256)))
      ntw = 3;
    if (vault_gemm8w_mx_supports(q, epi, ntw)) return ntw == 4 ? 5 : 6;
    if (cfg > 0) return -VAULT_EINVAL;
  }
  if (p.aux_u8 || p.out_hm || p.out_q != nullptr) return -VAULT_EINVAL;   // (8-wave form only)
  if (p.M % 256 || p.N % 256 || p.K % BKB) return -VAULT_EINVAL;
  if (epi != EPI_BF16 && epi != EPI_BF16_GELU && epi != EPI_F32_RES) return -VAULT_EINVAL;
  return 0;
}

int vault_gemm_mx8_launch(const GemmParams& p, const void* a_scale, const void* b_scale, int epi, int cfg, hipStream_t st) {
  GemmParams q;
  const int k = vault_gemm_mx8_resolve(p, a_scale, b_scale, epi, cfg, q);
  if (k < 0) return -k;
  if (k != 0) return vault_gemm8w_mx_launch(q, epi, k == 5 ? 4 : 3, st);
  const uint8_t* as = reinterpret_cast<const uint8_t*>(a_scale);
  const uint8_t* bs = reinterpret_cast<const uint8_t*>(b_scale);
  switch (epi) {
    case EPI_BF16: return launch_mx8<EPI_BF16>(p, as, bs, st);
    case EPI_BF16_GELU: return launch_mx8<EPI_BF16_GELU>(p, as, bs, st);
    case EPI_F32_RES: return launch_mx8<EPI_F32_RES>(p, as, bs, st);
    default: return VAULT_EINVAL;
  }
}
