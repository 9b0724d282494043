// Internal (C++) interface of the bf16 MFMA GEMM.  The public C-ABI wrapper is vault_gemm in
// include/vault_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

enum GemmEpi {
  EPI_BF16 = 0,        // out_bf16 = acc (+ bias)
  EPI_BF16_GELU = 1,   // out_bf16 = gelu(acc + bias) ; out2_bf16 (optional) = gelu'(acc + bias)
  EPI_BF16_DGELU = 2,  // out_bf16 = acc * aux_bf16   (aux = the gelu' stored by EPI_BF16_GELU)
  EPI_F32_RES = 3,     // out_f32 = dropout(acc + bias) + res_f32
  EPI_F32_PATCH = 4,   // out_f32[rowmap(m)] = acc + addtab[m % rpg]   (patch embedding into the fused sequence)
  EPI_F32_ATOMIC = 5,  // out_f32 += acc   (wgrad; split-K partials by float atomics)
  EPI_BF16_GELU_INF = 6,  // internal (ring kernel): EPI_BF16_GELU without the gelu' output, picked by the launcher when out2 == null
  EPI_BF16_GELU_U8 = 7,   // internal (8-wave kernel): EPI_BF16_GELU with the gelu' output as 8-bit values in tile-native order
  EPI_BF16_DGELU_U8 = 8,  // internal (8-wave kernel): EPI_BF16_DGELU reading that 8-bit gelu'
};

struct GemmParams {
  const h16* A;
  const h16* B;
  int M, N, K;      // output M x N, contraction K; M % BM == N % BN == K % 64 == 0 (buffers padded)
  int lda, ldb;     // leading dimensions in elements (see A_MODE / B_MODE in gemm.hip)
  int m_valid;      // rows >= m_valid are never stored
  int splits;       // split-K factor (grid.z)
  int accumulate;   // EPI_F32_ATOMIC: 1 = add into out even when splits == 1
  void* out;
  int ldo;
  void* out2;
  const float* bias;
  const float* res;
  const h16* aux;
  const float* addtab;
  int split3;       // bf16 epilogues: store [hi | lo | hi] (row stride ldo = 3N) for the split-bf16 precise path
  float* colsum;    // bf16 epilogues: += column sums of the stored values (bias gradient), or null
  int rpg, gstride, goff;
  // inverted dropout on (acc + bias) for EPI_F32_RES; thresh == 0 disables it
  uint32_t drop_thresh, drop_seed, drop_stream;
  float drop_scale;
  int persist;      // bit 0: ring kernel hands its work items out dynamically (ticket counters); bit 1: double-buffered
                    // kernel launches one block per tile instead of its persistent grid.  3 = both: for GEMMs that share
                    // the GPU with another kernel (RCCL collectives of a data-parallel step)
  int gn;           // n-tiles per raster group (set by the launcher: B panel of a group stays L2-resident)
  // batched weight gradients (EPI_F32_ATOMIC): `batch` independent GEMMs of one shape in one launch, problem b at
  // A + b * batch_a, B + b * batch_b (bf16 elements), out + b * batch_o (floats).  Double-buffered kernel: grid.z =
  // batch * splits; ring kernel: (problem, split, tile) work items, problem-major.  batch <= 1: a single GEMM.
  int batch;
  long long batch_a, batch_b, batch_o;
  int aux_u8;       // 8-wave kernel only: gelu' (out2 of EPI_BF16_GELU, aux of EPI_BF16_DGELU) is the 8-bit tile-native form
  // Head-major 16-bit tensors (the attention kernels' qkv / dqkv, vault_attn_args.qkv_hm): [N / 64][R][64] instead of [R][N].
  //   out_hm = R: the EPI_BF16 output of the 8-wave kernel is written in that layout (QKV forward);
  //   a_hm = R:   the A operand of the ring kernel's (0, *) modes is read from it ([K / 64][R][64]: a K tile = one plane,
  //               rows contiguous - the QKV data gradient's dqkv); 0 = row-major.
  int out_hm, a_hm;
  // MXFP8 operands (gemm_mx8.hip, the 8-wave kernel's MX form): E8M0 block scales [rows][lds_*] of A and B (one byte per 32
  // consecutive k), null for the 16-bit kernels
  const uint8_t* a_scale;
  const uint8_t* b_scale;
  int lds_a, lds_b;
  // MXFP8 image of the 16-bit output (8-wave kernel's MXFP8 form, 256-wide tiles, GELU epilogue): e4m3 bytes [M][N] + one E8M0 scale per 32
  // consecutive columns [M][N / 32], quantised from the ROUNDED 16-bit values (= what vault_quant_mxfp8 makes of `out`) - the
  // next forward Linear's A operand without a pass over `out`.  Null = off.
  void* out_q;
  void* out_scale;
  // Split-K with the reduction inside the launch (ring kernel, gemm256.hip SK: (0,1) EPI_BF16 and (0,0) EPI_F32_RES, 192- / 256-wide
  // tiles, splits > 1): the caller's workspace - GEMM_SK_COUNTER_BYTES of tile counters (ZERO before the first launch; every
  // launch leaves them zero), then one f32 slab of 256 x tile-width floats per (tile, split).  Not shared by launches that can
  // run at the same time.  Null: splits > 1 is refused for these epilogues.
  void* sk_ws;
  long long sk_bytes;
  // Grouped weight gradients (ring kernel, (1,1) operand modes, EPI_F32_ATOMIC): ONE launch over up to three segments of
  // DIFFERENT weight-gradient kinds that share the contraction (the tokens) and therefore the cost per 256 x 256 tile: the
  // work list is their concatenation, so a launch can be sized to exactly one round of the 256 CUs (e.g. the 216 FFN-out
  // tiles of six layers + 40 attention-out tiles) instead of one partial round per kind.  nseg == 0: a plain launch.
  int nseg;
  struct Seg {
    const h16* A;                     // dY of the kind's first problem, [tokens][lda]
    const h16* B;                     // X of the kind's first problem, [tokens][ldb]
    float* out;                       // dW of the kind's first problem, [M][ldo]
    int tiles_n, tiles;               // 256-wide tiles per row of tiles / per problem
    int lda, ldb, ldo;
    int first, count;                 // items [first, first + count) of the kind's (problem-major, tile-minor) numbering
    long long batch_a, batch_b, batch_o;   // element strides between the kind's problems (layers)
    int a_hm;                         // > 0: dY is head-major [M / 64][a_hm rows][64] (the QKV kind's dqkv)
  } seg[3];
};

constexpr int GEMM_SK_COUNTER_BYTES = 16384;                       // 4096 tile counters in front of the slabs
constexpr int GEMM_SK_MAX_TILES = GEMM_SK_COUNTER_BYTES / 4;

#ifdef __HIPCC__
// Linear block id -> (tile_m, tile_n).  Blocks with equal id % 8 share an XCD (and its 4 MiB L2): each XCD
// gets a contiguous run of the raster order; the raster walks column GROUPS of `gn` n-tiles, all m-tiles of a
// group before the next group, so that a group's B panel (gn x BN x K) stays L2-resident while the A row
// panels stream through once per group.
// XCD-contiguous renumbering: ids with equal id % 8 run on one XCD; give each XCD a contiguous run of `n` items
__device__ __forceinline__ int gemm_xcd_contiguous(int n, int id) {
  const int q = n >> 3, r = n & 7, xcd = id & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}
// raster position -> tile: column groups of `gn` n-tiles, all m-tiles of a group before the next group
__device__ __forceinline__ void gemm_raster(int wg, int tiles_m, int tiles_n, int gn, int& tile_m, int& tile_n) {
  const int per_group = tiles_m * gn;
  const int grp = wg / per_group;
  const int rem = wg - grp * per_group;
  const int gw = min(gn, tiles_n - grp * gn);
  tile_m = rem / gw;
  tile_n = grp * gn + (rem - tile_m * gw);
}
__device__ __forceinline__ void gemm_tile_of_block(int nwg, int id, int tiles_m, int tiles_n, int gn, int& tile_m,
                                                   int& tile_n) {
  gemm_raster(gemm_xcd_contiguous(nwg, id), tiles_m, tiles_n, gn, tile_m, tile_n);
}
#endif

int vault_gemm_launch(const GemmParams& p, int a_mode, int b_mode, int epi, int cfg, hipStream_t st);
int vault_gemm256_grouped_launch(const GemmParams& p, hipStream_t st);
