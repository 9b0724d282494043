// 256x256x64 (and 256x192x64) bf16 MFMA GEMM for the forward and data-gradient Linears (A [M][K], B [N][K], both
// K-contiguous), 8 waves = two per SIMD, accumulators stored STRAIGHT FROM REGISTERS: built for the short contractions
// of the path (K = 768: 12 K tiles), where the ring kernel (gemm256.hip, one wave per SIMD) spends a third of a tile's
// time in its epilogue - 16 LDS write -> read round trips to transpose the accumulators (~8 us against a 16 us main loop)
// and, for the GELU forms, ~10 us of VALU work issued by a single wave per SIMD (4 cycles per instruction; two waves
// share a SIMD at 2).
//
// Accumulator layout = store layout: the MFMA runs with SWAPPED operands, D' = W_tile . X_tile^T, so lane (l15, g)
// holds FOUR CONSECUTIVE output columns (n = 4 g + r) of row m = l15 of a 16 x 16 tile instead of four rows of one
// column.  For bf16 outputs the weight rows are additionally PERMUTED while they are staged into LDS (the per-lane
// source address of global_load_lds is free): LDS row 16 nt + 4 g + r of a wave's 64 holds weight row
// 32 (nt >> 1) + 8 g + 4 (nt & 1) + r, so that the tiles nt = 2 j, 2 j + 1 of a lane are 8 consecutive columns = one
// 16-byte store, and the four lanes g of a row write 64 contiguous bytes per instruction - the store shape of the
// LDS-transposed epilogue, from registers (192-wide tiles: 48 columns per wave = one such pair + a 4-column tail, an
// 8-byte store).  f32 outputs keep the natural order (4 floats per lane and tile = 16 bytes, 64 contiguous bytes per
// row and instruction).
//
// Pipeline: two LDS stages of one K tile each (A 256 rows x 128 B + B 256 / 192 rows x 128 B = 64 / 56 KiB); K tile t
// lives in stage t & 1; every wave stages 4 + NTW pieces (1 KiB global_load_lds each) per K tile.  ONE barrier per K tile:
//   k-step 0   8 x NTW MFMAs on FA0 / FB0 (in registers since the previous K tile), the A fragments of k-step 1 read
//              meanwhile
//   mid of t   wait vmcnt(0) [tile t + 1 landed: its pieces are the only staging loads in flight], lgkmcnt(0), barrier:
//              stage (t + 1) & 1 is readable, stage t & 1 is free
//   k-step 1   8 x NTW MFMAs on FA1 / FB1 with, spread between them, the staging pieces of tile t + 2 (into stage t & 1)
//              and the reads of tile t + 1's FA0 / FB0 (then FB1): the next K tile starts on registers
// The K tiles of consecutive work items form ONE sequence: the loads run ahead across the item boundary, so the next
// item's first tile has landed and its second is in flight before the epilogue's stores are issued; the mid wait of an
// item's first K tile then allows for the stores certain to sit behind that tile (VMEM operations retire in order).
// Past the last item the loads re-issue valid tiles into stages nobody reads (uniform counts), drained at exit.
// A variant with two independent 4-wave workgroups per CU (256 x 128 tiles, 80 KiB each: the epilogue of one beside the
// main loop of the other) was built first and measured (profiles/r02_dev_paired_wg_*.txt): per-tile fixed cost 2.7 us,
// but 1.5 x the staging bytes per FLOP made the global -> LDS path the bound (main loop 1.05 against 1.56 PFLOP/s with
// the staging loads ablated): the block tile has to stay 256 wide.
//
// Replaces (with gemm256.hip / gemm.hip): every nn.Linear forward of HF:models/vilt/modeling_vilt.py:303-414 and
// HF:models/roberta/modeling_roberta.py:222-398, and - on the transposed bf16 weight shadow - their data gradients.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "gemm.h"

#ifndef W8_ABLATE
#define W8_ABLATE 0
#endif
#ifndef W8_STORE_MOD
#define W8_STORE_MOD ""   // development: cache policy of the bf16 output stores (" nt", " sc1", ...)
#endif

// Diagnostic build only (-DW8_STAMP=1, tools/clock_stamp.py): see gemm256.hip R256_STAMP.
#ifndef W8_STAMP
#define W8_STAMP 0
#endif
#if W8_STAMP
__device__ unsigned long long g_w8_stamp[256 * 2];
extern "C" int vault_debug_w8_stamps(unsigned long long* out512) {
  return (int)hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_w8_stamp), sizeof(g_w8_stamp));
}
#endif

namespace {

constexpr int W8_A_BYTES = 32768;   // 256 rows x 64 k x 2 B
constexpr int W8_ZERO_BIAS = 8192;  // floats: widest output the bias-free form takes
__device__ float g_w8_zero_bias[W8_ZERO_BIAS];   // zero-initialised, never written

// global_load_lds through inline asm: hipcc keeps no record of an asm LDS-DMA, so it neither answers the fragment
// reads with `s_waitcnt vmcnt(0)` (it cannot tell which stage a DMA fills) nor counts them into waits of its own.
// All ordering is by the counted waits below; "memory" keeps the LDS reads on their side of the barriers.
// sbase: uniform 64-bit address, voff: per-lane byte offset, lds: uniform LDS byte address (lane l lands at + 16 l).
__device__ __forceinline__ void w8_glds16(const char* sbase, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}

// the 4-byte form (MXFP8: one dword = the four E8M0 block scales of a row's 128-deep K tile; lane l lands at + 4 l)
__device__ __forceinline__ void w8_glds4(const char* sbase, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
               :: "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef int w8_i32x8 __attribute__((ext_vector_type(8)));
typedef int w8_i32x4 __attribute__((ext_vector_type(4)));

// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), result in every lane of the row
__device__ __forceinline__ float w8_row16_sum(float t) {
#define W8_DPP(X, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, X), CTRL, 0xF, 0xF, true))
  t += W8_DPP(t, 0xB1);    // quad_perm [1,0,3,2]
  t += W8_DPP(t, 0x4E);    // quad_perm [2,3,0,1]
  t += W8_DPP(t, 0x141);   // row_half_mirror
  t += W8_DPP(t, 0x140);   // row_mirror
#undef W8_DPP
  return t;
}

// hipcc pads the "VALU writes an SGPR -> VMEM reads it" hazard (5 wait states on gfx9) for its own instructions only: an
// asm load / store / atomic whose SGPR base hipcc has parked in a VGPR lane (register pressure) gets the v_readlane_b32 that
// brings it back directly in front of it and then runs on a stale base (seen: garbage bias in the first two of four back-to-back
// loads; tools/isa_hazard.py finds the pattern in hipcc's -S output).  The first VMEM statement of a group that may follow
// such a reload starts with this pad; statements behind other asm statements of the same base are far enough.
#define W8_SGPR_PAD "s_nop 4\n\t"
#define W8_WAITBAR(N) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(N) : "memory")
#define W8_LGKBAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// `s_waitcnt vmcnt(N)` with N a compile-time expression of an unrolled loop variable (0 .. 24)
#define W8_VMCNT_CASES(X)                                                                                          \
  switch (X) {                                                                                                     \
    W8_C(0) W8_C(1) W8_C(2) W8_C(3) W8_C(4) W8_C(5) W8_C(6) W8_C(7) W8_C(8) W8_C(9) W8_C(10) W8_C(11) W8_C(12)     \
    W8_C(13) W8_C(14) W8_C(15) W8_C(16) W8_C(17) W8_C(18) W8_C(19) W8_C(20) W8_C(21) W8_C(22) W8_C(23) W8_C(24)    \
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;                                               \
  }
#define W8_C(N) case N: asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); break;

// A3 (round 4): THREE slots for the A image, two for B (3 x 32 + 2 x 32 / 24 KiB = 160 / 144 KiB).  The activation panel is the
// operand that misses the L2 (the weight panels of a column group stay there: launch8w's raster), and with one K tile of
// look-ahead an HBM round trip does not fit under a K tile's 64 MFMAs per wave: A tile t + 3 is requested while tile t is
// multiplied (B tile t + 2, as before), the B pieces first, so that the mid wait leaves the 4 A pieces of tile t + 2 in flight.
//
// MX (round 5): the same kernel on MXFP8 operands (e4m3 bytes + one E8M0 scale per 32 consecutive k, gemm_mx8.hip's formats):
// a tile row is still 128 BYTES, so a K tile is 128 deep, and v_mfma_scale_f32_16x16x128_f8f6f4 takes as its eight operand
// registers exactly the two fragments of the 16-bit form's k-steps (16-byte chunks g and 4 + g of the row image, gemm_mx8.hip
// header) - same LDS images, same staging, same swizzle, same accumulator layout and therefore the same epilogues, half the
// K tiles.  One MFMA set per K tile (8 x NTW instructions of twice the cycles); the two halves of the pipeline are the ROW
// halves of the tile instead of its k-steps:
//   rows 0-3   4 x NTW MFMAs on XA[0..3] / XB (in registers since the previous K tile), the fragments XA[4..7] read meanwhile
//   mid        as above
//   rows 4-7   4 x NTW MFMAs on XA[4..7] / XB with tile t + 2's staging pieces and tile t + 1's XA[0..3] between them; XB
//              of tile t + 1 behind the last MFMA
// The block scales ride along: per K tile and wave ONE 4-byte LDS-DMA piece (a dword per row: waves 0-3 the 256 activation
// rows, waves 4-7 the weight rows in the order of the permuted LDS image), read back as single bytes (byte g of the lane's
// rows).  Two stages (the third A slot has no room beside the scales: 2 x 66 KiB).
template <int EPI, int NTW, bool A3 = false, bool MX = false>
__global__ __launch_bounds__(512, 2) void gemm8w_kernel(const GemmParams p) {
  static_assert(!(MX && A3), "the MXFP8 form is two-stage");
  H16_SATURATE();
#if W8_STAMP
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr bool F32OUT = (EPI == EPI_F32_RES);
  constexpr int BN = 64 * NTW;                       // block tile width
  constexpr int B_BYTES = BN * 128;
  constexpr int SC_BYTES = MX ? 2048 : 0;             // 512 dwords: 256 activation rows, up to 256 weight rows
  constexpr int ESZ = MX ? 1 : 2;                      // bytes per operand element
  constexpr int STAGE = W8_A_BYTES + B_BYTES + SC_BYTES;
  // LDS layout: [A0 B0][A1 B1], or (A3) [A0][A1][A2][B0][B1], or (MX) [A0 B0 S0][A1 B1 S1]
  auto a_base = [](int sa) -> int { return A3 ? sa * W8_A_BYTES : sa * STAGE; };
  auto b_base = [](int sb) -> int { return A3 ? 3 * W8_A_BYTES + sb * B_BYTES : sb * STAGE + W8_A_BYTES; };
  auto s_base = [](int sb) -> int { return sb * STAGE + W8_A_BYTES + B_BYTES; };
  constexpr int NPC = 4 + NTW + (MX ? 1 : 0);        // staging pieces per wave and K tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_m = p.M >> 8, tiles_n = p.N / BN, nwork = tiles_m * tiles_n;
  const int nk = p.K >> (MX ? 7 : 6);
  const int G = (int)gridDim.x;

  // ---- staging sources: scalar base (per work item and K tile) + scalar piece offset + a per-lane offset
  const int r8 = lane >> 3;
  const int csw = (lane & 7) ^ (((r8 >> 1) & 3) << 1);     // source chunk of LDS position lane & 7 (XOR swizzle by row)
  const uint32_t a_voff = (uint32_t)(r8 * p.lda * ESZ + csw * 16);
  // B: LDS row 8 j + r8 of piece j  <-  weight row (header); the piece part is scalar, the row-in-piece part per lane
  const uint32_t b_voff = (uint32_t)((F32OUT ? r8 : (8 * (r8 >> 2) + (r8 & 3))) * p.ldb * ESZ + csw * 16);
  const uint32_t b_voff2 = (uint32_t)((4 * (r8 >> 2) + (r8 & 3)) * p.ldb * ESZ + csw * 16);   // NTW == 3: the 4-column tail tile
  const size_t a_piece = (size_t)8 * p.lda * ESZ;          // bytes between consecutive A pieces (8 rows)
  const size_t b_row = (size_t)p.ldb * ESZ;
  const uint32_t lds0 = (uint32_t)(size_t)LDS_PTR(char, smem);   // LDS byte address of the dynamic segment

  // per work item: base of its A row panel / B column panel (the divisions of the raster run once per item)
  auto bases_of = [&](int w, const char*& ab, const char*& bb, int& m0, int& n0) {
    int tm, tn;
    gemm_raster(gemm_xcd_contiguous(nwork, w), tiles_m, tiles_n, p.gn, tm, tn);
    m0 = tm << 8; n0 = tn * BN;
#if W8_ABLATE == 3 || W8_ABLATE == 4   // development: every tile loads the first row / column panel (operands stay in L2)
    ab = reinterpret_cast<const char*>(p.A);
    bb = reinterpret_cast<const char*>(p.B);
#else
    ab = reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * ESZ;
    bb = reinterpret_cast<const char*>(p.B) + (size_t)n0 * p.ldb * ESZ;
#endif
  };
  // MX: this wave's scale piece - dword (wave < 4 ? 64 wave : 256 + 64 (wave - 4)) + lane of the stage's scale block; waves 0-3
  // take the activation rows in order, waves 4-7 the weight row that pieceB stages into LDS row R = 64 (wave - 4) + lane
  uint32_t s_voff = 0;
  if constexpr (MX) {
    if (wave < 4) {
      s_voff = (uint32_t)((wave * 64 + lane) * p.lds_a);
    } else {
      int R = (wave - 4) * 64 + lane;
      if (R >= BN) R -= 64;                         // (192-wide tiles: wave 7 re-loads rows 128..191 into the spare dwords)
      const int j = R >> 3, q8 = R & 7;
      int wrow;
      if constexpr (F32OUT) {
        wrow = R;
      } else if constexpr (NTW == 4) {
        wrow = (j >> 3) * 64 + ((j >> 2) & 1) * 32 + (j & 1) * 16 + ((j >> 1) & 1) * 4 + 8 * (q8 >> 2) + (q8 & 3);
      } else {
        const int wcol = j / 6, jj = j - wcol * 6, nt = jj >> 1;
        wrow = (nt < 2) ? wcol * 48 + (jj & 1) * 16 + nt * 4 + 8 * (q8 >> 2) + (q8 & 3)
                        : wcol * 48 + 32 + (jj & 1) * 8 + 4 * (q8 >> 2) + (q8 & 3);
      }
      s_voff = (uint32_t)(wrow * p.lds_b);
    }
  }
  const uint32_t s_lds = (uint32_t)((wave < 4 ? wave * 64 : 256 + (wave - 4) * 64) * 4);
  auto scales_of = [&](int m0_, int n0_) -> const char* {   // scale panel of a work item, this wave's side
    return wave < 4 ? reinterpret_cast<const char*>(p.a_scale) + (size_t)m0_ * p.lds_a
                    : reinterpret_cast<const char*>(p.b_scale) + (size_t)n0_ * p.lds_b;
  };
  auto pieceS = [&](const char* tile, int st_) {   // tile = scale panel + 4 kt
    w8_glds4(tile, s_voff, lds0 + s_base(st_) + s_lds);
  };
  auto pieceA = [&](const char* tile, int sa_, int i) {   // tile = panel base + 128 kt ; piece i of this wave's 4 ; A slot
    const int j = wave * 4 + i;
    w8_glds16(tile + (size_t)j * a_piece, a_voff, lds0 + a_base(sa_) + j * 1024);
  };
  auto pieceB = [&](const char* tile, int sb_, int i) {   // piece i of this wave's NTW ; B slot
    const int j = wave * NTW + i;
    int rows;
    uint32_t vo = b_voff;
    if constexpr (F32OUT) {
      rows = 8 * j;
    } else if constexpr (NTW == 4) {
      rows = (j >> 3) * 64 + ((j >> 2) & 1) * 32 + (j & 1) * 16 + ((j >> 1) & 1) * 4;
    } else {
      const int wcol = j / 6, jj = j - wcol * 6, nt = jj >> 1;
      if (nt < 2) { rows = wcol * 48 + (jj & 1) * 16 + nt * 4; }
      else { rows = wcol * 48 + 32 + (jj & 1) * 8; vo = b_voff2; }
    }
    w8_glds16(tile + (size_t)rows * b_row, vo, lds0 + b_base(sb_) + j * 1024);
  };

  const char *a_cur, *b_cur, *a_nxt, *b_nxt;
  const char *s_cur = nullptr, *s_nxt = nullptr;
  int m0, n0, m0n, n0n;
  bases_of((int)blockIdx.x, a_cur, b_cur, m0, n0);
  if constexpr (MX) s_cur = scales_of(m0, n0);
  // Start stagger (persist bits 4..7 = sixteenths of one tile's duration, spread linearly over the blocks): a tile's
  // 128 KiB of output leaves each CU in one burst, and with all 256 CUs in step a round's 32 MiB take HBM ~7 us to
  // absorb - the waves then wait for those stores in front of the second K tile's staging loads (VMEM operations
  // retire in order).  Blocks that start later are the ones with one work item fewer when the items do not divide
  // evenly (block b walks b, b + grid, ...), so the spread costs little at the end of the launch.
  {
    const int sixteenths = (p.persist >> 4) & 15;
    if (sixteenths) {
      // one tile ~ nk x 2048 cycles; s_sleep 16 = 1024 cycles
      const int n = (int)(((long long)blockIdx.x * nk * 2 * sixteenths) / ((long long)G * 16));
      for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }
  }
  // K tiles 0 and 1 of the first item (K >= 128: the launcher) - and (A3; K >= 256) the A part of tile 2
#pragma unroll
  for (int i = 0; i < NTW; ++i) pieceB(b_cur, 0, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) pieceA(a_cur, 0, i);
  if constexpr (MX) pieceS(s_cur, 0);
#pragma unroll
  for (int i = 0; i < NTW; ++i) pieceB(b_cur + 128, 1, i);
#pragma unroll
  for (int i = 0; i < 4; ++i) pieceA(a_cur + 128, 1, i);
  if constexpr (MX) pieceS(s_cur + 4, 1);
  if constexpr (A3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pieceA(a_cur + 256, 2, i);
    W8_WAITBAR(NPC + 4);
  } else {
    W8_WAITBAR(NPC);   // tile 0 landed in every wave's share
  }

  // ---- fragment read offsets (bytes inside a stage)
  const int g = lane >> 4, l15 = lane & 15;
  const int fx = ((l15 >> 1) & 3) << 1;
  const uint32_t a_rd0 = (uint32_t)((wr * 128 + l15) * 128 + ((g ^ fx) << 4));
  const uint32_t a_rd1 = (uint32_t)((wr * 128 + l15) * 128 + (((4 + g) ^ fx) << 4));
  const uint32_t b_rd0 = (uint32_t)((wc * 16 * NTW + l15) * 128 + ((g ^ fx) << 4));          // (inside the B image)
  const uint32_t b_rd1 = (uint32_t)((wc * 16 * NTW + l15) * 128 + (((4 + g) ^ fx) << 4));

  f32x4 acc[8][NTW];
  h16x8 FA0[8], FA1[8], FB0[NTW], FB1[NTW];
  // MFMAs through inline asm with the accumulator tied in place in the AGPR half of the register file ("+a").
  // Operands swapped: A-operand = weight fragment (rows = n), B-operand = activation fragment (columns = m).
#define W8_MMA(MT, NT, FA, FB) \
  if constexpr ((NT) < NTW) \
    asm volatile(MFMA16_ASM " %0, %1, %2, %0" : "+a"(acc[MT][NT]) : "v"(FB[NT]), "v"(FA[MT]))
#define W8_ROW(MT, FA, FB) W8_MMA(MT, 0, FA, FB); W8_MMA(MT, 1, FA, FB); W8_MMA(MT, 2, FA, FB); W8_MMA(MT, 3, FA, FB)
  // first k-step of a work item: C = 0 instead of a pass that zeroes 32 NTW registers
#define W8_MMA0(MT, NT, FA, FB) \
  if constexpr ((NT) < NTW) \
    asm volatile(MFMA16_ASM " %0, %1, %2, 0" : "=a"(acc[MT][NT]) : "v"(FB[NT]), "v"(FA[MT]))
#define W8_ROW0(MT, FA, FB) W8_MMA0(MT, 0, FA, FB); W8_MMA0(MT, 1, FA, FB); W8_MMA0(MT, 2, FA, FB); W8_MMA0(MT, 3, FA, FB)

  f32x4 bq[4];     // bias of the lane's columns (loaded in an item's last K tile, used by its epilogue)
  int stage = 0;   // B slot (and A slot of the two-stage form)
  int sa = 0;      // A slot
  int extra = 0;   // VMEM operations of the previous epilogue certain to have been issued behind the staged tile
  int kt = 0;      // K tile of the current item
  // One K tile.  FIRST: first K tile of an item (its mid wait must let the previous epilogue's stores pass);
  // LAST: last K tile of an item (the next item's first fragments are read after the epilogue, not under k-step 1:
  // they would be live across it).
  auto ktile = [&](auto first_tag, auto last_tag) {
    constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
    const int san = A3 ? (sa == 2 ? 0 : sa + 1) : (sa ^ 1);
    const char* stA = smem + a_base(sa);
    const char* snA = smem + a_base(san);
    const char* snB = smem + b_base(stage ^ 1);
    const int kn = kt + 2, kna = kt + (A3 ? 3 : 2);
    const char* ta = ((kna < nk) ? a_cur : a_nxt) + (size_t)((kna < nk) ? kna : kna - nk) * 128;
    const char* tb = ((kn < nk) ? b_cur : b_nxt) + (size_t)((kn < nk) ? kn : kn - nk) * 128;
    // ---- k-step 0 (fragments FA0, FB0, FB1 of this tile are in registers), A fragments of k-step 1 read meanwhile
    if constexpr (LAST) {
      // the bias of this item's columns: requested here, retired by the mid wait below (nothing of the epilogue
      // waits for it), held in registers across k-step 1 only
      if constexpr (F32OUT) {
        const uint32_t bo = (uint32_t)(n0 + wc * 16 * NTW + 4 * g) * 4u;
        asm volatile(W8_SGPR_PAD "global_load_dwordx4 %0, %1, %2" : "=&v"(bq[0]) : "v"(bo), "s"(p.bias));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=&v"(bq[1]) : "v"(bo), "s"(p.bias));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "=&v"(bq[2]) : "v"(bo), "s"(p.bias));
        if constexpr (NTW == 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:192" : "=&v"(bq[3]) : "v"(bo), "s"(p.bias));
      } else {
        const uint32_t bo = (uint32_t)(n0 + wc * 16 * NTW + 8 * g) * 4u;
        const uint32_t bo1 = bo + ((NTW == 4) ? 128u : (uint32_t)(128 - 16 * g));
        asm volatile(W8_SGPR_PAD "global_load_dwordx4 %0, %1, %2" : "=&v"(bq[0]) : "v"(bo), "s"(p.bias));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=&v"(bq[1]) : "v"(bo), "s"(p.bias));
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(bq[2]) : "v"(bo1), "s"(p.bias));
        if constexpr (NTW == 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=&v"(bq[3]) : "v"(bo1), "s"(p.bias));
      }
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      if constexpr (FIRST) { W8_ROW0(mt, FA0, FB0); } else { W8_ROW(mt, FA0, FB0); }
      FA1[mt] = *LDS_PTR(const h16x8, stA + a_rd1 + mt * 2048);
    }
    // ---- mid: tile t + 1 landed (this wave's share: its pieces are the only staging loads in flight), every fragment
    //      of tile t is in registers; barrier: stage (t + 1) & 1 is readable, stage t & 1 is free
    if constexpr (FIRST && A3) {
      // behind tile t + 1's B pieces sit the 4 A pieces of tile t + 2 and the previous epilogue's `extra` operations
      if (extra >= 59) W8_WAITBAR(63);
      else if (extra >= 48) W8_WAITBAR(52);
      else if (extra >= 32) W8_WAITBAR(36);
      else if (extra >= 16) W8_WAITBAR(20);
      else W8_WAITBAR(4);
    } else if constexpr (FIRST) {
      // behind tile t + 1's pieces sit the previous epilogue's `extra` operations (VMEM retires in order)
      if (extra >= 63) W8_WAITBAR(63);
      else if (extra >= 48) W8_WAITBAR(48);
      else if (extra >= 32) W8_WAITBAR(32);
      else if (extra >= 16) W8_WAITBAR(16);
      else W8_WAITBAR(0);
    } else if constexpr (A3) {
      if constexpr (LAST) { W8_WAITBAR(8); } else { W8_WAITBAR(4); }   // (LAST: + the 4 bias loads requested above)
    } else {
      W8_WAITBAR(0);
    }
    // ---- k-step 1, with tile t + 2's staging pieces (into the stage just freed) and the first fragments of
    //      tile t + 1 (FA0, FB0: their registers are free) spread between the MFMAs
#define W8_NEXT_A(MT) if constexpr (!LAST) FA0[MT] = *LDS_PTR(const h16x8, snA + a_rd0 + (MT) * 2048)
#define W8_NEXT_B(NT) if constexpr (!LAST && (NT) < NTW) FB0[NT] = *LDS_PTR(const h16x8, snB + b_rd0 + (NT) * 2048)
    if constexpr (A3) {   // B pieces (tile t + 2) first, then the A pieces (tile t + 3): the next mid wait passes over those
      W8_ROW(0, FA1, FB1); pieceB(tb, stage, 0); W8_NEXT_B(0); W8_NEXT_B(1);
      W8_ROW(1, FA1, FB1); pieceB(tb, stage, 1); W8_NEXT_B(2); W8_NEXT_B(3);
      W8_ROW(2, FA1, FB1); pieceB(tb, stage, 2); W8_NEXT_A(0); W8_NEXT_A(1);
      W8_ROW(3, FA1, FB1); if constexpr (NTW == 4) pieceB(tb, stage, 3); W8_NEXT_A(2); W8_NEXT_A(3);
      W8_ROW(4, FA1, FB1); pieceA(ta, sa, 0); W8_NEXT_A(4); W8_NEXT_A(5);
      W8_ROW(5, FA1, FB1); pieceA(ta, sa, 1); W8_NEXT_A(6); W8_NEXT_A(7);
      W8_ROW(6, FA1, FB1); pieceA(ta, sa, 2);
      W8_ROW(7, FA1, FB1); pieceA(ta, sa, 3);
    } else {
      W8_ROW(0, FA1, FB1); pieceA(ta, sa, 0); W8_NEXT_B(0); W8_NEXT_B(1);
      W8_ROW(1, FA1, FB1); pieceA(ta, sa, 1); W8_NEXT_B(2); W8_NEXT_B(3);
      W8_ROW(2, FA1, FB1); pieceA(ta, sa, 2); W8_NEXT_A(0); W8_NEXT_A(1);
      W8_ROW(3, FA1, FB1); pieceA(ta, sa, 3); W8_NEXT_A(2); W8_NEXT_A(3);
      W8_ROW(4, FA1, FB1); pieceB(tb, stage, 0); W8_NEXT_A(4); W8_NEXT_A(5);
      W8_ROW(5, FA1, FB1); pieceB(tb, stage, 1); W8_NEXT_A(6); W8_NEXT_A(7);
      W8_ROW(6, FA1, FB1); pieceB(tb, stage, 2);
      W8_ROW(7, FA1, FB1); if constexpr (NTW == 4) pieceB(tb, stage, 3);
    }
#undef W8_NEXT_A
#undef W8_NEXT_B
    if constexpr (!LAST) {   // (FB1 is free once the last MFMA of the k-step has been issued)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) FB1[nt] = *LDS_PTR(const h16x8, snB + b_rd1 + nt * 2048);
    }
    stage ^= 1;
    sa = san;
    ++kt;
  };

  // ---- MXFP8 form: fragments = the two 16-byte chunks side by side (eight registers per MFMA operand), scale bytes
  w8_i32x8 XA[MX ? 8 : 1], XB[MX ? NTW : 1];
  int sX[MX ? 4 : 1] = {}, sW[MX ? 2 : 1] = {};   // scale bytes, two per register: byte 0 = tile 2 k, byte 2 = tile 2 k + 1 (op_sel_hi)
  // byte g of the row's scale dword: activation row wr * 128 + 16 mt + l15, weight LDS row wc * 16 NTW + 16 nt + l15
  const uint32_t sx_rd = (uint32_t)((wr * 128 + l15) * 4 + g);
  const uint32_t sw_rd = (uint32_t)((256 + wc * 16 * NTW + l15) * 4 + g);
  // (scale byte select: op_sel = bit 0, op_sel_hi = bit 1 of the byte index, first entry the A-operand's scale = weights)
  // The host pass of hipcc parses kernel bodies too and drops - silently: the stub stays an undefined symbol - a kernel whose
  // asm operands it cannot type for x86 (a 256-bit "v" operand without AVX): the host sees no statement at all.
#define W8_XASM(CSRC, HI) "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, " CSRC ", %3, %4 op_sel_hi:" HI
#if !defined(__HIP_DEVICE_COMPILE__)
#define W8_XMMA(MT, NT) (void)0
#define W8_XMMA0(MT, NT) (void)0
#else
#define W8_XMMA(MT, NT) \
  if constexpr ((NT) < NTW) { \
    if constexpr (((NT) & 1) == 0 && ((MT) & 1) == 0) \
      asm volatile(W8_XASM("%0", "[0,0,0]") : "+a"(acc[MT][NT]) : "v"(XB[NT]), "v"(XA[MT]), "v"(sW[(NT) >> 1]), "v"(sX[(MT) >> 1])); \
    else if constexpr (((NT) & 1) == 1 && ((MT) & 1) == 0) \
      asm volatile(W8_XASM("%0", "[1,0,0]") : "+a"(acc[MT][NT]) : "v"(XB[NT]), "v"(XA[MT]), "v"(sW[(NT) >> 1]), "v"(sX[(MT) >> 1])); \
    else if constexpr (((NT) & 1) == 0) \
      asm volatile(W8_XASM("%0", "[0,1,0]") : "+a"(acc[MT][NT]) : "v"(XB[NT]), "v"(XA[MT]), "v"(sW[(NT) >> 1]), "v"(sX[(MT) >> 1])); \
    else \
      asm volatile(W8_XASM("%0", "[1,1,0]") : "+a"(acc[MT][NT]) : "v"(XB[NT]), "v"(XA[MT]), "v"(sW[(NT) >> 1]), "v"(sX[(MT) >> 1])); \
  }
#define W8_XMMA0(MT, NT) \
  if constexpr ((NT) < NTW) { \
    if constexpr (((NT) & 1) == 0 && ((MT) & 1) == 0) \
      asm volatile(W8_XASM("0", "[0,0,0]") : "=a"(acc[MT][NT]) : "v"(XB[NT]), "v"(XA[MT]), "v"(sW[(NT) >> 1]), "v"(sX[(MT) >> 1])); \
    else if constexpr (((NT) & 1) == 1 && ((MT) & 1) == 0) \
      asm volatile(W8_XASM("0", "[1,0,0]") : "=a"(acc[MT][NT]) : "v"(XB[NT]), "v"(XA[MT]), "v"(sW[(NT) >> 1]), "v"(sX[(MT) >> 1])); \
    else if constexpr (((NT) & 1) == 0) \
      asm volatile(W8_XASM("0", "[0,1,0]") : "=a"(acc[MT][NT]) : "v"(XB[NT]), "v"(XA[MT]), "v"(sW[(NT) >> 1]), "v"(sX[(MT) >> 1])); \
    else \
      asm volatile(W8_XASM("0", "[1,1,0]") : "=a"(acc[MT][NT]) : "v"(XB[NT]), "v"(XA[MT]), "v"(sW[(NT) >> 1]), "v"(sX[(MT) >> 1])); \
  }
#endif
  // four row tiles against ONE weight fragment (nt-major order: a weight fragment is free - and the next tile's can be read -
  // as soon as its group has been issued; the first group of a tile waits for one weight fragment, not for all of them)
#define W8_XCOL(NT, M0) \
  if constexpr (FIRST) { W8_XMMA0((M0), NT); W8_XMMA0((M0) + 1, NT); W8_XMMA0((M0) + 2, NT); W8_XMMA0((M0) + 3, NT); } \
  else { W8_XMMA((M0), NT); W8_XMMA((M0) + 1, NT); W8_XMMA((M0) + 2, NT); W8_XMMA((M0) + 3, NT); }
  // even tile: its byte 0, keeping the odd tile's byte 2 ; odd tile: its byte 2, keeping byte 0
#define W8_XSET(REG, ODD, BYTE) \
  REG = (ODD) ? (((REG) & 0xffff) | ((BYTE) << 16)) : ((int)((uint32_t)(REG) & 0xffff0000u) | (BYTE))
#define W8_XREAD_A(MT, SA, SS) \
  XA[MT].lo = *LDS_PTR(const w8_i32x4, (SA) + a_rd0 + (MT) * 2048); \
  XA[MT].hi = *LDS_PTR(const w8_i32x4, (SA) + a_rd1 + (MT) * 2048); \
  W8_XSET(sX[(MT) >> 1], (MT) & 1, (int)*LDS_PTR(const uint8_t, (SS) + sx_rd + (MT) * 64))
#define W8_XREAD_B(NT, SB, SS) \
  XB[NT].lo = *LDS_PTR(const w8_i32x4, (SB) + b_rd0 + (NT) * 2048); \
  XB[NT].hi = *LDS_PTR(const w8_i32x4, (SB) + b_rd1 + (NT) * 2048); \
  W8_XSET(sW[(NT) >> 1], (NT) & 1, (int)*LDS_PTR(const uint8_t, (SS) + sw_rd + (NT) * 64))
  auto ktile_mx = [&](auto first_tag, auto last_tag) {
    constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
    const char* stA = smem + a_base(stage);
    const char* stS = smem + s_base(stage);
    const char* snA = smem + a_base(stage ^ 1);
    const char* snB = smem + b_base(stage ^ 1);
    const char* snS = smem + s_base(stage ^ 1);
    const int kn = kt + 2;
    const bool here = kn < nk;
    const int ko = here ? kn : kn - nk;
    const char* ta = (here ? a_cur : a_nxt) + (size_t)ko * 128;
    const char* tb = (here ? b_cur : b_nxt) + (size_t)ko * 128;
    const char* ts = (here ? s_cur : s_nxt) + (size_t)ko * 4;
    // ---- rows 0-3 (XA[0..3], XB of this tile are in registers), the fragments of rows 4-7 read meanwhile
    W8_XCOL(0, 0); W8_XREAD_A(4, stA, stS);
    W8_XCOL(1, 0); W8_XREAD_A(5, stA, stS);
    W8_XCOL(2, 0); W8_XREAD_A(6, stA, stS);
    if constexpr (NTW == 4) { W8_XCOL(3, 0); }
    W8_XREAD_A(7, stA, stS);
    // ---- mid: tile t + 1 landed, every fragment of tile t is in registers
    if constexpr (FIRST) {
      if (extra >= 63) W8_WAITBAR(63);
      else if (extra >= 48) W8_WAITBAR(48);
      else if (extra >= 32) W8_WAITBAR(32);
      else if (extra >= 16) W8_WAITBAR(16);
      else W8_WAITBAR(0);
    } else {
      W8_WAITBAR(0);
    }
    // ---- rows 4-7, with tile t + 2's staging pieces (into the stage just freed) and tile t + 1's fragments between the groups
    if constexpr (LAST) {   // the bias of this item's columns: requested here (the registers of XA[0..3] are free: no next tile
                            // is read under a last tile), retired behind the loop by a wait that leaves this tile's pieces in flight
      if constexpr (F32OUT) {
        const uint32_t bo = (uint32_t)(n0 + wc * 16 * NTW + 4 * g) * 4u;
        asm volatile(W8_SGPR_PAD "global_load_dwordx4 %0, %1, %2" : "=&v"(bq[0]) : "v"(bo), "s"(p.bias));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=&v"(bq[1]) : "v"(bo), "s"(p.bias));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" : "=&v"(bq[2]) : "v"(bo), "s"(p.bias));
        if constexpr (NTW == 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:192" : "=&v"(bq[3]) : "v"(bo), "s"(p.bias));
      } else {
        const uint32_t bo = (uint32_t)(n0 + wc * 16 * NTW + 8 * g) * 4u;
        const uint32_t bo1 = bo + ((NTW == 4) ? 128u : (uint32_t)(128 - 16 * g));
        asm volatile(W8_SGPR_PAD "global_load_dwordx4 %0, %1, %2" : "=&v"(bq[0]) : "v"(bo), "s"(p.bias));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=&v"(bq[1]) : "v"(bo), "s"(p.bias));
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(bq[2]) : "v"(bo1), "s"(p.bias));
        if constexpr (NTW == 4) asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=&v"(bq[3]) : "v"(bo1), "s"(p.bias));
      }
    }
    W8_XCOL(0, 4); pieceA(ta, stage, 0); pieceA(ta, stage, 1);
    if constexpr (!LAST) { W8_XREAD_B(0, snB, snS); W8_XREAD_A(0, snA, snS); }
    W8_XCOL(1, 4); pieceA(ta, stage, 2); pieceA(ta, stage, 3);
    if constexpr (!LAST) { W8_XREAD_B(1, snB, snS); W8_XREAD_A(1, snA, snS); }
    W8_XCOL(2, 4); pieceB(tb, stage, 0); pieceB(tb, stage, 1);
    if constexpr (NTW == 3) { pieceB(tb, stage, 2); pieceS(ts, stage); }
    if constexpr (!LAST) {
      W8_XREAD_B(2, snB, snS); W8_XREAD_A(2, snA, snS);
      if constexpr (NTW == 3) { W8_XREAD_A(3, snA, snS); }
    }
    if constexpr (NTW == 4) {
      W8_XCOL(3, 4); pieceB(tb, stage, 2); pieceB(tb, stage, 3); pieceS(ts, stage);
      if constexpr (!LAST) { W8_XREAD_B(3, snB, snS); W8_XREAD_A(3, snA, snS); }
    }
    stage ^= 1;
    ++kt;
  };

  for (int w = (int)blockIdx.x; w < nwork; w += G) {
    // the load cursor (two K tiles ahead) crosses into the next item inside this item's loop
    if (w + G < nwork) bases_of(w + G, a_nxt, b_nxt, m0n, n0n);
    else { a_nxt = a_cur; b_nxt = b_cur; m0n = m0; n0n = n0; }   // past the end: valid tiles into stages nobody reads
    if constexpr (MX) {
      s_nxt = scales_of(m0n, n0n);
      const char* stA = smem + a_base(stage);
      const char* stB = smem + b_base(stage);
      const char* stS = smem + s_base(stage);
      W8_XREAD_B(0, stB, stS); W8_XREAD_B(1, stB, stS); W8_XREAD_B(2, stB, stS);
      if constexpr (NTW == 4) { W8_XREAD_B(3, stB, stS); }
      W8_XREAD_A(0, stA, stS); W8_XREAD_A(1, stA, stS); W8_XREAD_A(2, stA, stS); W8_XREAD_A(3, stA, stS);
      kt = 0;
      ktile_mx(std::true_type{}, std::false_type{});
#pragma clang loop unroll(disable)
      while (kt < nk - 1) ktile_mx(std::false_type{}, std::false_type{});
      ktile_mx(std::false_type{}, std::true_type{});   // (K >= 384: the launcher)
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_nop 15\n\ts_nop 15" ::"i"(NPC) : "memory");   // the bias has landed
    } else {
    {   // first fragments of the item's first K tile (landed: the wait that ended the previous item / the prologue)
      const char* stA = smem + a_base(sa);
      const char* stB = smem + b_base(stage);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        FB0[nt] = *LDS_PTR(const h16x8, stB + b_rd0 + nt * 2048);
        FB1[nt] = *LDS_PTR(const h16x8, stB + b_rd1 + nt * 2048);
      }
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) FA0[mt] = *LDS_PTR(const h16x8, stA + a_rd0 + mt * 2048);
    }
    kt = 0;
    ktile(std::true_type{}, std::false_type{});
#pragma clang loop unroll(disable)
    while (kt < nk - 1) ktile(std::false_type{}, std::false_type{});
    ktile(std::false_type{}, std::true_type{});   // (K >= 192: the launcher; two-tile items take the path below)
    if constexpr (A3) {
      // the bias loads (requested at the start of the last tile) have landed; behind them: that tile's staging pieces
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NTW + 4) : "memory");
    }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> accumulator reads (invisible inside asm)

    // =============================== epilogue: registers -> global ===============================
    // (every asm store ends with `s_nop 1` inside its string: hipcc does not pad the hazard of a 16-byte store whose
    //  data registers the next instruction overwrites - seen as rare garbage in rows 12-15 of a tile's first stores)
    // Every global load / store / atomic of the epilogue goes through asm, like the staging loads: hipcc then has
    // no VMEM operation of its own in this kernel and inserts no `vmcnt` wait anywhere (one of its waits - it adds
    // them in front of register re-use across the persistent loop - would also wait for the staging loads in
    // flight).  Loads are retired by counted waits: in a fully valid tile (uniform per wave) every operation below
    // is issued, so the number of operations behind a load is known; otherwise the waits are vmcnt(0).
    // Accumulators are read where they are used (volatile asm: program order) - left to itself hipcc copies most of
    // them to VGPRs right behind the loop and spills.
#define W8_ACC(MT, NT, E) ({ float x_; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x_) : "a"(acc[MT][NT][E])); x_; })
    const int mw = m0 + wr * 128;                // first row of this wave
    const bool full = (mw + 128 <= p.m_valid);
    // lane coordinates recomputed from the hardware lane id: not values kept live (or spilled) across the main loop
    const int eln = lane_id_volatile();
    const int eg = eln >> 4, el15 = eln & 15;
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(bq[k]));   // (consumers stay behind the wait that retired the bias)
    if constexpr (!F32OUT) {
      // lane: row m = mw + 16 mt + l15.  Unit 0 of a row = tiles 0, 1: the 8 columns nc + {0..7}; unit 1 = tiles 2, 3:
      // nc + 32 + {0..7} (NTW == 4) or tile 2 alone: the 4 columns nc + 32 - 4 g + {0..3} (NTW == 3).
      constexpr bool HAS_AUX = (EPI == EPI_BF16_DGELU || EPI == EPI_BF16_DGELU_U8);
      constexpr bool GELU2 = (EPI == EPI_BF16_GELU || EPI == EPI_BF16_GELU_U8);   // gelu and gelu' come out
      // 8-bit gelu' in TILE-NATIVE order (EPI_*_U8): one 16-byte slot per (tile, wave, mt, lane) holding the lane's 16 (12 at
      // NTW == 3) values of row-tile mt as q = rne(200 g + 26) (0, 0.5 and 1 are grid points; step 0.005, range -0.13 .. 1.145),
      // written here by the FFN-in forward and read back by the gelu'-product dgrad of the same shape and tile width: half the
      // bytes of the bf16 tensor and one 1-KiB-contiguous store / load per wave and mt instead of two
      constexpr bool U8 = (EPI == EPI_BF16_GELU_U8 || EPI == EPI_BF16_DGELU_U8);
      constexpr bool HAS_CSUM = (EPI == EPI_BF16 || HAS_AUX);   // (the GELU forms have no registers for it)
      constexpr int U = 16, PD = 4;                                 // units per wave tile ; aux prefetch distance (units)
      constexpr int PDM = 2;                                        // 8-bit aux: prefetch distance in row tiles
      constexpr int SU = (EPI == EPI_BF16_GELU) ? 2 : 1;            // stores per unit (the 8-bit form: + one per mt)
      constexpr int NV1 = (NTW == 4) ? 8 : 4;                       // values of unit 1
      // Whole-line stores (256-wide tiles): the two units of a row tile are the two 64-byte
      // halves of the wave's 128-byte row segments.  Stored unit by unit, an instruction writes 16 half lines; with the
      // halves of rows 0-7 / 8-15 exchanged between lanes l15 and l15 ^ 8 (DPP row_ror:8) an instruction writes 8 whole
      // lines.  Built and measured on one box (tools/gemm_bench.py, M = 47360): gelu'-product dgrad 261.5 -> 250-257 us; the
      // GELU forward got slower (285 -> 288-295: unit 0 stays in registers under unit 1's GELU arithmetic) and the plain
      // bf16 epilogue did not move (QKV 172 -> 172-176) - so only the 8-bit gelu'-product form stores this way
      constexpr bool WL = (NTW == 4) && EPI == EPI_BF16_DGELU_U8;
      const int nc = n0 + wc * 16 * NTW + 8 * eg;
      const uint32_t c1 = (NTW == 4) ? 64u : (uint32_t)(64 - 8 * eg);   // byte offset of unit 1 behind unit 0 (bf16)
      // (bias: bq, loaded under the item's last K tile - always a valid pointer: the launcher substitutes zeros)
      float csum[HAS_CSUM ? 16 : 1];
#pragma unroll
      for (int e = 0; e < (HAS_CSUM ? 16 : 1); ++e) csum[e] = 0.f;
      // addresses: uniform 64-bit base (SGPR pair) + ONE 32-bit byte offset per lane, stepped by 16 rows per mt
      // (the launcher keeps M x ldo x 4 B below 4 GiB)
#if W8_ABLATE == 2 || W8_ABLATE == 4   // development: every tile writes the same 256 rows (the stores stay in L2)
      const uint32_t off0 = (uint32_t)(((size_t)(wr * 128 + el15) * p.ldo + nc) * 2);
#else
      const uint32_t off0 = (uint32_t)(((size_t)(mw + el15) * p.ldo + nc) * 2);
#endif
      const uint32_t step = (uint32_t)p.ldo * 32u;
      // store offsets: the output rows, or - head-major output (p.out_hm = rows per plane: the QKV forward writing what the
      // attention kernels read, [N / 64][out_hm][64]) - column c in plane c >> 6 at position c & 63: the units of a lane are
      // 16 (8) bytes inside ONE plane row, so only the three constants change
      uint32_t so0 = off0, sstep = step, sc1 = c1;
      if (EPI == EPI_BF16 && p.out_hm) {
        const int nc1 = nc + (int)(c1 >> 1);
        const uint32_t cp0 = (uint32_t)(nc >> 6) * (uint32_t)p.out_hm * 128u + (uint32_t)(nc & 63) * 2u;
        const uint32_t cp1 = (uint32_t)(nc1 >> 6) * (uint32_t)p.out_hm * 128u + (uint32_t)(nc1 & 63) * 2u;
        so0 = (uint32_t)(mw + el15) * 128u + cp0;
        sstep = 2048u;
        sc1 = cp1 - cp0;
      }
      // WL: rows (l15 & 7) [+ 8 for the second instruction], byte half (l15 >> 3) of the 128-byte segment
      const uint32_t offw = (uint32_t)(((size_t)(mw + (el15 & 7)) * p.ldo + nc) * 2) + (uint32_t)(el15 >> 3) * 64u;
      const char* outp = reinterpret_cast<const char*>(p.out);
      const char* out2p = reinterpret_cast<const char*>(p.out2);
      const char* auxp = reinterpret_cast<const char*>(p.aux);
      // 8-bit gelu' slots of this wave's tile: uniform base + lane * 16 (+ 1024 mt)
      const char* u8p = (GELU2 ? out2p : auxp) + ((size_t)((m0 >> 8) * tiles_n + n0 / BN) * 8 + wave) * 8192;
      const uint32_t voff8 = (uint32_t)eln * 16u;
      // MXFP8 image of the output (MX form, 256-wide tiles): the lanes g = 0..3 of a row hold the 32 consecutive columns of
      // one block (8 each); element offset of unit 0 of row tile 0, and of its scale byte
      constexpr bool QOUT = MX && NTW == 4 && (GELU2 || EPI == EPI_BF16_GELU_INF);   // (the plain form has no registers for it)
      const char* oqp = QOUT ? reinterpret_cast<const char*>(p.out_q) : nullptr;
      const char* osp = reinterpret_cast<const char*>(p.out_scale);
      const uint32_t offq = (uint32_t)(mw + el15) * (uint32_t)p.N + (uint32_t)nc;
      const uint32_t offs = (uint32_t)(mw + el15) * (uint32_t)(p.N >> 5) + (uint32_t)((n0 + wc * 16 * NTW) >> 5);
      auto run = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const char* const o1_ = outp;     // (named here: an asm operand alone does not capture in a generic lambda)
        const char* const o2_ = out2p;
        const char* const ax_ = auxp;
        const char* const u8_ = u8p;
        const uint32_t v8_ = voff8;
        const char* const oq_ = oqp;
        const char* const os_ = osp;
        uint32_t q8a = 0u, q8b = 0u;      // 8-bit gelu' of unit 0, held until unit 1 completes the slot
        u32x4 wv0 = {0u, 0u, 0u, 0u};     // WL: unit 0 of the row tile, held until unit 1 is packed
        u32x4 a8q[(HAS_AUX && U8) ? PDM : 1];
        u32x4 a8 = {0u, 0u, 0u, 0u};
        auto aux8_load = [&](int mt) {
          const uint32_t o = v8_ + (uint32_t)mt * 1024u;
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(a8q[mt % PDM]) : "v"(o), "s"(u8_));
        };
        // prefetch queue of the gelu' operand: slot u % PD (PD even: odd slots always hold odd units - with 192-wide
        // tiles those are the 8-byte tails, kept in their own registers: an asm load's destination must be consumed
        // as it is, behind the wait - a copy into a wider register would read it before the data arrives)
        h16x8 axq[HAS_AUX ? PD : 1];
        u32x2 axh[(HAS_AUX && NTW == 3) ? PD : 1];
        // (loads are never predicated: a guarded asm load merges its destination with the old value - a register copy
        //  hipcc would place before our wait; rows that are not stored read a valid row instead)
        auto aux_load = [&](int u) {
          const int m = mw + (u >> 1) * 16 + el15;
          const uint32_t orow = FULL ? off0 + (uint32_t)(u >> 1) * step
                                     : (uint32_t)(((size_t)min(m, p.m_valid - 1) * p.ldo + nc) * 2);
          const uint32_t o = orow + ((u & 1) ? c1 : 0u);
          if ((u & 1) && NTW == 3) asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(axh[u % PD]) : "v"(o), "s"(ax_));
          else asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(axq[u % PD]) : "v"(o), "s"(ax_));
        };
        if constexpr (HAS_AUX && U8) {
#pragma unroll
          for (int mt = 0; mt < PDM; ++mt) aux8_load(mt);
        } else if constexpr (HAS_AUX) {
#pragma unroll
          for (int u = 0; u < PD; ++u) aux_load(u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int mt = u >> 1, j = u & 1;
          const int m = mw + mt * 16 + el15;
          const int nv = (j == 0) ? 8 : NV1;
          h16x8 ax;
          if constexpr (HAS_AUX && U8) {
            if (j == 0) {
              // behind load mt: the loads mt + 1 .. mt + PDM - 1 and the stores of the units since it was issued
              const int loads_behind = (mt + PDM - 1 < 8 ? PDM - 1 : 7 - mt);
              const int stores_behind = 2 * (mt < PDM ? mt : PDM);
              if (FULL) { W8_VMCNT_CASES(loads_behind + stores_behind) } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
              a8 = a8q[mt % PDM];
              asm volatile("" : "+v"(a8));
              if (mt + PDM < 8) aux8_load(mt + PDM);
            }
          } else if constexpr (HAS_AUX) {
            // operations behind load u: the loads u+1 .. u+PD-1 issued so far (load u+PD is issued below, after the
            // wait) and the stores of the units since load u was issued
            const int loads_behind = (u + PD - 1 < U ? PD - 1 : U - 1 - u);
            const int stores_behind = SU * (u < PD ? u : PD);
            if (FULL) { W8_VMCNT_CASES(loads_behind + stores_behind) } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if ((u & 1) && NTW == 3) {
              u32x2 h = axh[u % PD];
              asm volatile("" : "+v"(h));
              ax = __builtin_bit_cast(h16x8, u32x4{h[0], h[1], 0u, 0u});
            } else {
              ax = axq[u % PD];
              asm volatile("" : "+v"(ax));
            }
            if (u + PD < U) aux_load(u + PD);
          }
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = W8_ACC(mt, 2 * j, e) + bq[2 * j][e];
            if (j == 0 || NTW == 4) v[4 + e] = W8_ACC(mt, (2 * j + 1 < NTW ? 2 * j + 1 : 0), e) + bq[2 * j + 1][e];
            else v[4 + e] = 0.f;
          }
          u32x4 w2 = {0u, 0u, 0u, 0u};
          float gp4[4] = {0.f, 0.f, 0.f, 0.f};
          if constexpr (GELU2) {
            float gp[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              if (e < nv) {
                f32x2 y2, d2;
                gelu_fwd_f2(f32x2{v[e], v[e + 1]}, y2, d2);
                v[e] = y2[0]; v[e + 1] = y2[1];
                gp[e] = d2[0]; gp[e + 1] = d2[1];
              } else { gp[e] = 0.f; gp[e + 1] = 0.f; }
            }
            if constexpr (U8) {
              uint32_t d0 = 0u, d1 = 0u;     // v_cvt_pk_u8_f32: round to nearest even, saturating (tools/micro/cvt_probe.hip)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                d0 = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(gp[e], 200.f, 26.f), (uint32_t)e, d0);
                d1 = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(gp[4 + e], 200.f, 26.f), (uint32_t)e, d1);
              }
              if (j == 0) { q8a = d0; q8b = d1; }
              else w2 = u32x4{q8a, q8b, d0, d1};
            } else {
              w2 = u32x4{pack_h16x2(gp[0], gp[1]), pack_h16x2(gp[2], gp[3]), pack_h16x2(gp[4], gp[5]),
                         pack_h16x2(gp[6], gp[7])};
              gp4[0] = gp[0]; gp4[1] = gp[1]; gp4[2] = gp[2]; gp4[3] = gp[3];
            }
          } else if constexpr (EPI == EPI_BF16_GELU_INF) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              if (e < nv) {
                const f32x2 x2 = {v[e], v[e + 1]};
                const f32x2 y2 = x2 * norm_cdf_f2(x2);
                v[e] = y2[0]; v[e + 1] = y2[1];
              }
            }
          } else if constexpr (EPI == EPI_BF16_DGELU_U8) {
            const uint32_t b0 = j ? a8[2] : a8[0], b1 = j ? a8[3] : a8[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] *= __builtin_fmaf((float)((b0 >> (8 * e)) & 255u), 0.005f, -0.13f);
              if (4 + e < nv) v[4 + e] *= __builtin_fmaf((float)((b1 >> (8 * e)) & 255u), 0.005f, -0.13f);
            }
          } else if constexpr (EPI == EPI_BF16_DGELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (e < nv) v[e] *= (float)ax[e];
          }
          if constexpr (HAS_CSUM) {   // (rows that are not stored do not count)
            if (FULL || m < p.m_valid) {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (e < nv) csum[8 * j + e] += v[e];
            }
          }
          const u32x4 wv = {pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3]), pack_h16x2(v[4], v[5]),
                            pack_h16x2(v[6], v[7])};
          if constexpr (QOUT) {
            if (oq_ != nullptr) {   // uniform
              float xq[8];
              float amax = 0.f;
#pragma unroll
              for (int e = 0; e < 8; ++e) { xq[e] = (float)(h16)v[e]; amax = fmaxf(amax, fabsf(xq[e])); }
              // max over the lanes g = 0..3 of the row (lane = 16 g + l15): the row pairs, then the wave halves
              {
                const uint32_t ab = __builtin_bit_cast(uint32_t, amax);
                const auto s16 = __builtin_amdgcn_permlane16_swap(ab, ab, false, false);
                amax = fmaxf(__builtin_bit_cast(float, (uint32_t)s16[0]), __builtin_bit_cast(float, (uint32_t)s16[1]));
                const uint32_t ab2 = __builtin_bit_cast(uint32_t, amax);
                const auto s32 = __builtin_amdgcn_permlane32_swap(ab2, ab2, false, false);
                amax = fmaxf(__builtin_bit_cast(float, (uint32_t)s32[0]), __builtin_bit_cast(float, (uint32_t)s32[1]));
              }
              int e8;
              const uint2 q2 = mx8_quant8(xq, amax, e8);
              if (FULL || m < p.m_valid) {
                const u32x2 qv = {q2.x, q2.y};
                const uint32_t oq = offq + (uint32_t)mt * (16u * (uint32_t)p.N) + (j ? 32u : 0u);
                asm volatile(W8_SGPR_PAD "global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(oq), "v"(qv), "s"(oq_) : "memory");
                if (eg == 0) {
                  const uint32_t osv = offs + (uint32_t)mt * (16u * (uint32_t)(p.N >> 5)) + (uint32_t)j;
                  asm volatile(W8_SGPR_PAD "global_store_byte %0, %1, %2\n\ts_nop 1" ::"v"(osv), "v"(e8), "s"(os_) : "memory");
                }
              }
            }
          }
          if constexpr (WL) {
            if (j == 0) {
              wv0 = wv;
            } else {
              // lanes l15 < 8 keep unit 0 of their row for instruction A and take unit 0 of row l15 + 8 for instruction B;
              // lanes l15 >= 8 take unit 1 of row l15 - 8 for A and keep their own unit 1 for B
              const bool lo8 = el15 < 8;
              u32x4 da, db;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const uint32_t xr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wv0[k], 0x128, 0xF, 0xF, true);   // row_ror:8
                const uint32_t yr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wv[k], 0x128, 0xF, 0xF, true);
                da[k] = lo8 ? wv0[k] : yr;
                db[k] = lo8 ? xr : wv[k];
              }
              const int ma = mw + mt * 16 + (el15 & 7);
              const uint32_t oa = offw + (uint32_t)mt * step;
              if (FULL || ma < p.m_valid)
                asm volatile("global_store_dwordx4 %0, %1, %2" W8_STORE_MOD "\n\ts_nop 1" ::"v"(oa), "v"(da), "s"(o1_) : "memory");
              if (FULL || ma + 8 < p.m_valid) {
                const uint32_t ob = oa + (uint32_t)p.ldo * 16u;
                asm volatile("global_store_dwordx4 %0, %1, %2" W8_STORE_MOD "\n\ts_nop 1" ::"v"(ob), "v"(db), "s"(o1_) : "memory");
              }
            }
          } else
#if W8_ABLATE == 1   // development: half of the stores
          if ((FULL || m < p.m_valid) && j == 0) {
#else
          if (FULL || m < p.m_valid) {
#endif
            const uint32_t o = so0 + (uint32_t)mt * sstep + (j ? sc1 : 0u);
            if (nv == 8) {
              if constexpr (EPI == EPI_BF16_GELU)
                asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(o), "v"(w2), "s"(o2_) : "memory");
              asm volatile("global_store_dwordx4 %0, %1, %2" W8_STORE_MOD "\n\ts_nop 1" ::"v"(o), "v"(wv), "s"(o1_) : "memory");
            } else {
              if constexpr (EPI == EPI_BF16_GELU) {
                const u32x2 h2 = {pack_h16x2(gp4[0], gp4[1]), pack_h16x2(gp4[2], gp4[3])};
                asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(o), "v"(h2), "s"(o2_) : "memory");
              }
              const u32x2 h1 = {pack_h16x2(v[0], v[1]), pack_h16x2(v[2], v[3])};
              asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" ::"v"(o), "v"(h1), "s"(o1_) : "memory");
            }
          }
          if constexpr (EPI == EPI_BF16_GELU_U8) {
            if (j == 1) {   // the row tile's slot is complete (every lane stores: the slot array covers whole tiles)
              const uint32_t o8 = v8_ + (uint32_t)mt * 1024u;
              asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(o8), "v"(w2), "s"(u8_) : "memory");
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // one unit at a time: interleaved units do not fit 128 registers
        }
      };
      if (full) run(std::true_type{}); else run(std::false_type{});
      // stores (+ gelu' loads) of a fully valid wave tile
      extra = full ? (U8 ? 24 : (16 * SU + (HAS_AUX ? 16 : 0))) + ((QOUT && oqp != nullptr) ? 32 : 0) : 0;
      if (HAS_CSUM && p.colsum != nullptr) {   // uniform: bias gradient = column sums of the stored values
        // fold the 16 lanes (rows) of each column group (DPP: no LDS traffic, no index registers)
#pragma unroll
        for (int e = 0; e < 8 + NV1; ++e) csum[e] = w8_row16_sum(csum[e]);
        // lane e of a row takes column e's sum: ONE atomic instruction per wave (two 128-byte lines of 32 sums each)
        // instead of 16 with four active lanes - an eighth of the requests that queue up on one address when the
        // blocks of a round finish together (small M: no stagger between the rounds)
        float sel = csum[0];
#pragma unroll
        for (int e = 1; e < 8 + NV1; ++e) sel = (el15 == e) ? csum[e] : sel;
        if (el15 < 8 + NV1) {
          const uint32_t co = (uint32_t)nc * 4u + (el15 < 8 ? (uint32_t)el15 * 4u : 2u * c1 + (uint32_t)(el15 - 8) * 4u);
          asm volatile(W8_SGPR_PAD "global_atomic_add_f32 %0, %1, %2\n\ts_nop 1" ::"v"(co), "v"(sel), "s"(p.colsum) : "memory");
        }
      }
    } else {
      // f32 out = acc + bias + res ; lane: row m, columns nc + 16 nt + {0..3}.  Unit u = NTW mt + nt.
      constexpr int U = 8 * NTW, PD = 8;
      const int nc = n0 + wc * 16 * NTW + 4 * eg;
      const uint32_t off0 = (uint32_t)(((size_t)(mw + el15) * p.ldo + nc) * 4);
      const uint32_t step = (uint32_t)p.ldo * 64u;
      const char* outp = reinterpret_cast<const char*>(p.out);
      const char* resp = reinterpret_cast<const char*>(p.res);
      auto run = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const char* const o1_ = outp;
        const char* const rs_ = resp;
        f32x4 rq[PD];
#define W8_OFF4(OP, NT, ...) \
        switch (NT) { \
          case 0: asm volatile(OP " offset:0\n\ts_nop 1" __VA_ARGS__); break; \
          case 1: asm volatile(OP " offset:64\n\ts_nop 1" __VA_ARGS__); break; \
          case 2: asm volatile(OP " offset:128\n\ts_nop 1" __VA_ARGS__); break; \
          default: asm volatile(OP " offset:192\n\ts_nop 1" __VA_ARGS__); break; \
        }
        auto res_load = [&](int u, f32x4& d) {   // (never predicated: see the bf16 form)
          const int m = mw + (u / NTW) * 16 + el15;
          const uint32_t o = FULL ? off0 + (uint32_t)(u / NTW) * step
                                  : (uint32_t)(((size_t)min(m, p.m_valid - 1) * p.ldo + nc) * 4);
          W8_OFF4("global_load_dwordx4 %0, %1, %2", u % NTW, : "=&v"(d) : "v"(o), "s"(rs_))
        };
#pragma unroll
        for (int u = 0; u < PD; ++u) res_load(u, rq[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int mt = u / NTW, nt = u % NTW;
          const int m = mw + mt * 16 + el15;
          f32x4 rv = rq[u % PD];
          const int loads_behind = (u + PD - 1 < U ? PD - 1 : U - 1 - u);
          const int stores_behind = (u < PD ? u : PD);
          if (FULL) { W8_VMCNT_CASES(loads_behind + stores_behind) } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
          asm volatile("" : "+v"(rv));
          if (u + PD < U) res_load(u + PD, rq[u % PD]);
          const f32x4 a4 = {W8_ACC(mt, nt, 0), W8_ACC(mt, nt, 1), W8_ACC(mt, nt, 2), W8_ACC(mt, nt, 3)};
          const f32x4 ov = a4 + bq[nt] + rv;
          if (FULL || m < p.m_valid) {
            const uint32_t o = off0 + (uint32_t)mt * step;
            W8_OFF4("global_store_dwordx4 %0, %1, %2", nt, ::"v"(o), "v"(ov), "s"(o1_) : "memory")
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#undef W8_OFF4
      };
      if (full) run(std::true_type{}); else run(std::false_type{});
      extra = full ? (16 * NTW > 63 ? 63 : 16 * NTW) : 0;   // 8 NTW loads + 8 NTW stores
    }
#undef W8_ACC
    a_cur = a_nxt; b_cur = b_nxt; m0 = m0n; n0 = n0n;
    s_cur = s_nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the cursor's last (unused) tiles must not land after exit
#if W8_STAMP
  if (threadIdx.x == 0) {
    g_w8_stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
    g_w8_stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
#undef W8_ROW
#undef W8_MMA
#undef W8_XCOL
#undef W8_XSET
#undef W8_XASM
#undef W8_XMMA
#undef W8_XMMA0
#undef W8_XREAD_A
#undef W8_XREAD_B
}

template <int EPI, int NTW, bool A3 = false, bool MX = false>
int launch8w(const GemmParams& p, hipStream_t st) {
  if constexpr (!A3 && !MX) {
    // three A slots (kernel header) where the contraction has at least four K tiles
    if (p.K >= 256) return launch8w<EPI, NTW, true>(p, st);
  }
  constexpr int LDS = (A3 ? 3 : 2) * W8_A_BYTES + 2 * 64 * NTW * 128 + (MX ? 2 * 2048 : 0);
  auto kern = gemm8w_kernel<EPI, NTW, A3, MX>;
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VAULT_EINVAL;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_done[dev] = true;
  }
  GemmParams q = p;
  if (q.bias == nullptr) {   // the kernel always loads a bias: a zero vector of the library stands in
    void* z = nullptr;
    if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_w8_zero_bias)) != hipSuccess || z == nullptr) return VAULT_EINVAL;
    q.bias = reinterpret_cast<const float*>(z);
  }
  const int tiles_n = p.N / (64 * NTW);
  // Raster: column groups of gn n-tiles, all row panels of a group before the next group (gemm_raster).  Default: a group is
  // about ONE round of the chip - gn = ceil(256 / row panels), at least 2: the CUs of an XCD then keep the same gn weight panels
  // in their L2 while the row panels stream through (an m-major walk cycles all N / 256 weight panels - 4.7 MB at N = 3072,
  // more than an XCD's L2 - through every round).  tools/raster_bench.py, same box, us: M = 47,360 FFN-in forward 280 -> 265,
  // gelu'-product dgrad 247 -> 233, QKV 171 -> 165; M = 23,808: 156 -> 139, 158 -> 144, 84-90 -> 77; M = 12,032 (192-wide
  // tiles): 73-82 -> 64-65, 76-80 -> 65-66, QKV unchanged.  The ring kernel (deeper pipeline, N = 768: 4 n-tiles) gains nothing.
  const int tiles_m = p.M >> 8;
  const int gn_auto = std::max(2, (256 + tiles_m - 1) / std::max(tiles_m, 1));
  q.gn = std::min((p.gn > 0) ? p.gn : gn_auto, tiles_n);
  const int nwork = (p.M >> 8) * tiles_n;
  dim3 grid(std::min(nwork, 256), 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), LDS, st, q);
  return (int)hipGetLastError();
}

template <int NTW>
int dispatch8w_mx(const GemmParams& p, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch8w<EPI_BF16, NTW, false, true>(p, st);
    case EPI_BF16_GELU:
      if (p.aux_u8) return p.out2 != nullptr ? launch8w<EPI_BF16_GELU_U8, NTW, false, true>(p, st) : VAULT_EINVAL;
      return p.out2 != nullptr ? launch8w<EPI_BF16_GELU, NTW, false, true>(p, st)
                               : launch8w<EPI_BF16_GELU_INF, NTW, false, true>(p, st);
    case EPI_F32_RES: return launch8w<EPI_F32_RES, NTW, false, true>(p, st);
    default: return VAULT_EINVAL;
  }
}

template <int NTW>
int dispatch8w(const GemmParams& p, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_BF16: return launch8w<EPI_BF16, NTW>(p, st);
    case EPI_BF16_GELU:
      if (p.aux_u8) return p.out2 != nullptr ? launch8w<EPI_BF16_GELU_U8, NTW>(p, st) : VAULT_EINVAL;
      return p.out2 != nullptr ? launch8w<EPI_BF16_GELU, NTW>(p, st) : launch8w<EPI_BF16_GELU_INF, NTW>(p, st);
    case EPI_BF16_DGELU:
      return p.aux_u8 ? launch8w<EPI_BF16_DGELU_U8, NTW>(p, st) : launch8w<EPI_BF16_DGELU, NTW>(p, st);
    case EPI_F32_RES: return launch8w<EPI_F32_RES, NTW>(p, st);
    default: return VAULT_EINVAL;
  }
}

}  // namespace

// the shapes / epilogues this kernel takes (else the caller falls back to the ring / double-buffered kernels);
// ntw = 4: 256-wide tiles, 3: 192-wide
bool vault_gemm8w_supports(const GemmParams& p, int a_mode, int b_mode, int epi, int ntw) {
  if (a_mode != 0 || b_mode != 0 || p.out_q != nullptr || p.out_scale != nullptr) return false;
  if ((p.M & 255) || (p.N % (64 * ntw)) || (p.K & 63) || p.K < 128 || p.splits > 1 || p.split3 || p.batch > 1) return false;
  if ((long long)p.M * p.ldo * 4 >= (1ll << 32) || p.m_valid < 1) return false;
  if (p.bias == nullptr && p.N > 8192) return false;   // (W8_ZERO_BIAS)   // 32-bit byte offsets in the epilogue
  if (p.aux_u8 && !(epi == EPI_BF16_DGELU || (epi == EPI_BF16_GELU && p.out2 != nullptr))) return false;
  if (epi == EPI_F32_RES) return p.res != nullptr && p.drop_thresh == 0u && p.colsum == nullptr;
  if (epi == EPI_BF16_DGELU) return p.aux != nullptr;
  if (epi == EPI_BF16_GELU) return p.colsum == nullptr;
  return epi == EPI_BF16;
}

// MXFP8 operands (p.A / p.B: e4m3 bytes, lda / ldb in bytes; p.a_scale / p.b_scale: E8M0 [rows][lds_*]): the forward epilogues
bool vault_gemm8w_mx_supports(const GemmParams& p, int epi, int ntw) {
  if ((p.M & 255) || (p.N % (64 * ntw)) || (p.K & 127) || p.K < 384 || p.splits > 1 || p.split3 || p.batch > 1) return false;
  if ((p.lda & 15) || (p.ldb & 15) || p.a_scale == nullptr || p.b_scale == nullptr || p.lds_a < p.K / 32 || p.lds_b < p.K / 32 ||
      (p.lds_a & 3) || (p.lds_b & 3))
    return false;
  if ((long long)p.M * p.ldo * 4 >= (1ll << 32) || (long long)p.M * p.lda >= (1ll << 32) || p.m_valid < 1) return false;
  if (p.bias == nullptr && p.N > 8192) return false;
  if (p.aux_u8 && !(epi == EPI_BF16_GELU && p.out2 != nullptr)) return false;
  if (p.out_hm && !(epi == EPI_BF16 && p.out_hm >= p.M && p.N % 64 == 0 &&
                    (long long)(p.N / 64) * p.out_hm * 128 < (1ll << 32)))
    return false;
  // the MXFP8 image of the output: 256-wide tiles (the four lanes g of a row = one block of 32 columns), GELU epilogues
  if ((p.out_q == nullptr) != (p.out_scale == nullptr)) return false;
  if (p.out_q != nullptr && (ntw != 4 || epi != EPI_BF16_GELU || (long long)p.M * p.N >= (1ll << 32))) return false;
  if (epi == EPI_F32_RES) return p.res != nullptr && p.drop_thresh == 0u && p.colsum == nullptr;
  if (epi == EPI_BF16_GELU) return p.colsum == nullptr;
  return epi == EPI_BF16 && p.colsum == nullptr;
}

int vault_gemm8w_mx_launch(const GemmParams& p, int epi, int ntw, hipStream_t st) {
  return ntw == 3 ? dispatch8w_mx<3>(p, epi, st) : dispatch8w_mx<4>(p, epi, st);
}

int vault_gemm8w_launch(const GemmParams& p, int epi, int ntw, hipStream_t st) {
  return ntw == 3 ? dispatch8w<3>(p, epi, st) : dispatch8w<4>(p, epi, st);
}
