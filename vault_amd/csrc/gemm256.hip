// 256x256x64 bf16 MFMA GEMM, 4 waves (2 x 2, one per SIMD, each owning the SIMD's whole 512-entry
// register file: 256 accumulator registers + fragments), software-pipelined in 8 phases per two K tiles.
//
// Same operand modes and epilogues as gemm.hip, for the large shapes of the path (M = B*185 tokens).
// What is different from the simple double-buffered kernel:
//   * each 64-deep K tile lives in LDS as four 16 KiB half-tiles (A0/A1 = the upper/lower 64 rows of
//     every wave's 128 output rows, B0/B1 = the left/right 64 columns of every wave's 128 columns);
//     two K tiles (128 KiB) form a ring of eight half-tile slots;
//   * one half-tile (4 global_load_lds per wave) is issued per phase, six phases before its first
//     reader, and retired with a COUNTED `s_waitcnt vmcnt(20)` (five half-tiles stay in flight across
//     every barrier) - the L2/HBM latency is hidden behind ~1.5 K tiles of MFMA work instead of one;
//   * each phase multiplies one quadrant (64 x 64 per wave, 32 MFMAs) while the fragments of the next
//     quadrant are read from LDS; quadrants are walked 00,01,11,10 | 01,00,10,11 so that every
//     fragment set is loaded exactly once per K tile and two of the four sets are always reusable.
// The grid is persistent (<= 256 blocks): block b walks work items b, b + grid, ...  With (GemmParams.persist & 1)
// the items are handed out DYNAMICALLY instead: the items with id % 8 == x belong to XCD x (contiguous tile runs
// per L2, see gemm_tile_of_block); block b starts with item b and then draws tickets from its XCD's counter (one
// returning atomic per tile), stealing from the other XCDs' counters when its own is exhausted.  A block that
// cannot get a CU at launch (another kernel - an RCCL collective on a second stream - holds it) then costs its
// share of throughput, not a second pass over a statically assigned tile list: the mode for data-parallel runs.  The epilogue of one
// tile overlaps the start of the next: its LDS scratch lies outside the ring, one half-tile of the next
// tile's first two K tiles is issued per 16-row epilogue step, and the global stores of the epilogue are
// never waited for explicitly - VMEM operations retire in issue order, so the counted waits of the next
// tile's first five phases simply allow for the stores that are certain to have been issued behind the
// half-tile they retire (a lower bound: `SPS` per epilogue step of a fully valid tile, else none).  The
// stores drain to L2/HBM while the matrix pipe is already working on the next tile.
// Ordering rules (all formal, no timing assumptions):
//   RAW  a half-tile is read one phase after the `vmcnt` + `s_barrier` that retired it;
//   WAR  a slot is overwritten two phases after its ds_reads were issued, and every phase ends with
//        `lgkmcnt(0)` before the barrier, so those reads have completed in every wave.
#include <algorithm>
#include "common.h"
#include "gemm.h"
#include "gemm_epi.h"

#ifndef R256_ABLATE
#define R256_ABLATE 0
#endif
// Two translation units compile this file (round 6): gemm256.hip itself (R256_DYN 0: the static walk only - every block takes
// items b, b + grid, ...) and gemm256_dyn.hip (R256_DYN 1: the kernels that can also hand their items out dynamically, for
// data-parallel steps).  The scheduler's state (ticket in hand, last ticket, counters, per-XCD list sizes) lives across the
// main loop in SGPRs the kernel does not have: with it in every instantiation all 256-wide forms spilled 3-7 registers
// (compiler-issued scratch traffic in the `vmcnt` the asm loads are counted in); without it they do not.
#ifndef R256_DYN
#define R256_DYN 0
#endif

// Diagnostic build only (-DR256_STAMP=1, tools/clock_stamp.py; no stamp executes in the shipped kernel): every block leaves the
// shader-clock and the 100 MHz real-time ticks it ran for - the clock the chip HOLDS under this kernel is their quotient
// (MI355X_MICROARCH.md "DVFS give-back" item 6).  The values go to a buffer nothing else reads.
#ifndef R256_STAMP
#define R256_STAMP 0
#endif
#if R256_DYN   // (the stamps are taken in the static translation unit only: one definition of the buffer and its reader)
#undef R256_STAMP
#define R256_STAMP 0
#endif
#if R256_STAMP
__device__ unsigned long long g_r256_stamp[256 * 2];
extern "C" int vault_debug_r256_stamps(unsigned long long* out512) {
  return (int)hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_r256_stamp), sizeof(g_r256_stamp));
}
#endif

namespace {

// Dynamic tile scheduler state (a `__device__` global: one copy per device; the launcher grants dynamic hand-out to
// ONE stream per device, whose kernels are serialised - see RingSched below): two sets of eight per-XCD ticket counters.  Launch k draws from set k & 1 and clears the
// other set for launch k + 1 (no end-of-kernel reset, no "last block" bookkeeping).
__device__ unsigned int g_ring_tickets[2][8];

constexpr int HT = 16384;      // half-tile bytes
constexpr int BUFB = 4 * HT;   // one K tile: A0 A1 B0 B1

// One asm statement: counted vmcnt (hipcc has no counted form for LDS-DMA), lgkmcnt(0), barrier.  hipcc does
// not see these waits: fragment reads are therefore issued between the two halves of a phase's MFMA
// cluster, after the point where hipcc places its own (already satisfied) lgkmcnt wait.
#define WAITBAR(N) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"i"(N) : "memory")
// in front of an asm VMEM statement whose SGPR base hipcc may have restored from a VGPR lane right before it (gemm8w.hip
// W8_SGPR_PAD; vault_amd/isa_check.py fails the build where one is missing): the slab accesses of the split-K hand-off
#define R256_SGPR_PAD "s_nop 4\n\t"

// The same with the count selected by a uniform flag INSIDE the asm statement (relative branches): a C++ branch
// here would cut the loop body into basic blocks, and hipcc then shuffles the accumulators between register
// files at the block boundaries.
#define WAITBAR2(FLAG, NF, N)                                                                                  \
  asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 2\n\ts_waitcnt vmcnt(%1) lgkmcnt(0)\n\ts_branch 1\n"     \
               "\ts_waitcnt vmcnt(%2) lgkmcnt(0)\n\ts_barrier" ::"s"(__builtin_amdgcn_readfirstlane((int)(FLAG))), "i"(NF), "i"(N) \
               : "memory", "scc")   /* branch operands = dwords to skip: [wait NF, s_branch] / [wait N] */

// ds_read_b64_tr_b16 through asm: the builtin form makes hipcc drain every in-flight global_load_lds
// (vmcnt(0)) in front of it, which would serialise the ring.  The data is consumed only after the
// phase-ending WAITBAR (lgkmcnt(0)), by the MFMA asm of a later phase.
template <int OFF>
__device__ __forceinline__ s16x4 tr16_asm(uint32_t lds_addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(OFF));
  return v;
}

#ifndef R256_ORDER
#define R256_ORDER -1     // issue order inside a phase: see PHASE_BODY
#endif
// NTQ = 16-column MFMA tiles per wave per B half: 4 -> 256-wide block tile, 3 -> 192-wide (N = 768 gives
// 4 x 185 = 740 tiles = 2.9 rounds of 256 CUs instead of 555 = 2.2 rounds: 96 % instead of 72 % of the
// last round's CUs busy)
// SK (round 6): the instantiation also takes SPLIT-K work items of the 16-bit / f32-residual epilogues (GemmParams::splits > 1 with
// a workspace, sk_ws): an item contracts one K range of its tile; every split stores its accumulators as they stand into its
// slab of the workspace (write-through stores), the block's last wave-0 lane then draws a ticket from the tile's counter, and
// the split whose ticket is the last one adds the slabs up (ascending z: the result does not depend on who came last) and runs
// the epilogue - the in-launch reduction of cdna_hip_programming.md's projection-GEMM recipe (sc1 stores -> vmcnt(0) -> barrier ->
// one relaxed agent-scope add; the reducer reads with sc1 loads).  Nobody waits for anybody: no residency assumption.  For the
// N = 768 Linears with long contractions at small batches: 96 tiles x 48 K tiles at B = 32 are one 37 %-full round, 192 items
// of 24 K tiles fill 75 % of the CUs for half the time.  A separate instantiation, so that the un-split kernels keep their code.
template <int A_MODE, int B_MODE, int EPI, int NTQ, bool DYN, bool SK = false>
__global__ __launch_bounds__(256) void gemm256_kernel(const GemmParams p) {
  H16_SATURATE();
#if R256_STAMP
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int BNT = NTQ * 64;                     // block tile width
  constexpr int PH_ORDER = (R256_ORDER >= 0) ? R256_ORDER : (EPI == EPI_F32_ATOMIC ? 0 : 1);   // (PHASE_BODY, below)
  constexpr int PB = (B_MODE == 0) ? NTQ : 4;       // global_load_lds pieces per wave per B half-tile
  constexpr int W32 = 3 * 4 + 2 * PB;               // pieces of the five youngest half-tiles: 3 A + 2 B
  constexpr int W23 = 2 * 4 + 3 * PB;               //                                       2 A + 3 B
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  constexpr int TN = 2 * NTQ;
  constexpr int SCRATCH_BYTES = 4 * 16 * ((NTQ == 3 ? 96 : 64) + 4) * 4;   // epilogue scratch (gemm_epi.h), then 16 B scheduler word
  constexpr bool F32OUT = (EPI == EPI_F32_RES || EPI == EPI_F32_PATCH);
  // VMEM operations every wave is certain to issue in one 16-row epilogue step of a fully valid tile
  constexpr int SPS = (EPI == EPI_F32_ATOMIC) ? 16 * (TN * 16 / 64) : (F32OUT ? TN : TN / 2);
#define CAPW(N, STEPS) (((N) + (STEPS) * SPS) > 63 ? 63 : ((N) + (STEPS) * SPS))

  // ---- work items: (split z, tile) pairs, z-major; every split owns >= 1 K tile (launcher)
  const int tiles_m = p.M >> 8, tiles_n = p.N / BNT, ntiles = tiles_m * tiles_n;
  // (batched weight gradients, EPI_F32_ATOMIC: `batch` problems of one shape, problem-major in the work list - the
  //  XCD-contiguous renumbering then gives an XCD whole problems, so a problem's panels fill ONE L2)
  const int nbatch = (EPI == EPI_F32_ATOMIC && p.batch > 1) ? p.batch : 1;
  // grouped weight gradients (GemmParams::seg): the list is the concatenation of the segments' items, z-major as below
  bool grouped = false;
  int seg_total = 0;
  if constexpr (EPI == EPI_F32_ATOMIC && A_MODE == 1 && B_MODE == 1) {
    grouped = p.nseg > 0;
    seg_total = p.seg[0].count + (p.nseg > 1 ? p.seg[1].count : 0) + (p.nseg > 2 ? p.seg[2].count : 0);
  }
  const int nwork = grouped ? seg_total * p.splits : ntiles * p.splits * nbatch;
  const int nk_total = p.K >> 6;
  const int per = (nk_total + p.splits - 1) / p.splits;

  // ---- staging sources: uniform 64-bit base (SGPRs) + per-lane 32-bit byte offset, two 1-KiB pieces
  //      per wave per half-tile
  int m0, n0, nk;
  int cur_z = 0, cur_tl = 0;                          // SK: split and tile of the current work item
  float* out_cur = reinterpret_cast<float*>(p.out);   // EPI_F32_ATOMIC: this work item's problem (batched launches)
  int ldo_cur = p.ldo, mvalid_cur = p.m_valid;        // ... and its output geometry (grouped launches)
  const char* a_base;
  const char* b_base;
  uint32_t a_off[4], b_off[4];
  uint32_t a_half, b_half, a_step, b_step;   // bytes
  // (A_MODE 0 with a head-major A, p.a_hm rows per plane [K / 64][a_hm][64]: a K tile is one plane, lda = 64 - the launcher)
  if constexpr (A_MODE == 0) { a_half = 64u * p.lda * 2u; a_step = p.a_hm ? (uint32_t)p.a_hm * 128u : 128u; }
  else { a_half = 128u; a_step = 64u * p.lda * 2u; }
  if constexpr (B_MODE == 0) { b_half = (uint32_t)(NTQ * 16) * p.ldb * 2u; b_step = 128u; }
  else { b_half = (uint32_t)(NTQ * 16) * 2u; b_step = 64u * p.ldb * 2u; }
  // per-lane staging offsets of the four 1-KiB pieces per half-tile (they depend on the leading dimensions: recomputed
  // per work item in grouped launches, from an opaque copy of the lane id so that nothing extra stays live)
  auto lane_offsets = [&](int lda_, int ldb_, int a_hm_ = 0) {
    const int ln_ = lane_id_volatile();
    int wv_ = wave;     // (opaque too: the splat of 16 wave would otherwise be kept in a VGPR across the main loop - and spilled)
    asm volatile("" : "+s"(wv_));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = wv_ * 4 + i;
      const int r8 = ln_ >> 3, r = 8 * j + r8;
      const int c = (ln_ & 7) ^ (((r8 >> 1) & 3) << 1);
      const int krow = 4 * j + (ln_ >> 4), pos16 = ln_ & 15;
      const int hk = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int cb = (pos16 >> 1) ^ hk;
      if constexpr (A_MODE == 0) a_off[i] = (uint32_t)(((r >> 6) * 128 + (r & 63)) * lda_ + c * 8) * 2u;
      else if (a_hm_ == 0) a_off[i] = (uint32_t)(krow * lda_ + (cb >> 2) * 128 + (cb & 3) * 16 + (pos16 & 1) * 8) * 2u;
      // head-major dY [M / 64][a_hm][64]: column (cb >> 2) * 128 + 64 h + (cb & 3) * 16 + .. lies in plane 2 (cb >> 2) + h
      else a_off[i] = ((uint32_t)krow * 64u + (uint32_t)(cb >> 2) * 2u * (uint32_t)a_hm_ * 64u + (uint32_t)((cb & 3) * 16 + (pos16 & 1) * 8)) * 2u;
      if constexpr (B_MODE == 0) {
        // half-tile image: 2 wave-columns x NTQ*16 rows of 128 B; piece jb = wave*PB + i covers rows 8jb..8jb+7
        const int jb = wv_ * PB + (i < PB ? i : PB - 1), rb = 8 * jb + r8;
        const int wcol = rb / (NTQ * 16), within = rb - wcol * (NTQ * 16);
        b_off[i] = (uint32_t)((wcol * (NTQ * 32) + within) * ldb_ + c * 8) * 2u;
      } else {
        // [64 k][8 col-blocks of 16] image, blocks wc*4 + nt; nt >= NTQ slots are never read: clamp their source
        const int ntc = min(cb & 3, NTQ - 1);
        b_off[i] = (uint32_t)(krow * ldb_ + (cb >> 2) * (NTQ * 32) + ntc * 16 + (pos16 & 1) * 8) * 2u;
      }
    }
  };
  auto setup = [&](int w) {
    if constexpr (EPI == EPI_F32_ATOMIC && A_MODE == 1 && B_MODE == 1) {
      if (grouped) {
        const int lin = gemm_xcd_contiguous(nwork, w);
        const int z = lin / seg_total;
        int r = lin - z * seg_total;
        // segment of list position r (uniform selects: no dynamic indexing of the kernel argument)
        const int c0 = p.seg[0].count, c1 = (p.nseg > 1) ? p.seg[1].count : 0;
        const int si = (r < c0) ? 0 : ((r < c0 + c1) ? 1 : 2);
        r -= (si == 0) ? 0 : (si == 1 ? c0 : c0 + c1);
#define SEGF(F) ((si == 0) ? p.seg[0].F : ((si == 1) ? p.seg[1].F : p.seg[2].F))
        const int idx = SEGF(first) + r, tiles_k = SEGF(tiles), tn_k = SEGF(tiles_n);
        const int bz = idx / tiles_k, tl = idx - bz * tiles_k;
        const int lda_ = SEGF(lda), ldb_ = SEGF(ldb), ahm_ = SEGF(a_hm);
#if R256_ABLATE & 1   // development: every item contracts the first layer's first panels (operands stay in L2)
        const h16* pA = SEGF(A);
        const h16* pB = SEGF(B);
#else
        const h16* pA = SEGF(A) + (size_t)bz * SEGF(batch_a);
        const h16* pB = SEGF(B) + (size_t)bz * SEGF(batch_b);
#endif
        out_cur = SEGF(out) + (size_t)bz * SEGF(batch_o);
        ldo_cur = SEGF(ldo);
        mvalid_cur = (tiles_k / tn_k) << 8;            // every row of the kind's dW exists
#undef SEGF
        // tiles of a problem are walked with the SHORT side of its tile grid innermost: a run of consecutive items (an XCD's 32)
        // that is cut off inside a problem then streams few whole panels of the long side instead of all of them - FFN-out's
        // 3 x 12 grid column by column: 772 -> 700 panel fetches per stack against 576 if every panel were fetched once per
        // layer (tools: the count in DESIGN 4.1)
        const int tm_k = tiles_k / tn_k;
        int tile_m, tile_n;
        if (tn_k > tm_k) { tile_n = tl / tm_k; tile_m = tl - tile_n * tm_k; }
        else { tile_m = tl / tn_k; tile_n = tl - tile_m * tn_k; }
        m0 = tile_m << 8; n0 = tile_n << 8;
        const int kt0 = z * per;
        nk = min(nk_total, kt0 + per) - kt0;
        a_step = 64u * (uint32_t)lda_ * 2u; b_step = 64u * (uint32_t)ldb_ * 2u;       // (head-major dY: lda_ = 64)
        a_half = ahm_ ? (uint32_t)ahm_ * 128u : 128u;
        lane_offsets(lda_, ldb_, ahm_);
#if R256_ABLATE & 1
        const int lm0 = 0, ln0 = 0;
#else
        const int lm0 = m0, ln0 = n0;
#endif
        a_base = ahm_ ? reinterpret_cast<const char*>(pA + ((size_t)(lm0 >> 6) * ahm_ + (size_t)kt0 * 64) * 64)
                      : reinterpret_cast<const char*>(pA + (size_t)kt0 * 64 * lda_ + lm0);
        b_base = reinterpret_cast<const char*>(pB + (size_t)kt0 * 64 * ldb_ + ln0);
        return;
      }
    }
    // the XCD-contiguous renumbering runs over the WHOLE work list (z-major): with split-K an XCD then works on
    // one or two K ranges only, so the A / B panels of a range are fetched into one or two L2s instead of all
    // eight (weight gradients, 36 tiles x 7 splits: L2 fill 970 -> ~460 MB per launch by this count)
    int lin = gemm_xcd_contiguous(nwork, w);
    const h16* pA = p.A;
    const h16* pB = p.B;
    if constexpr (EPI == EPI_F32_ATOMIC) {
      if (nbatch > 1) {
        const int per_problem = ntiles * p.splits;
        const int bz = lin / per_problem;
        lin -= bz * per_problem;
        pA += (size_t)bz * p.batch_a; pB += (size_t)bz * p.batch_b;
        out_cur = reinterpret_cast<float*>(p.out) + (size_t)bz * p.batch_o;
      }
    }
    const int z = lin / ntiles, tl = lin - z * ntiles;
    if constexpr (SK) { cur_z = z; cur_tl = tl; }
    int tile_m, tile_n;
    gemm_raster(tl, tiles_m, tiles_n, p.gn, tile_m, tile_n);
    m0 = tile_m << 8; n0 = tile_n * BNT;
    const int kt0 = z * per;
    nk = min(nk_total, kt0 + per) - kt0;
#if R256_ABLATE & 1   // development: every tile loads the first row / column panel (operands stay in L2)
    const int lm0 = 0, ln0 = 0;
#else
    const int lm0 = m0, ln0 = n0;
#endif
    if constexpr (A_MODE == 0) a_base = p.a_hm ? reinterpret_cast<const char*>(pA + ((size_t)kt0 * p.a_hm + lm0) * 64)
                                               : reinterpret_cast<const char*>(pA + (size_t)lm0 * p.lda + (size_t)kt0 * 64);
    else a_base = reinterpret_cast<const char*>(pA + (size_t)kt0 * 64 * p.lda + lm0);
    if constexpr (B_MODE == 0) b_base = reinterpret_cast<const char*>(pB + (size_t)ln0 * p.ldb + (size_t)kt0 * 64);
    else b_base = reinterpret_cast<const char*>(pB + (size_t)kt0 * 64 * p.ldb + ln0);
  };
  int w = blockIdx.x;
  setup(w);
  // dynamic scheduler: items of XCD x are x + 8 j ; the first n_static_x of them are the blocks' own start items
  const int xcd = blockIdx.x & 7;
  auto items_of = [&](int x) { return x < nwork ? (nwork - x + 7) >> 3 : 0; };          // j < items_of(x)
  auto static_of = [&](int x) { return x < (int)gridDim.x ? ((int)gridDim.x - x + 7) >> 3 : 0; };
  // a ticket stands for TICKET_ITEMS consecutive items of an XCD's list: one returning atomic per that many tiles.
  // Work lists of at most two items per block (batched weight gradients: 486 items on 256 blocks) take single-item
  // tickets: pairs would give half of the blocks three items and the rest one - three rounds instead of two.
  const int TICKET_ITEMS = (nwork <= 2 * (int)gridDim.x) ? 1 : 2;
  int pend_w = -1;           // thread 0: second item of the current ticket, not yet started
  int last_tk = 0;           // thread 0: last ticket drawn from the own list
  const bool dyn = DYN && (p.persist & 1) != 0;   // dynamic hand-out of work items (else: block b walks b, b + grid, ...)
  unsigned int* const tickets = g_ring_tickets[(p.persist >> 8) & 1];   // bit 8: launch parity (set by the launcher)
  if (dyn && blockIdx.x == 0 && tid < 8) g_ring_tickets[((p.persist >> 8) & 1) ^ 1][tid] = 0u;
  if (!grouped) lane_offsets(p.lda, p.ldb);

  // one 1-KiB piece (i = 0..3) of a half-tile; a half-tile is 4 pieces per wave
  // The per-lane offset passes through an opaque move at every use: hipcc otherwise hoists its 64-bit
  // extension out of the persistent loop (8 register pairs live across everything) instead of folding the
  // 32-bit offset into the SGPR-base addressing form.
  auto pieceA = [&](int h, int buf, int t, int i) {
    uint32_t o = a_off[i];
    asm volatile("" : "+v"(o));
    glds16(a_base + (size_t)h * a_half + (size_t)t * a_step + o, smem + buf * BUFB + h * HT + wave * 4096 + i * 1024);
  };
  auto pieceB = [&](int h, int buf, int t, int i) {
    if (i < PB) {
      uint32_t o = b_off[i];
      asm volatile("" : "+v"(o));
      glds16(b_base + (size_t)h * b_half + (size_t)t * b_step + o,
             smem + buf * BUFB + 2 * HT + h * HT + (wave * PB + i) * 1024);
    }
  };
  auto issueA = [&](int h, int buf, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pieceA(h, buf, t, i);
  };
  auto issueB = [&](int h, int buf, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pieceB(h, buf, t, i);
  };

  // BUF / H are compile-time at every call site (macro-expanded), so the slot offset folds into the
  // 16-bit DS immediate; ring buffer 1 (+64 KiB) exceeds it and is added to the address instead
#define LOAD_TR(DST, ADDR, BUF, H)                                                                       \
  {                                                                                                      \
    const uint32_t ad_ = (ADDR) + (BUF) * BUFB;                                                          \
    DST[0] = cat_tr(tr16_asm<(H) * HT>(ad_), tr16_asm<(H) * HT + 1024>(ad_));                            \
    DST[1] = cat_tr(tr16_asm<(H) * HT + 8192>(ad_), tr16_asm<(H) * HT + 8192 + 1024>(ad_));             \
  }
#define LOADA_U(RA, BUF, H, U)                                                                           \
  {                                                                                                      \
    if constexpr (A_MODE == 0) {                                                                         \
      const char* base_ = smem + (BUF) * BUFB + (H) * HT;                                                \
      RA[U][0] = *LDS_PTR(const h16x8, base_ + a_o0 + (U) * 2048);                                      \
      RA[U][1] = *LDS_PTR(const h16x8, base_ + a_o1 + (U) * 2048);                                      \
    } else {                                                                                             \
      LOAD_TR(RA[U], a_tr[U], BUF, H)                                                                    \
    }                                                                                                    \
  }
#define LOADB_U(RB, BUF, H, U)                                                                           \
  if constexpr ((U) < NTQ) {                                                                             \
    if constexpr (B_MODE == 0) {                                                                         \
      const char* base_ = smem + (BUF) * BUFB + 2 * HT + (H) * HT;                                       \
      RB[U][0] = *LDS_PTR(const h16x8, base_ + b_o0 + (U) * 2048);                                      \
      RB[U][1] = *LDS_PTR(const h16x8, base_ + b_o1 + (U) * 2048);                                      \
    } else {                                                                                             \
      LOAD_TR(RB[U], b_tr[U], BUF, H)                                                                    \
    }                                                                                                    \
  }
#define LOADA(RA, BUF, H) { LOADA_U(RA, BUF, H, 0) LOADA_U(RA, BUF, H, 1) LOADA_U(RA, BUF, H, 2) LOADA_U(RA, BUF, H, 3) }
#define LOADB(RB, BUF, H) { LOADB_U(RB, BUF, H, 0) LOADB_U(RB, BUF, H, 1) LOADB_U(RB, BUF, H, 2) LOADB_U(RB, BUF, H, 3) }

  f32x4 acc[8][2 * NTQ];
  // MFMAs are issued through inline asm with the accumulator tied in place in the AGPR half of the
  // register file ("+a"): hipcc otherwise allocates out-of-place destinations for this many live
  // accumulators and spills.  `volatile` keeps every cluster inside its phase.

  h16x8 RA0[4][2], RA1[4][2], RB0[4][2], RB1[4][2];

  // the first two K tiles of the current work item in ring order, one half-tile per call (h8 = 0..7)
  auto stage_first = [&](int h8) {
    const int t1 = nk > 1 ? 1 : 0;   // single-K-tile items stage tile 0 twice (never read): no branches, uniform waits
    switch (h8) {
      case 0: issueA(0, 0, 0); break;
      case 1: issueB(0, 0, 0); break;
      case 2: issueB(1, 0, 0); break;
      case 3: issueA(1, 0, 0); break;
      case 4: issueA(0, 1, t1); break;
      case 5: issueB(1, 1, t1); break;
      case 6: issueB(0, 1, t1); break;
      default: issueA(1, 1, t1); break;
    }
  };
#pragma unroll
  for (int h8 = 0; h8 < 8; ++h8) stage_first(h8);

  // one loop iteration = two K tiles = eight phases.  A phase = 32 MFMAs with the next fragment set's LDS
  // reads and one half-tile's four global_load_lds pieces spread between the 4-MFMA blocks (one wave per
  // SIMD: anything issued in a burst would leave the matrix pipe idle for its whole issue time).
  // There is ONE copy of the loop body (instruction cache: main loop + epilogue must fit 64 KiB):
  //   * every phase always issues its half-tile; past the end of the contraction the K tile index is clamped
  //     (a duplicate of the last tile lands in a slot nobody reads), so the counted waits never change;
  //   * FIRST (first iteration behind an overlapped epilogue) selects the store-tolerant wait of phases 1-5
  //     by a uniform branch.
#define LOAD_TR_H(DST, ADDR, BUF, H, KS)                                                                 \
  {                                                                                                      \
    const uint32_t ad_ = (ADDR) + (BUF) * BUFB;                                                          \
    DST[KS] = cat_tr(tr16_asm<(H) * HT + (KS) * 8192>(ad_), tr16_asm<(H) * HT + (KS) * 8192 + 1024>(ad_)); \
  }
#define LOADA_UH(RA, BUF, H, U, KS)                                                                      \
  {                                                                                                      \
    if constexpr (A_MODE == 0) {                                                                         \
      const char* base_ = smem + (BUF) * BUFB + (H) * HT;                                                \
      RA[U][KS] = *LDS_PTR(const h16x8, base_ + ((KS) ? a_o1 : a_o0) + (U) * 2048);                     \
    } else {                                                                                             \
      LOAD_TR_H(RA[U], a_tr[U], BUF, H, KS)                                                              \
    }                                                                                                    \
  }
#define LOADB_UH(RB, BUF, H, U, KS)                                                                      \
  if constexpr ((U) < NTQ) {                                                                             \
    if constexpr (B_MODE == 0) {                                                                         \
      const char* base_ = smem + (BUF) * BUFB + 2 * HT + (H) * HT;                                       \
      RB[U][KS] = *LDS_PTR(const h16x8, base_ + ((KS) ? b_o1 : b_o0) + (U) * 2048);                     \
    } else {                                                                                             \
      LOAD_TR_H(RB[U], b_tr[U], BUF, H, KS)                                                              \
    }                                                                                                    \
  }
  // one MFMA; GA(MT, KS) = the group of NTQ MFMAs of one A fragment, with two insertion points (after the
  // first and after the third MFMA) for one staging piece or one fragment read each: never more than one
  // memory instruction (+ its address arithmetic) between two MFMAs
#define MMA1(KS, MT, NT, MB, NB, RA, RB)                                                   \
  if constexpr ((NT) < NTQ)                                                                \
    asm volatile(MFMA16_ASM " %0, %1, %2, %0"                                 \
                 : "+a"(acc[(MB) + (MT)][(NB) + (NT)])                                     \
                 : "v"(RA[MT][KS]), "v"(RB[NT][KS]));
#define GA(KS, MT, MB, NB, RA, RB, INS_A, INS_B)                                           \
    MMA1(KS, MT, 0, MB, NB, RA, RB) INS_A;                                                 \
    MMA1(KS, MT, 1, MB, NB, RA, RB) MMA1(KS, MT, 2, MB, NB, RA, RB) INS_B;                 \
    MMA1(KS, MT, 3, MB, NB, RA, RB)
// Placement of a phase's four staging pieces against its eight fragment reads.  1: the reads in the first four MFMA groups, the
// pieces in gaps of their own behind them - in the step 38.02 -> 37.91 ms (three interleaved same-box pairs, tools/r05_call15.sh);
// 0: a piece and a read per group (rounds 1-4); 2: pieces first (slower).  tools/pf_bench.py: within 1 % on the isolated kernels
// either way - the kernel is bound by the aggregate of its address path, not by where in a phase its instructions sit.
// Default (R256_ORDER undefined or < 0): 1 for the forward and data-gradient instantiations, 0 for the weight gradients - there 1 is
// worth 3.4 us of a 575 us launch but raises the HBM fetch of a launch by 8-12 % (1,019 -> 1,105-1,140 MB, same-box --pmc passes,
// tools/r05_call27.sh: the pieces of a phase leave later, the blocks of an XCD drift apart and share fewer L2 hits).
#define PHASE_BODY_0(LOADUH, RN, LBUF, LH, PIECE, IH, IBUF, IT, MB, NB, RA, RB) \
    GA(0, 0, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 0), LOADUH(RN, LBUF, LH, 0, 0)) \
    GA(0, 1, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 1), LOADUH(RN, LBUF, LH, 1, 0)) \
    GA(0, 2, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 2), LOADUH(RN, LBUF, LH, 2, 0)) \
    GA(0, 3, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 3), LOADUH(RN, LBUF, LH, 3, 0)) \
    GA(1, 0, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 0, 1), LOADUH(RN, LBUF, LH, 1, 1)) \
    GA(1, 1, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 2, 1), LOADUH(RN, LBUF, LH, 3, 1)) \
    GA(1, 2, MB, NB, RA, RB, , )                                               \
    GA(1, 3, MB, NB, RA, RB, , )
// the reads first, the pieces in gaps of their own
#define PHASE_BODY_1(LOADUH, RN, LBUF, LH, PIECE, IH, IBUF, IT, MB, NB, RA, RB) \
    GA(0, 0, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 0, 0), LOADUH(RN, LBUF, LH, 1, 0)) \
    GA(0, 1, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 2, 0), LOADUH(RN, LBUF, LH, 3, 0)) \
    GA(0, 2, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 0, 1), LOADUH(RN, LBUF, LH, 1, 1)) \
    GA(0, 3, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 2, 1), LOADUH(RN, LBUF, LH, 3, 1)) \
    GA(1, 0, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 0), )                         \
    GA(1, 1, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 1), )                         \
    GA(1, 2, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 2), )                         \
    GA(1, 3, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 3), )
// the pieces first, in gaps of their own, then the reads
#define PHASE_BODY_2(LOADUH, RN, LBUF, LH, PIECE, IH, IBUF, IT, MB, NB, RA, RB) \
    GA(0, 0, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 0), )                         \
    GA(0, 1, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 1), )                         \
    GA(0, 2, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 2), )                         \
    GA(0, 3, MB, NB, RA, RB, PIECE(IH, IBUF, IT, 3), )                         \
    GA(1, 0, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 0, 0), LOADUH(RN, LBUF, LH, 1, 0)) \
    GA(1, 1, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 2, 0), LOADUH(RN, LBUF, LH, 3, 0)) \
    GA(1, 2, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 0, 1), LOADUH(RN, LBUF, LH, 1, 1)) \
    GA(1, 3, MB, NB, RA, RB, LOADUH(RN, LBUF, LH, 2, 1), LOADUH(RN, LBUF, LH, 3, 1))
#define PHASE_BODY(...)                                                        \
    if constexpr (PH_ORDER == 0) { PHASE_BODY_0(__VA_ARGS__) }                 \
    else if constexpr (PH_ORDER == 1) { PHASE_BODY_1(__VA_ARGS__) }            \
    else { PHASE_BODY_2(__VA_ARGS__) }
#define PHASE(WAITN, WAITF, LOADUH, RN, LBUF, LH, PIECE, IH, IBUF, IT, MB, NB, RA, RB) \
  {                                                                            \
    __builtin_amdgcn_s_setprio(1);                                             \
    asm volatile("s_nop 1");                                                   \
    PHASE_BODY(LOADUH, RN, LBUF, LH, PIECE, IH, IBUF, IT, MB, NB, RA, RB)      \
    __builtin_amdgcn_s_setprio(0);                                             \
    if constexpr ((WAITF) != (WAITN)) { WAITBAR2(first, WAITF, WAITN); } else { WAITBAR(WAITN); } \
  }
#define HALF_A                                                                                       \
    PHASE(W32, CAPW(W32, 5), LOADB_UH, RB1, 0, 1, pieceA, 0, 0, t2, 0, 0, RA0, RB0)                   \
    PHASE(W23, CAPW(W23, 4), LOADA_UH, RA1, 0, 1, pieceB, 0, 0, t2, 0, NTQ, RA0, RB1)                 \
    PHASE(W23, CAPW(W23, 3), LOADA_UH, RA0, 1, 0, pieceB, 1, 0, t2, 4, NTQ, RA1, RB1)                 \
    PHASE(W32, CAPW(W32, 2), LOADB_UH, RB1, 1, 1, pieceA, 1, 0, t2, 4, 0, RA1, RB0)
  // the same for the last (odd) K tile: no fragments are read ahead in its last two phases.  (Not only to save
  // the reads: their results would be dead, and hipcc - which cannot know that these asm outputs arrive
  // asynchronously - would reuse the destination registers for staging addresses while the data is on its way.)
#define NOLOAD_UH(R, BUF, H, U, KS)
#define HALF_A_LAST                                                                                  \
    PHASE(W32, CAPW(W32, 5), LOADB_UH, RB1, 0, 1, pieceA, 0, 0, t2, 0, 0, RA0, RB0)                   \
    PHASE(W23, CAPW(W23, 4), LOADA_UH, RA1, 0, 1, pieceB, 0, 0, t2, 0, NTQ, RA0, RB1)                 \
    PHASE(W23, CAPW(W23, 3), NOLOAD_UH, RA0, 1, 0, pieceB, 1, 0, t2, 4, NTQ, RA1, RB1)                \
    PHASE(W32, CAPW(W32, 2), NOLOAD_UH, RB1, 1, 1, pieceA, 1, 0, t2, 4, 0, RA1, RB0)
#define HALF_B                                                                                       \
    PHASE(W32, CAPW(W32, 1), LOADB_UH, RB0, 1, 0, pieceA, 0, 1, t3, 0, NTQ, RA0, RB1)                 \
    PHASE(W23, W23, LOADA_UH, RA1, 1, 1, pieceB, 1, 1, t3, 0, 0, RA0, RB0)                            \
    PHASE(W23, W23, LOADA_UH, RA0, 0, 0, pieceB, 0, 1, t3, 4, 0, RA1, RB0)                            \
    PHASE(W32, W32, LOADB_UH, RB0, 0, 0, pieceA, 1, 1, t3, 4, NTQ, RA1, RB1)
  bool behind_stores = false;   // the staged half-tiles were issued between the steps of a fully valid tile's epilogue
  while (true) {
    // ---- fragment read offsets: recomputed per work item from an opaque copy of the lane id, so that they
    //      are not live across the epilogue (where they would spill: the reload would wait for the stores)
    const int ln_ = lane_id_volatile();
    const int g = ln_ >> 4, l15 = ln_ & 15;
    int a_o0, a_o1, b_o0, b_o1;   // mode 0: byte offsets of k-step 0 / 1 ; mode 1: base offset / swizzle key
    {
      const int fx = ((l15 >> 1) & 3) << 1;
      const int q = l15 >> 2, pp = l15 & 3, hk = q | ((g & 1) << 2);
      if constexpr (A_MODE == 0) {
        a_o0 = (wr * 64 + l15) * 128 + ((g ^ fx) << 4);
        a_o1 = (wr * 64 + l15) * 128 + (((4 + g) ^ fx) << 4);
      } else {
        a_o0 = (8 * g + q) * 256 + pp * 8;
        a_o1 = hk;
      }
      if constexpr (B_MODE == 0) {
        b_o0 = (wc * (NTQ * 16) + l15) * 128 + ((g ^ fx) << 4);
        b_o1 = (wc * (NTQ * 16) + l15) * 128 + (((4 + g) ^ fx) << 4);
      } else {
        b_o0 = (8 * g + q) * 256 + pp * 8;
        b_o1 = hk;
      }
    }
    // transposed-read lane addresses (32-bit LDS byte addresses of ring buffer 0, k-step 0, first read)
    uint32_t a_tr[4], b_tr[4];
    if constexpr (A_MODE == 1) {
  #pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        a_tr[mt] = (uint32_t)(size_t)LDS_PTR(char, smem) + a_o0 + (((wr * 4 + mt) ^ a_o1) << 5);
    }
    if constexpr (B_MODE == 1) {
  #pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        b_tr[nt] = (uint32_t)(size_t)LDS_PTR(char, smem) + 2 * HT + b_o0 + (((wc * 4 + nt) ^ b_o1) << 5);
    }
    // ---- retire K tile 0's first half-tiles, preload the first quadrant's fragments ("ovl": the staged
    //      half-tiles were issued between the steps of a fully valid tile's epilogue, whose stores sit behind them)
    const bool ovl = behind_stores;
    WAITBAR2(ovl, CAPW(3 * 4 + 3 * PB, 7), 3 * 4 + 3 * PB);
    LOADA(RA0, 0, 0)
    LOADB(RB0, 0, 0)
    WAITBAR2(ovl, CAPW(W32, 6), W32);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 2 * NTQ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int t = 0;
#pragma clang loop unroll(disable)
    for (; t + 1 < nk; t += 2) {   // two K tiles per iteration, straight-line body
      const bool first = ovl && (t == 0);
      const int t2 = min(t + 2, nk - 1), t3 = min(t + 3, nk - 1);
      HALF_A
      HALF_B
    }
    if (t < nk) {                  // odd contraction length: one more K tile
      const bool first = ovl && (t == 0);
      const int t2 = nk - 1;
      HALF_A_LAST
    }
    WAITBAR(0);   // every wave has read its last fragments: the ring may be refilled
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> accumulator reads (hazard is invisible inside asm)

#if R256_ABLATE & 2   // development: every tile's epilogue reads / writes the first tile's rows and columns (they stay in L2)
    const int em0 = 0, en0 = 0;
#else
    const int em0 = m0, en0 = n0;
#endif
    float* const eout = out_cur;
    const int eldo = ldo_cur, emvalid = mvalid_cur;
    // ---- next work item: own XCD's ticket, else steal; broadcast through LDS
    int* sched_lds = reinterpret_cast<int*>(smem + 2 * BUFB + SCRATCH_BYTES);
    if (!dyn) {
      if (tid == 0) *sched_lds = (w + (int)gridDim.x < nwork) ? w + (int)gridDim.x : -1;
    } else if (tid == 0) {
      // dynamic mode.  The returning atomic is drawn HERE, where wave 0 has nothing in flight (a wait for its
      // result anywhere else would also wait for staging loads or for the previous tile's stores): about one
      // L2 round trip per TICKET_ITEMS tiles on the critical path.
      int wn = -1;
      const bool drew = pend_w < 0;
      if (!drew) {                      // second item of the ticket in hand
        wn = pend_w;
        pend_w = -1;
      } else {
        auto tickets_of = [&](int x) { return (items_of(x) - static_of(x) + TICKET_ITEMS - 1) / TICKET_ITEMS; };
        auto take = [&](int x, unsigned int tk) {   // items x + 8 j, j = static_of(x) + TICKET_ITEMS * tk + {0, 1}
          const int j = static_of(x) + TICKET_ITEMS * (int)tk;
          if (j < items_of(x)) {
            wn = x + 8 * j;
            if (TICKET_ITEMS == 2 && j + 1 < items_of(x)) pend_w = x + 8 * (j + 1);
          }
        };
        // Far from the end of this XCD's list (judged by the last ticket this block drew): draw directly.
        if (last_tk + 2 * static_of(xcd) < tickets_of(xcd)) {
          const unsigned int tk = atomicAdd(tickets + xcd, 1u);
          last_tk = (int)tk;
          take(xcd, tk);
        }
        // Near the end: one batch of plain (device-scope) loads first - every block finds its list exhausted at
        // about the same time, and 256 failing atomics on the same few addresses would serialise in L2.
        for (int attempt = 0; attempt < 4 && wn < 0; ++attempt) {
          unsigned int cnt[8];
#pragma unroll
          for (int y = 0; y < 8; ++y) cnt[y] = __hip_atomic_load(tickets + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int best = -1, best_left = 0;
#pragma unroll
          for (int y = 0; y < 8; ++y) {
            const int left = tickets_of(y) - (int)cnt[y];
            // own list first; otherwise steal from the list with the most tickets left
            if (left > 0 && (y == xcd || (best != xcd && left > best_left))) { best = y; best_left = left; }
          }
          if (best < 0) break;
          const unsigned int tk = atomicAdd(tickets + best, 1u);
          if (best == xcd) last_tk = (int)tk;
          take(best, tk);
        }
      }
      *sched_lds = wn;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int wnext = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(sched_lds));
    const bool more = wnext >= 0;
    const int ez = cur_z, etl = cur_tl;
    if (more) setup(wnext);
    bool staged = false;      // SK: the next item's first K tiles were issued with the slab stores (not by the epilogue's hook)
    if constexpr (SK) {
      if (p.splits > 1) {     // uniform
        constexpr int SLAB = 256 * BNT * 4;                 // one tile of f32 accumulators: [8 x TN registers of 4][256 threads]
        constexpr int TNK = 2 * NTQ;
        char* const slabs = reinterpret_cast<char*>(p.sk_ws) + GEMM_SK_COUNTER_BYTES;
        const uint32_t voff = (uint32_t)lane_id_volatile() * 16u + (uint32_t)wave * 1024u;
        {
          char* const mine = slabs + ((size_t)etl * p.splits + ez) * SLAB;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (more) stage_first(i);
#pragma unroll
            for (int j = 0; j < TNK; ++j)
              asm volatile(R256_SGPR_PAD "global_store_dwordx4 %0, %1, %2 sc1" ::"v"(voff), "a"(acc[i][j]), "s"(mine + (i * TNK + j) * 4096) : "memory");
          }
        }
        // every wave's stores (and the staged tiles) have retired; then ONE lane draws the tile's ticket
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid == 0) {
          unsigned int* cnt = reinterpret_cast<unsigned int*>(p.sk_ws) + etl;
          const unsigned int old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == (unsigned int)(p.splits - 1))            // the last ticket: nobody else touches this counter in this launch
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          sched_lds[1] = (int)old;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int ticket = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(sched_lds + 1));
        staged = more;
        if (ticket != p.splits - 1) {     // another split of this tile comes later and reduces
          if (!more) break;
          behind_stores = false;          // (everything issued so far has retired)
          w = wnext;
          continue;
        }
        // ---- reducer: acc = sum of the slabs in ascending z.  Two splits: registers + the other slab (IEEE addition commutes:
        //      the same bits whichever split reduces); more: all slabs re-read in order, this split's own included
        const bool two = p.splits == 2;
        if (!two) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TNK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int zz = 0; zz < p.splits; ++zz) {
          if (two && zz == ez) continue;
          const char* const src = slabs + ((size_t)etl * p.splits + zz) * SLAB;
          f32x4 q[2][TNK];
#pragma unroll
          for (int j = 0; j < TNK; ++j)
            asm volatile(R256_SGPR_PAD "global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(q[0][j]) : "v"(voff), "s"(src + j * 4096) : "memory");
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i + 1 < 8) {
#pragma unroll
              for (int j = 0; j < TNK; ++j)
                asm volatile(R256_SGPR_PAD "global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(q[(i + 1) & 1][j]) : "v"(voff),
                             "s"(src + ((i + 1) * TNK + j) * 4096) : "memory");
              asm volatile("s_waitcnt vmcnt(%0)" ::"i"(TNK) : "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < TNK; ++j) {
              asm volatile("" : "+v"(q[i & 1][j]));
              acc[i][j] += q[i & 1][j];
            }
          }
        }
      }
    }
    if constexpr (EPI == EPI_F32_ATOMIC) {
      GemmParams pe = p;
      pe.out = eout;   // (the tile being written belongs to the previous work item's problem)
      pe.ldo = eldo; pe.m_valid = emvalid;
      gemm_epilogue<8, TN, EPI, 2, 2, true>(acc, pe, smem + 2 * BUFB, em0, en0, wr * 128, wc * (NTQ * 32), wave, lane,
                                            [&](int step) { if (more) stage_first(step); }, more ? 4 : 0);
    } else {
      gemm_epilogue<8, TN, EPI, 2, 2, true>(acc, p, smem + 2 * BUFB, em0, en0, wr * 128, wc * (NTQ * 32), wave, lane,
                                            [&](int step) { if (more && !staged) stage_first(step); }, (more && !staged) ? 4 : 0);
    }
    if (!more) break;
    behind_stores = (em0 + 256 <= emvalid);
    w = wnext;
  }
#if R256_STAMP
  if (threadIdx.x == 0) {
    g_r256_stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
    g_r256_stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
#undef HALF_A
#undef HALF_A_LAST
#undef NOLOAD_UH
#undef HALF_B
#undef PHASE
#undef PHASE_BODY
#undef LOADA
#undef LOADB
#undef LOADA_U
#undef LOADB_U
#undef LOAD_TR
#undef LOAD_TR_H
#undef LOADA_UH
#undef LOADB_UH
#undef MMA1
#undef GA
#undef CAPW
}

// Host side of the dynamic scheduler, PER DEVICE (the counters are a `__device__` global: one copy per device): the
// launch parity - ONE sequence for every instantiation of the kernel - and the stream the dynamic launches of that
// device are serialised on.  Launch k clears the counter set launch k + 1 draws from, which is only sound when the
// launches of a device run one after the other: dynamic hand-out is therefore granted to ONE stream per device (the
// first that asks); a launch that asks for it on another stream gets the static walk (correct, no shared state).
struct RingSched {
  unsigned seq = 0;
  hipStream_t stream = nullptr;
  bool bound = false;
};
inline RingSched& ring_sched(int dev) {
  static RingSched s[64];
  return s[(dev >= 0 && dev < 64) ? dev : 0];
}

template <int A_MODE, int B_MODE, int EPI, int NTQ, bool SK>
int launch256_k(const GemmParams& p, hipStream_t st) {
  constexpr bool DYN = (R256_DYN != 0);
  constexpr int BNT = NTQ * 64;
  if ((p.M & 255) || (p.N % BNT) || (p.K & 63)) return VAULT_EINVAL;
  constexpr int LDS = 2 * BUFB + 4 * 16 * ((NTQ == 3 ? 96 : 64) + 4) * 4 + 16;   // ring + epilogue scratch (gemm_epi.h: 16 x LD floats per wave) + scheduler word
  auto kern = gemm256_kernel<A_MODE, B_MODE, EPI, NTQ, DYN, SK>;
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VAULT_EINVAL;
  if (!attr_done[dev]) {   // (function attributes are per device)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_done[dev] = true;
  }
  GemmParams q = p;
  if (p.a_hm) {     // head-major A: a_mode 0 only, planes of >= M rows, no batching
    if (A_MODE != 0 || p.a_hm < p.M || p.batch > 1) return VAULT_EINVAL;
    q.lda = 64;
  }
  if (p.out_hm) return VAULT_EINVAL;      // (head-major output: the 8-wave kernel's EPI_BF16 only)
  q.gn = (p.gn > 0) ? std::min(p.gn, p.N / BNT) : p.N / BNT;   // default: plain m-major raster (see gemm.hip)
  const int nk_total = p.K >> 6;
  const int per = (nk_total + p.splits - 1) / p.splits;
  q.splits = (nk_total + per - 1) / per;                        // every split owns at least one K tile
  if constexpr (SK) {   // split-K items with the in-launch reduction: the caller's counters + slabs must cover (tile, split)
    const long long tiles = (long long)(p.M >> 8) * (p.N / BNT);
    if (q.splits > 1 && (p.sk_ws == nullptr || p.batch > 1 || tiles > GEMM_SK_MAX_TILES ||
                         p.sk_bytes < GEMM_SK_COUNTER_BYTES + tiles * q.splits * (256LL * BNT * 4)))
      return VAULT_EINVAL;
  }
  const int nwork = (p.M >> 8) * (p.N / BNT) * q.splits * ((EPI == EPI_F32_ATOMIC && p.batch > 1) ? p.batch : 1);
  dim3 grid(std::min(nwork, 256), 1, 1);
  int persist = p.persist & 0xff;
  int parity = 0;
  if (!DYN) persist &= ~1;
  if (persist & 1) {
    RingSched& rs = ring_sched(dev);
    if (!rs.bound) { rs.bound = true; rs.stream = st; }
    if (rs.stream == st) parity = (int)(rs.seq++ & 1u);   // the sequence counts this device's dynamic launches only
    else persist &= ~1;                                   // another stream of this device: static walk
  }
  q.persist = persist | (parity << 8);
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, st, q);
  return (int)hipGetLastError();
}

template <int A_MODE, int B_MODE, int EPI, int NTQ>
int launch256(const GemmParams& p, hipStream_t st) {
  // split-K of a 16-bit / f32-residual epilogue: the SK instantiations (kernel header), where they exist
  // (192-wide tiles: N = 768; the 256-wide form's reducer does not fit the register file beside 256 accumulators)
  constexpr bool SK_OK = A_MODE == 0 && NTQ == 3 && ((B_MODE == 1 && EPI == EPI_BF16) || (B_MODE == 0 && EPI == EPI_F32_RES));
  if (EPI != EPI_F32_ATOMIC && p.splits > 1) {
    if constexpr (SK_OK) return launch256_k<A_MODE, B_MODE, EPI, NTQ, true>(p, st);
    else return VAULT_EINVAL;
  }
  return launch256_k<A_MODE, B_MODE, EPI, NTQ, false>(p, st);
}

template <int NTQ>
int dispatch256(const GemmParams& p, int a_mode, int b_mode, int epi, hipStream_t st) {
  const int key = a_mode * 2 + b_mode;
#define VAULT_DISPATCH(AM, BMD)                                                       \
  switch (epi) {                                                                      \
    case EPI_BF16: return launch256<AM, BMD, EPI_BF16, NTQ>(p, st);                   \
    case EPI_BF16_GELU:                                                               \
      if (p.out2 != nullptr) return launch256<AM, BMD, EPI_BF16_GELU, NTQ>(p, st);    \
      else return launch256<AM, BMD, EPI_BF16_GELU_INF, NTQ>(p, st);                  \
    case EPI_BF16_DGELU: return launch256<AM, BMD, EPI_BF16_DGELU, NTQ>(p, st);       \
    case EPI_F32_RES: return launch256<AM, BMD, EPI_F32_RES, NTQ>(p, st);             \
    case EPI_F32_PATCH:   /* (the patch embedding is a forward-form GEMM) */          \
      if constexpr (BMD == 0) return launch256<AM, BMD, EPI_F32_PATCH, NTQ>(p, st);   \
      else return VAULT_EINVAL;                                                       \
    case EPI_F32_ATOMIC:                                                              \
      if constexpr (NTQ == 4) return launch256<AM, BMD, EPI_F32_ATOMIC, NTQ>(p, st);  \
      else return VAULT_EINVAL;                                                       \
    default: return VAULT_EINVAL;                                                     \
  }
  switch (key) {
    case 0: VAULT_DISPATCH(0, 0)
    case 1: VAULT_DISPATCH(0, 1)
    case 3:   // (both operands K-major: the weight gradients only - the other epilogues of this form spill and nothing uses them)
      if constexpr (NTQ == 4) {
        if (epi == EPI_F32_ATOMIC) return launch256<1, 1, EPI_F32_ATOMIC, NTQ>(p, st);
      }
      return VAULT_EINVAL;
    default: return VAULT_EINVAL;
  }
#undef VAULT_DISPATCH
}

}  // namespace

#if R256_DYN
#define R256_PUBLIC(NAME) NAME##_dyn
#else
#define R256_PUBLIC(NAME) NAME
int vault_gemm256_grouped_launch_dyn(const GemmParams& p, hipStream_t st);
int vault_gemm256_launch_dyn(const GemmParams& p, int a_mode, int b_mode, int epi, int ntq, hipStream_t st);
#endif

// Grouped weight gradients (GemmParams::seg): validated here, launched on the (1,1) atomic-epilogue instantiation.
int R256_PUBLIC(vault_gemm256_grouped_launch)(const GemmParams& p, hipStream_t st) {
#if !R256_DYN
  if (p.persist & 1) return vault_gemm256_grouped_launch_dyn(p, st);   // dynamic hand-out: the other translation unit's kernels
#endif
  if (p.nseg < 1 || p.nseg > 3 || (p.K & 63) || p.splits < 1) return VAULT_EINVAL;
  GemmParams q = p;
  int total = 0;
  for (int k = 0; k < p.nseg; ++k) {
    const GemmParams::Seg& g = p.seg[k];
    if (!g.A || !g.B || !g.out || g.tiles_n < 1 || g.tiles < g.tiles_n || g.tiles % g.tiles_n || g.count < 1 || g.first < 0 ||
        (g.lda & 7) || (g.ldb & 7) || (g.ldo & 3) || (g.batch_a & 7) || (g.batch_b & 7) || (g.batch_o & 3) ||
        (g.a_hm == 0 && g.lda < (g.tiles / g.tiles_n) * 256) || (g.a_hm != 0 && g.lda != 64) || g.ldb < g.tiles_n * 256 ||
        g.ldo < g.tiles_n * 256)
      return VAULT_EINVAL;
    total += g.count;
  }
  for (int k = p.nseg; k < 3; ++k) q.seg[k] = GemmParams::Seg{};
  // plain-launch fields the kernel still reads
  q.A = p.seg[0].A; q.B = p.seg[0].B; q.out = p.seg[0].out;
  q.M = 256; q.N = 256; q.lda = p.seg[0].lda; q.ldb = p.seg[0].ldb; q.ldo = p.seg[0].ldo; q.m_valid = 256;
  q.batch = 0; q.gn = 1;
  constexpr int LDS = 2 * BUFB + 4 * 16 * (64 + 4) * 4 + 16;
  auto kern = gemm256_kernel<1, 1, EPI_F32_ATOMIC, 4, (R256_DYN != 0)>;
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return VAULT_EINVAL;
  if (!attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_done[dev] = true;
  }
  const int nk_total = p.K >> 6;
  const int per = (nk_total + p.splits - 1) / p.splits;
  q.splits = (nk_total + per - 1) / per;
  const int nwork = total * q.splits;
  int persist = p.persist & 0xff, parity = 0;
  if (!R256_DYN) persist &= ~1;
  if (persist & 1) {
    RingSched& rs = ring_sched(dev);
    if (!rs.bound) { rs.bound = true; rs.stream = st; }
    if (rs.stream == st) parity = (int)(rs.seq++ & 1u);
    else persist &= ~1;
  }
  q.persist = persist | (parity << 8);
  hipLaunchKernelGGL(kern, dim3(std::min(nwork, 256)), dim3(256), LDS, st, q);
  return (int)hipGetLastError();
}

int R256_PUBLIC(vault_gemm256_launch)(const GemmParams& p, int a_mode, int b_mode, int epi, int ntq, hipStream_t st) {
#if !R256_DYN
  if (p.persist & 1) return vault_gemm256_launch_dyn(p, a_mode, b_mode, epi, ntq, st);
#endif
  if (ntq == 2) {
    // 256 x 128 tiles (cfg 8): the N = 768 Linears of a 40-row-tile problem (the LM stack at batch 256) are 160 tiles at
    // 192 columns - 62 % of the CUs, one round - and 240 at 128; residual forward and (0,1) data gradient only
    if (a_mode == 0 && b_mode == 0 && epi == EPI_F32_RES) return launch256<0, 0, EPI_F32_RES, 2>(p, st);
    if (a_mode == 0 && b_mode == 1 && epi == EPI_BF16) return launch256<0, 1, EPI_BF16, 2>(p, st);
    return VAULT_EINVAL;
  }
  return ntq == 3 ? dispatch256<3>(p, a_mode, b_mode, epi, st) : dispatch256<4>(p, a_mode, b_mode, epi, st);
}
