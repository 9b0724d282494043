/* C ABI of libvault_hip.so: the MI355X (gfx950) implementation of VAuLT's stacked BERT -> ViLT
 * forward/backward hot path.
 *
 * The reference (gchochla/VAuLT) is pure Python on HuggingFace/ATen and has no FFI of its own
 * (SURVEY.md §8 b-2); each entry point below names the reference arithmetic it replaces.
 * Conventions:
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the library allocates
 *     nothing and keeps no state except lazily initialised kernel attributes;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it, never synchronised;
 *   - return 0 on success, 22 (EINVAL) on a shape/alignment violation detected before launch,
 *     otherwise a hipError_t value;
 *   - "f32" = IEEE binary32; row-major everywhere;
 *   - "bf16" in a name or comment = the library's 16-bit OPERAND type.  The same sources are built twice
 *     (vault_amd/build.py): libvault_hip.so computes on bf16 operands (v_mfma_f32_16x16x32_bf16),
 *     libvault_hip_f16.so on IEEE fp16 operands (v_mfma_f32_16x16x32_f16, same matrix rate; f32 -> f16 conversions
 *     saturate at +-65504 instead of overflowing to infinity).  Both export this one ABI, name for name;
 *     vault_operand_format() says which build a loaded library is.  A caller that wants the reference's fp32
 *     logits / loss within 1e-3 through a 24-layer stack binds the fp16 build and multiplies the loss gradient by a
 *     power of two (vault_head_loss_args.grad_scale) that the optimizer divides out again (vault_adamw_step
 *     grad_scale); bf16 needs no such scale and is 4e-3 from the reference (ABI 8).
 */
#ifndef VAULT_HIP_H
#define VAULT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int vault_abi_version(void);
/* 0 = bf16 operands (libvault_hip.so), 1 = IEEE fp16 operands (libvault_hip_f16.so).  ABI 8. */
int vault_operand_format(void);

/* ---- GEMM -------------------------------------------------------------------------------
 * C[M,N] = A.B with bf16 operands / f32 accumulation and a fused epilogue.  Replaces every
 * nn.Linear forward/backward of HF:models/vilt/modeling_vilt.py:303-414 and
 * HF:models/roberta/modeling_roberta.py:222-398 (called from ref: vault/models/vault/model.py:190,205)
 * and the Conv2d patch projection (modeling_vilt.py:290-300).
 * a_mode 0: A is [M][K]; 1: A is [K][M].   b_mode 0: B is [N][K]; 1: B is [K][N].
 * epi: 0 bf16 out (+bias) | 1 bf16 gelu(acc+bias) (+ out2 = gelu'(acc+bias)) | 2 bf16 acc*aux (aux = that gelu')
 *      3 f32 out = dropout(acc+bias)+res | 4 f32 patch rows (row remap + addtab) | 5 f32 out += acc.
 * M % 128 == 0, N % 128 == 0, K % 64 == 0; buffers must be allocated to those padded sizes.
 * Rows >= m_valid are not stored.  cfg < 0 selects the kernel/tile automatically; explicit values: 0 / 1 / 2 =
 * double-buffered kernel with 128x128 / 256x128 / 256x256 tiles, 3 / 4 = persistent ring kernel with
 * 256x256 / 256x192 tiles (M % 256 == 0, N % 256 / 192 == 0; its residual epilogue needs `res`, without one
 * the launcher takes the double-buffered kernel), 5 / 6 (ABI 4) = 8-wave kernel with 256x256 / 256x192 tiles whose
 * epilogue stores the accumulators straight from registers (a_mode = b_mode = 0 only, K >= 128, no split-K, no split3,
 * no dropout, epi 0 / 1 / 2 / 3; M x ldo x 4 B < 4 GiB): the automatic choice for the bf16-output Linears with K <= 1024;
 * 7 = 64x128 tiles, four stages (a_mode 0, epi 0..4, no split-K): the automatic choice while (M/64) x (N/128) <= 256 blocks;
 * 8 (ABI 8) = ring kernel with 256x128 tiles (a_mode 0; epi 3 with `res` and b_mode 0, or epi 0 with b_mode 1; no split-K):
 * the automatic choice where 256x192 tiles would be one partial round that 256x128 tiles still cover in one (N = 768 at
 * 32..42 row tiles of 256).
 * Threading: one host thread per device; the ring kernel's dynamic scheduler (persist bit 0) keeps per-device ticket
 * counters that assume its launches are serialised on ONE stream per device. */
typedef struct vault_gemm_args {
  const void* A; const void* B; void* out; void* out2;
  const float* bias; const float* res; const void* aux; const float* addtab;
  float* colsum;   /* optional, bf16 epilogues: += column sums of the output over rows < m_valid */
  int split3;      /* bf16 epilogues: store [hi | lo | hi] per row (ldo = 3N): A operand of a split-bf16 GEMM */
  int M, N, K, lda, ldb, ldo, m_valid;
  int a_mode, b_mode, epi, cfg, splits, accumulate;
  int rpg, gstride, goff;
  uint32_t drop_thresh, drop_seed, drop_stream; float drop_scale;
  int gn;   /* tuning: n-tiles per raster group (0 = default, plain m-major raster) */
  int persist; /* scheduling: bit 0 = ring kernel hands tiles out dynamically (per-XCD ticket counters), bit 1 = the
                  double-buffered kernel launches one block per tile instead of its persistent grid; 3 when the GEMMs
                  share the GPU with another kernel (RCCL collectives of a data-parallel step), 0 otherwise */
  int batch;   /* ABI 3, EPI_F32_ATOMIC only (cfg 0..3; default: the 256x256 ring kernel when M and N are multiples of 256,
                  else 128x128): `batch` > 1 weight gradients of one shape in one launch, problem b at A + b * batch_a,
                  B + b * batch_b (bf16 elements) and out + b * batch_o (floats).  The layers of an encoder stack share
                  their shapes: contracted in one launch they fill the GPU with few or no split-K partial sums, and
                  the ring kernel's layer-major work list keeps a layer's operand panels in one XCD's L2.
                  0 / 1 = a single GEMM */
  long long batch_a, batch_b, batch_o;
  int aux_u8;  /* ABI 5, kernels 5 / 6 only: the gelu' tensor (out2 of epi 1, aux of epi 2) is 8-bit in tile-native order - an
                  opaque M x N x (4/3 at 192-wide tiles) byte image that only the epi-2 GEMM of the SAME M, N and kernel (cfg)
                  can read back: q = rne(200 g + 26), step 0.005 over -0.13 .. 1.145 (0, 0.5, 1 exact).  Halves the bytes the
                  FFN-in forward writes for backward and the gelu'-product dgrad reads.  Ask vault_gemm_plan which kernel a
                  call would take and pass that cfg explicitly to both calls. */
  int out_hm;  /* ABI 8, epi 0 on the 8-wave kernel (cfg 5 / 6) only: R > 0 = write the output HEAD-MAJOR, [N / 64][R][64] instead
                  of [M][ldo] (R >= M rows per plane): the layout vault_attention_* reads with qkv_hm = R */
  int a_hm;    /* ABI 8, ring kernel (cfg 3 / 4 / 8), a_mode 0: R > 0 = A is head-major [K / 64][R][64] (lda is ignored): the QKV
                  data gradient reading the attention backward's dqkv */
  void* out_q; void* out_scale;   /* ABI 10, vault_gemm_mxfp8 on kernel 5 with epi 1 only (both or neither): also write the
                  MXFP8 image of the 16-bit output - e4m3 [M][N] + E8M0 [M][N / 32], byte for byte what vault_quant_mxfp8 makes
                  of `out` - the A operand of the next vault_gemm_mxfp8 (FFN-in's GELU output feeding FFN-out) */
  void* splitk_ws; long long splitk_bytes;   /* ABI 11, optional: workspace for SPLIT-K with the reduction inside the launch - a_mode 0,
                  epi 0 (b_mode 1) or epi 3 (b_mode 0), N % 192 == 0, on the ring kernel's 192-wide tiles (cfg 4): every K split of a
                  tile stores its f32 accumulators into a slab, the split that arrives last adds them up in split order and runs the
                  epilogue (bit-reproducible).  Layout: 16 KiB of tile counters - ZERO before the first call, left zero by every
                  call - then tiles x splits slabs of 256 x 192 floats.  With a workspace, cfg < 0 and splits <= 1 the library
                  splits where that fills the chip (few row tiles, long contractions: vault_gemm_plan reports cfg 4 either way);
                  splits > 1 with cfg 4 forces the count.  One workspace must not serve launches that can run concurrently. */
} vault_gemm_args;
int vault_gemm(const vault_gemm_args* args, void* stream);
/* the kernel / tile configuration vault_gemm would run these arguments on (the resolved `cfg`, 0..8), or -EINVAL */
int vault_gemm_plan(const vault_gemm_args* args);

/* ---- grouped weight gradients (ABI 8) ---------------------------------------------------------------------------
 * dW_k,l[n_out x n_in] (+)= dY_k,l[tokens x n_out]^T . X_k,l[tokens x n_in] for up to three KINDS k of Linear (FFN-out,
 * FFN-in, attention-out, QKV: autograd of HF:models/vilt/modeling_vilt.py:303-414 under ref: vault/tmsc_utils/trainer.py:365)
 * and the layers l of a stack in ONE launch.  Every 256 x 256 output tile costs the same whatever its kind (the contraction
 * runs over the tokens), so a launch is sized by ITEMS, not by kinds: a segment names a run [first, first + count) of its
 * kind's tiles in (layer-major, tile-minor) order, and e.g. the 216 FFN-out tiles of six layers + 40 attention-out tiles
 * fill the 256 CUs exactly once, where one launch per kind fills 84 % of them.  n_out, n_in multiples of 256, tokens a
 * multiple of 64 (zero rows beyond the valid ones contribute nothing); `splits` > 1 cuts the contraction of EVERY item
 * (partial sums by float atomics); accumulate = 0 with splits = 1 stores instead of adding (dW known to be zero / dead).
 * layer l of a kind: dy + l * batch_dy, x + l * batch_x (16-bit elements), dw + l * batch_dw (floats). */
typedef struct vault_wgrad_seg {
  const void* dy; const void* x; float* dw;
  int n_out, n_in, ld_dy, ld_x, ld_dw;
  int batch;       /* layers of this kind */
  int first, count;
  long long batch_dy, batch_x, batch_dw;
  int dy_hm;       /* R > 0: dy is head-major [n_out / 64][R][64] (ld_dy ignored): the QKV kind reading dqkv */
} vault_wgrad_seg;
typedef struct vault_wgrad_grouped_args {
  int nseg;
  vault_wgrad_seg seg[3];
  int tokens, splits, accumulate;
  int persist;     /* as vault_gemm_args.persist */
} vault_wgrad_grouped_args;
int vault_wgrad_grouped(const vault_wgrad_grouped_args* args, void* stream);

/* ---- MXFP8 forward GEMM (BASELINE config "fp8 MFMA forward, bf16 backward") -------------------------------------
 * OCP microscaling format: e4m3 elements [rows][K] (K contiguous) + one E8M0 scale byte per 32 consecutive k,
 * [rows][K/32].  vault_quant_mxfp8 converts a bf16 operand (activations of the forward pass, bf16 weight shadow):
 * scale exponent = floor(log2(max |x| of the block)) - 8, elements rounded to nearest even, saturating at +-448.
 * vault_gemm_mxfp8: out = epilogue(A . B^T) with A = [M][K], B = [N][K] in that format (args->A / args->B point to
 * the element bytes, lda = ldb = K, a_mode = b_mode = 0, splits <= 1), fp32 accumulation in
 * v_mfma_scale_f32_16x16x128_f8f6f4, epilogues EPI_BF16, EPI_BF16_GELU, EPI_F32_RES as vault_gemm.
 * M % 256 == 0, N % 256 == 0, K % 128 == 0.  Same Linear layers as vault_gemm (forward only).
 * Kernels (args->cfg; ABI 10): 0 = the simple double-buffered kernel (csrc/gemm_mx8.hip); 5 / 6 = the 8-wave kernel of
 * vault_gemm on MXFP8 operands, 256- / 192-wide tiles (csrc/gemm8w.hip, MX: K >= 384, no dropout in the residual epilogue;
 * N % 192 == 0 suffices for 6) - the only ones that take aux_u8 (epi 1: the gelu' image the epi-2 vault_gemm of the same cfg
 * reads back) and out_hm; -1 = 5 / 6 where they take the call, else 0.  The three give bit-identical accumulators (one scaled
 * MFMA per 128 k, same order).  vault_gemm_mxfp8_plan: the kernel a call would run on (0, 5, 6), or -EINVAL. */
int vault_quant_mxfp8(const void* src_bf16, long long rows, int K, int ld_src, void* dst_q, void* dst_scale, void* stream);
int vault_gemm_mxfp8(const vault_gemm_args* args, const void* a_scale, const void* b_scale, void* stream);
int vault_gemm_mxfp8_plan(const vault_gemm_args* args);


/* ---- LayerNorm ----------------------------------------------------------------------------
 * fp32 statistics, one wave per row, H in {256, 512, 768, 1024, 1536}.  Replaces nn.LayerNorm at
 * HF:models/vilt/modeling_vilt.py:431-447,637, HF:models/roberta/modeling_roberta.py:339,397 and
 * the embedding LayerNorms (modeling_vilt.py:267, modeling_roberta.py:119).
 * Row maps: logical row r lives at physical row (r / rpg) * gstride + goff + r % rpg (rpg == 0:
 * identity) - used to read/write the text or CLS rows of the fused [text | patch] sequence.
 * y = dropout(LN(x)) + post_add ; outputs y_f32 and/or y_bf16 ; mean/rstd indexed by logical row. */
typedef struct vault_ln_fwd_args {
  const float* x; const float* gamma; const float* beta; const float* post_add;
  void* y_bf16; float* y_f32; float* mean; float* rstd;
  int rows, H; float eps;
  int x_rpg, x_gstride, x_goff, y_rpg, y_gstride, y_goff;
  uint32_t drop_thresh, drop_seed, drop_stream; float drop_scale;
  void* y_split3;   /* optional bf16 [rows][3H] = [hi | lo | hi]: A operand of a split-bf16 (precise) GEMM */
  void* y_q; void* y_scale;   /* optional (ABI 2): MXFP8 image of y_bf16, e4m3 [rows][H] + E8M0 [rows][H/32] - the bytes
                                 vault_quant_mxfp8 gives for y_bf16: A operand of vault_gemm_mxfp8 */
} vault_ln_fwd_args;
int vault_layernorm_fwd(const vault_ln_fwd_args* args, void* stream);

/* dy = dy_bf16 + dy_f32 (either may be NULL); dx_f32 = LNbwd(dy) + dres ; dx_bf16 = bf16(dx_f32)
 * (optionally dropout-masked); dgamma/dbeta are accumulated (+=) with float atomics. */
typedef struct vault_ln_bwd_args {
  const void* dy_bf16; const float* dy_f32; const float* x; const float* mean; const float* rstd;
  const float* gamma; const float* dres;
  float* dx_f32; void* dx_bf16; float* dgamma; float* dbeta;
  int rows, H;
  int dy_rpg, dy_gstride, dy_goff, x_rpg, x_gstride, x_goff, dx_rpg, dx_gstride, dx_goff;
  uint32_t drop_thresh, drop_seed, drop_stream; float drop_scale;
  int drop_on_dy;   /* 0: mask dx_bf16 (Linear -> dropout -> +res -> LN) ; 1: mask dy (LN -> dropout) */
  float* dbias;     /* optional: += column sums of dx_bf16 (bias gradient of the Linear fed by that branch) */
  const void* dres_bf16;   /* ABI 5, optional: a residual gradient in bf16, added like dres (row map: dx's).  With it and
                              dx_f32 = NULL the residual-gradient stream of a pre-LN stack lives in bf16 only: dx_bf16 is
                              both the stream and the dY operand of the Linear below */
} vault_ln_bwd_args;
int vault_layernorm_bwd(const vault_ln_bwd_args* args, void* stream);

/* out[n] += sum_{r < rows} in_bf16[r][n]   (bias gradients); N % 256 == 0 */
int vault_colsum(const void* in_bf16, int ld, int rows, int N, float* out, void* stream);
/* ABI 7: the same for `batch` matrices at element stride batch_in, into `batch` vectors at float stride batch_out (the QKV bias
 * gradients of a group of layers in one launch) */
/* ABI 8: the same column sums over a HEAD-MAJOR tensor [planes][hm_rows][64] (vault_attn_args.qkv_hm): out[64 p + d] += sum over
 * rows < rows of in[(p * hm_rows + r) * 64 + d] for the first `planes` planes - the query third of the QKV bias gradient. */
int vault_colsum_hm(const void* in_bf16, int rows, int hm_rows, int planes, float* out, int batch, long long batch_in,
                    long long batch_out, void* stream);
int vault_colsum_batched(const void* in_bf16, int ld, int rows, int N, float* out, int batch, long long batch_in,
                         long long batch_out, void* stream);


/* ---- attention ------------------------------------------------------------------------------
 * softmax(Q K^T / 8 + keymask) V per (batch, head), head dim 64, S <= 192 keys.  qkv is the packed
 * [B*S][3H] bf16 output of the fused QKV GEMM (q | k | v, head h at columns h*64 of each part),
 * ctx/dctx are [B*S][H] bf16, lse is [B][heads][S] f32 (natural-log-sum-exp of the scaled scores,
 * written by fwd, read by bwd), keymask [B][S] f32 (1 keep / 0 masked) or NULL, dqkv like qkv.
 * Replaces HF:models/vilt/modeling_vilt.py:322-351 and HF:models/roberta/modeling_roberta.py:158-250
 * (+ autograd).  drop_*: attention-probability dropout (LM in train mode), thresh 0 = off. */
typedef struct vault_attn_args {
  const void* qkv; const float* keymask; void* ctx; float* lse;
  const void* dctx; void* dqkv;
  int B, S, H, heads;
  uint32_t drop_thresh, drop_seed, drop_stream; float drop_scale;
  void* ctx_split3; /* fwd, optional bf16 [B*S][3H] = [hi | lo | hi] instead of ctx (precise path) */
  int qkv_hm;       /* ABI 8: 0 = qkv / dqkv are [B*S][3H] (q | k | v, head h at columns 64 h); R > 0 = HEAD-MAJOR
                       [3][heads][R][64] with R >= B*S padded token rows: the S rows of a (batch, head) item are contiguous,
                       so the attention kernels' loads and - above all - the backward's dq / dk / dv stores stream instead of
                       touching 128-byte segments at a 6 H byte stride.  Written by vault_gemm with out_hm, read by
                       vault_gemm a_hm / vault_wgrad_seg.dy_hm / vault_colsum_hm.  S <= 192 in backward, no ctx_split3. */
  float* bias_partials; /* ABI 9, bwd, optional: the QKV bias gradient (column sums of dq | dk | dv over the tokens: autograd of the
                       q / k / v nn.Linear biases, HF modeling_vilt.py:303-320 / modeling_roberta.py:222-224) without a pass over
                       dqkv - every workgroup of the launch writes the sums over ITS (batch, head) items to row blockIdx of this
                       [vault_attention_bwd_partials(args)][bias_thirds * H] f32 matrix (plain stores, every row written whole);
                       vault_colsum_partials adds the rows up.  S <= 192. */
  int bias_thirds;   /* 1 = the query third only (H sums; the key third is analytically zero and, without attention dropout, the
                       value third is the column sum of dctx: the caller's shortcut), 3 = q | k | v (S <= 64 only) */
} vault_attn_args;
int vault_attention_fwd(const vault_attn_args* args, void* stream);
int vault_attention_bwd(const vault_attn_args* args, void* stream);
int vault_attention_bwd_partials(const vault_attn_args* args);   /* rows of bias_partials that launch writes; 0 = unsupported shape */
/* out[b][c] += sum over p < nparts of part[b][p][c], c < n  (b < batch; element strides batch_in / batch_out) */
int vault_colsum_partials(const float* part, int nparts, int n, float* out, int batch, long long batch_in,
                          long long batch_out, void* stream);


/* ---- embeddings -----------------------------------------------------------------------------
 * vault_position_ids: mode 1 = RoBERTa/BERTweet ids (cumsum(ids != pad) * (ids != pad) + pad,
 * HF:models/roberta/modeling_roberta.py:142-155), mode 0 = arange (BERT). T <= 64. */
int vault_position_ids(const int64_t* ids, int* pos, int B, int T, int mode, int pad, void* stream);

/* gather: out[r] = (src ? src[r] : 0) + sum_k tab[k][index_k(r)] over up to three f32 tables of row
 * width H (H % 256 == 0); index_k(r) = idx[k][r] (int64 if is64[k] else int32) or, when idx[k] is NULL,
 * fixed[k] (>= 0) or r % period (fixed[k] == -2); tab[k] == NULL skips the table.
 * Replaces the embedding sums of modeling_roberta.py:75-121 / modeling_bert.py:69-107 /
 * modeling_vilt.py:237-269.  scatter (backward): tab[k][index_k(r)] += src[r] (float atomics). */
typedef struct vault_gather_args {
  const float* src; float* out;
  const float* tab[3]; const void* idx[3]; int is64[3]; int fixed[3];
  int period, rows, H;
  const float* rowmask;   /* scatter only, optional [rows]: rows with mask 0 are skipped (their gradient is 0) */
} vault_gather_args;
int vault_gather_sum(const vault_gather_args* args, void* stream);
int vault_scatter_add(const vault_gather_args* args, void* stream);

/* pixel_values [B][C][IMG][IMG] f32 -> patch matrix [B*(IMG/ps)^2][C*ps*ps] bf16 (k = c*ps*ps + py*ps
 * + px, row-major patch order): the unfold half of the Conv2d at modeling_vilt.py:290-300. ps % 8 == 0. */
int vault_im2col(const float* pixel_values, void* out_bf16, int B, int C, int IMG, int ps, int split3, void* stream);

/* addtab[p] = conv_bias + pos_emb[1+p] + modality_type[1] (p < P), and the CLS rows
 * x[b*S + T] = cls_token + pos_emb[0] + modality_type[1]  (modeling_vilt.py:160-166,204-215). */
int vault_image_consts(const float* conv_bias, const float* pos_emb, const float* mtype1, const float* cls,
                       float* addtab, float* x, int P, int H, int B, int S, int T, void* stream);
/* backward over the image rows of dx [B*S][H]: dpos, dmtype1, dcls, dconv_bias (+=) and the compact
 * bf16 copy dyp [B*P][H] of the patch-row gradients (operand of the projection wgrad). */
int vault_image_rows_bwd(const float* dx, float* dpos, float* dmtype1, float* dcls, float* dconv_bias,
                         void* dyp_bf16, int P, int H, int B, int S, int T, void* stream);

/* ---- padded batches of differently sized images (pixel_mask != 1 and/or a canvas that is not the pre-training
 * grid): the device half of ViltEmbeddings.visual_embed, HF modeling_vilt.py:92-178, reached from the reference
 * through VaultMixin.vilt_forward (ref: vault/models/vault/model.py:204-205) with the per-item processor
 * output of ref: vault/models/vault/dataset.py:323-347.  The host picks the patch slots that enter the sequence
 * (vault_amd.spec.select_patches: the reference draws the order / the masked padding at random, here it is
 * deterministic): sel int32 [B][L] = slot (row * gw + col) on the (HP/ps) x (WP/ps) grid of the canvas,
 * hw int32 [B][2] = patch rows / cols of each image; G x G = grid of the position table (image_size / ps). */
/* unfold the selected patches: A [B*L][C*ps*ps] bf16 (split3: [hi | lo | hi], 3x wide) */
int vault_im2col_sel(const float* pixel_values, void* out_bf16, const int* sel, int B, int L, int C, int HP, int WP,
                     int ps, int split3, void* stream);
/* addtab[l] = conv_bias + modality_type[1] for all L rows; CLS rows x[b*S + T] = cls + pos_emb[0] + modality_type[1] */
int vault_image_sel_consts(const float* conv_bias, const float* pos_emb, const float* mtype1, const float* cls,
                           float* addtab, float* x, int L, int H, int B, int S, int T, void* stream);
/* x[b*S + T + 1 + l] += bilinear_resize(pos_emb[1:], to hw[b], align_corners)[sel[b][l]]  (0 outside the image) */
int vault_image_pos_sel_fwd(float* x, const float* pos_emb, const int* sel, const int* hw, int B, int L, int S, int T,
                            int H, int gw, int G, void* stream);
/* backward of the three above over dx [B*S][H]: dpos (transposed interpolation), dmtype1, dcls, dconv_bias (+=),
 * dyp [B*L][H] bf16 = patch-row gradients (operand of the projection wgrad) */
int vault_image_sel_bwd(const float* dx, float* dpos, float* dmtype1, float* dcls, float* dconv_bias, void* dyp_bf16,
                        const int* sel, const int* hw, int B, int L, int S, int T, int H, int gw, int G, void* stream);
int vault_axpy_f32(float* dst, const float* src, float a, long long n, void* stream);
/* ABI 8: x *= a over n floats (any n, x 4-byte aligned).  The fp16 build's caller multiplies the loss gradient by a
 * power of two so that the 16-bit data gradients stay inside fp16's exponent range; the flat f32 gradient buffer then
 * holds scaled sums, and this pass (exact for a power of two) removes the scale where no fused optimizer does it
 * (p.grad of the autograd bridge; ref: loss.backward() at vault/tmsc_utils/trainer.py:365). */
int vault_scale_f32(float* x, float a, long long n, void* stream);
/* ABI 4: externally supplied image embeddings (HF ViltEmbeddings.forward with `image_embeds`, modeling_vilt.py:190-207;
 * reached from the reference through TomViltForTMSC, ref: vault/models/tomvilt/model.py:281-287): only the modality type
 * is added.  out[map(r)] = src[r] + vec with map(r) = (r / rpg) * gstride + goff + r % rpg (the image rows of the fused
 * sequence); backward: dsrc[r] = dx[map(r)], dvec (optional) += their column sums. */
int vault_rows_add_f32(const float* src, const float* vec, float* out, int rows, int H, int rpg, int gstride, int goff,
                       void* stream);
int vault_rows_gather_bwd_f32(const float* dx, float* dsrc, float* dvec, int rows, int H, int rpg, int gstride, int goff,
                              void* stream);

/* ---- head + loss ----------------------------------------------------------------------------
 * pooled = tanh(pre) ; logits = dropout(pooled) Wc^T + bc ; loss_sum += loss_scale * sum_b CE_b
 * (ref: vault/models/vault/model.py:547-550,567-570 ; vault/tmsc_utils/trainer.py:241-242).
 * bwd: dlogits = (softmax - onehot) * grad_scale unless `dlogits` is given; dWc/dbc accumulated;
 * dpre_bf16 [B][H] = d loss / d pre.  C <= 8. */
typedef struct vault_head_args {
  const float* pre; const float* Wc; const float* bc; const int64_t* labels; const float* dlogits;
  float* pooled; float* logits; float* loss_sum; float* dWc; float* dbc; void* dpre_bf16;
  int B, H, C; float loss_scale, grad_scale;
  uint32_t drop_thresh, drop_seed, drop_stream; float drop_scale;
  /* ABI 5: loss_kind 1 = binary cross-entropy with logits on `targets` [B] (f32; C == 1), mean over the batch
   * (nn.BCEWithLogitsLoss, ref: vault/models/vault/trainer.py:55-56: the n_classes = 1 fine-tune); 0 = cross-entropy
   * on the int64 `labels` */
  const float* targets; int loss_kind;
} vault_head_args;
int vault_head_fwd(const vault_head_args* args, void* stream);
int vault_head_bwd(const vault_head_args* args, void* stream);
int vault_tanh_bwd(const float* pooled, const float* dpooled, void* dpre_bf16, long long n, void* stream);
/* elementwise exact-erf GELU of the MLP task heads (HF ViltForQuestionAnswering.classifier, modeling_vilt.py: Linear -
 * LayerNorm - GELU - Linear on the pooled output; reached through ref: vault/models/vault/model.py:472-509):
 * y_bf16 = gelu(x) ; dx = dy * gelu'(x) */
int vault_gelu_fwd(const float* x, void* y_bf16, long long n, void* stream);
int vault_gelu_fwd_f32(const float* x, float* y_f32, long long n, void* stream);   /* MLM head: dense - GELU - LayerNorm */
int vault_gelu_bwd(const float* x, const float* dy, float* dx, long long n, void* stream);

/* ---- optimizer ------------------------------------------------------------------------------
 * transformers==4.48 AdamW as the reference calls it (ref: vault/tmsc_utils/trainer.py:244-254):
 * bias_corr_factor = 1 (correct_bias=False) or sqrt(1-b2^t)/(1-b1^t); decoupled decay after the
 * update.  g is multiplied by grad_scale first (1/world for DP averaging) and zeroed if zero_grad.
 * Also refreshes the bf16 shadow copy used by the GEMMs.  n % 4 == 0.
 * zero_mask (ABI 9; may be null = zero everything): one byte per 64 elements of [0, n) (then n % 64 == 0), 0 = leave g alone
 * there - ranges whose next gradient will be STORED, not accumulated (the un-split weight-gradient tiles of the fused train
 * step: vault_wgrad_grouped with accumulate = 0), which saves their 4 B/param of zeroing. */
int vault_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, long long n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, float bias_corr_factor, float grad_scale,
                     int zero_grad, const unsigned char* zero_mask, void* stream);
int vault_cast_bf16(const float* x, void* y_bf16, long long n, void* stream);
/* Debug census of a 16-bit tensor in the library's operand format (ABI 9; nothing in the reference, which runs fp32:
 * ref vault/tmsc_utils/trainer.py:353-367 has no autocast): out4[0] += elements at the largest finite magnitude (what a
 * saturating conversion leaves), out4[1] += infinities / NaNs, out4[2] += subnormals, out4[3] += zeros.  out4: four
 * unsigned 64-bit counters the caller zeroed.  Behind VaultEngine's VAULT_H16_CENSUS=1 switch (tests/test_gpu_fp16.py). */
int vault_h16_census(const void* x_h16, long long n, unsigned long long* out4, void* stream);
/* ---- data-parallel gradient exchange (ABI 6) -------------------------------------------------------------------
 * The reference is single-device (ref: vault/tmsc_utils/trainer.py:353-369: backward -> optimizer.step on one GPU); these
 * serve the build's own data-parallel step, between the RCCL collectives that vault_amd/train.py issues.
 *   vault_rows_union      sorted list of the distinct values of keys[0..n_keys) that lie in [0, V) (the token ids of ALL
 *                         ranks after an all-gather: the rows of an embedding table any rank touched) -> uniq[0..*count);
 *                         flags_zeroed: V zero ints of scratch, left zeroed.  Same list on every rank.  `count` points at
 *                         TWO ints (ABI 8): count[0] = length of the list, count[1] = number of keys outside [0, V) other
 *                         than the padding value -1 (must be 0: such a row would be left out of the exchange).
 *   vault_rows_gather_f32 out[j][0..H) = table[idx[j]][0..H)     (compact image of the touched gradient rows)
 *   vault_rows_scatter_f32 table[idx[j]][0..H) = src[j][0..H)    (the all-reduced rows back into the dense gradient)
 *   vault_sum_chunks_bf16 out[i] = bf16(sum_k f32(src[k * chunk + i])), k = 0..n_src-1 in that order (f32 accumulation of
 *                         the peers' bf16 gradient chunks after an all-to-all; chunk % 8 == 0)
 *   vault_widen_bf16      y[i] = f32(x[i])  (n % 4 == 0) */
int vault_rows_union(const long long* keys, long long n_keys, int V, int* flags_zeroed, long long* uniq, int* count,
                     void* stream);
int vault_rows_gather_f32(const float* table, const long long* idx, int n_rows, int H, float* out, void* stream);
int vault_rows_scatter_f32(const float* src, const long long* idx, int n_rows, int H, float* table, void* stream);
int vault_sum_chunks_bf16(const void* src_bf16, int n_src, long long chunk, void* out_bf16, void* stream);
int vault_widen_bf16(const void* x_bf16, float* y, long long n, void* stream);
/* ABI 4: dst[b][c][r] = src[b][r][c] for `batch` bf16 matrices of rows x cols (multiples of 64) at uniform element strides
 * (8-aligned): the transposed weight shadow W^T [in][out] of a Linear.  The data gradient dX = dY . W (autograd of
 * HF:models/vilt/modeling_vilt.py:355-414) then runs as a forward-form GEMM (b_mode 0) on the register-direct kernel
 * (cfg 5 / 6), which takes no k-strided weights. */
int vault_transpose_bf16(const void* src, void* dst, int rows, int cols, int batch, long long stride_src,
                         long long stride_dst, void* stream);
/* Split-bf16 operands: out[r][3K] = [hi | lo | hi] (layout 0, activations) or [hi | hi | lo] (layout 1,
 * weights), hi = bf16(x), lo = bf16(x - hi).  One bf16 GEMM over the 3K-long contraction then equals
 * A_hi B_hi + A_lo B_hi + A_hi B_lo: fp32-class products (2^-17) on the bf16 MFMA path - the precise
 * inference mode used to meet the 1e-3 logits parity bar. */
int vault_split3_bf16(const float* x, void* out_bf16, long long rows, int K, int layout, void* stream);

/* ---- stage-level entries (ABI 4): one encoder layer forward / backward per call ---------------------------------
 * For hosts that do not want to re-implement the kernel order.  The caller owns every buffer
 * (vault_layer_workspace_bytes tells how much a layer saves for backward); the calls only enqueue on `stream`.
 *   vault_vilt_layer_fwd/bwd  HF ViltLayer.forward, modeling_vilt.py:430-451 (+ autograd): pre-LN
 *   vault_lm_layer_fwd/bwd    HF RobertaLayer / BertLayer.forward, modeling_roberta.py:421-463: post-LN, dropouts
 * Buffers are token-major [rows_pad][width], rows_pad = rows rounded up to 256 with zero rows behind the valid ones.
 * ViLT uses: x_in (f32) -> n1, qkv, ctx, lse, xm (= x + attention block, f32), n2, act, u (gelu', training) -> x_out.
 * LM uses:   x_in (f32) + x_in_bf16 -> qkv, ctx, lse, xm (= h1, f32), y1 (f32) + n2 (= y1 in bf16), act, u, h2 (f32)
 *            -> x_out (f32) + x_out_bf16.   m1/r1, m2/r2: mean / rstd of the two LayerNorms.
 * wo_t / wf_t: optional transposed bf16 shadows (vault_transpose_bf16) of the attention-out / FFN-out weights: their
 * data gradients then run as forward-form GEMMs on the register-direct kernel. */
typedef struct vault_layer_args {
  int B, S, H, FF, heads, rows, rows_pad; float eps;
  const void *wqkv, *wo, *wi, *wf, *wo_t, *wf_t;                       /* bf16 [3H,H] [H,H] [FF,H] [H,FF] ([H,H]^T [FF,H]^T...) */
  const float *bqkv, *bo, *bi, *bf, *ln1w, *ln1b, *ln2w, *ln2b;        /* f32 */
  const float* x_in; const void* x_in_bf16; float* x_out; void* x_out_bf16; const float* keymask;
  void *n1, *qkv, *ctx; float* lse; float* xm; float* y1; void* n2; void* act; void* u; float* h2;
  float *m1, *r1, *m2, *r2;
  uint32_t attn_drop_thresh, hid_drop_thresh, drop_seed, drop_stream_base; float attn_drop_scale, hid_drop_scale;
  int persist;                                                          /* GEMM scheduling, as vault_gemm_args.persist */
  void* splitk_ws; long long splitk_bytes;                              /* ABI 11, optional: vault_gemm_args.splitk_ws for the layer's
                                                                           N = H Linears with long contractions (FFN-out forward, FFN-in /
                                                                           QKV data gradients) */
} vault_layer_args;
/* backward: dy = gradient at the layer output (ViLT: dy_f32 = residual-stream gradient + dy_bf16 = its bf16 copy, the
 * FFN-out dY; LM: dy_bf16 (optional) + dy_f32, summed).  Outputs: dx_f32 / dx_bf16 at the layer input (LM: the two parts
 * of d y, summed by the consumer: dx_bf16 = dqkv . Wqkv, dx_f32 = d h1).  Scratch: dU [rows_pad][FF] bf16, dN / dctx /
 * dmid_bf16 (LM also dh1_bf16) [rows_pad][H] bf16, dqkv [rows_pad][3H] bf16, dmid_f32 [rows_pad][H] f32.  Parameter gradients are ACCUMULATED (+=)
 * into the g_* f32 pointers; NULL skips one; do_wgrad = 0 leaves the four weight gradients to the caller (the engine
 * batches them over layers).  g_bf_below (ViLT): bias gradient of the layer below's FFN-out (= column sums of dx).
 * ViLT with dy_f32 = NULL (ABI 5): the residual-gradient stream lives in bf16 only - dy_bf16 is the stream and the FFN-out dY at
 * once, dmid_f32 is unused, dx_f32 is written only when given (the bottom layer, whose consumers read f32). */
typedef struct vault_layer_bwd_args {
  const vault_layer_args* fwd;
  const void* dy_bf16; const float* dy_f32;
  float* dx_f32; void* dx_bf16;
  void *dU, *dN, *dctx, *dqkv, *dmid_bf16, *dh1_bf16; float* dmid_f32;
  float *g_wqkv, *g_bqkv, *g_wo, *g_bo, *g_wi, *g_bi, *g_wf, *g_bf, *g_ln1w, *g_ln1b, *g_ln2w, *g_ln2b, *g_bf_below;
  int do_wgrad;
} vault_layer_bwd_args;
long long vault_layer_workspace_bytes(int B, int S, int H, int FF, int heads, int train, long long* rows_pad_out);
int vault_vilt_layer_fwd(const vault_layer_args* args, void* stream);
int vault_vilt_layer_bwd(const vault_layer_bwd_args* args, void* stream);
int vault_lm_layer_fwd(const vault_layer_args* args, void* stream);
int vault_lm_layer_bwd(const vault_layer_bwd_args* args, void* stream);

/* ---- stage-level entries (ABI 5): the stages around the encoder layers -------------------------------------------
 * Same contract as the layer entries: the caller owns every buffer, the calls only enqueue on `stream`.  One struct per
 * stage serves both directions (forward ignores the backward-only members).  Token-major f32 buffers; *_pad row counts
 * are multiples of 256 with zero rows behind the valid ones.  Parameter gradients g_* are ACCUMULATED (+=); NULL skips.
 *
 * vault_lm_embed_fwd/bwd: BERT / RoBERTa embeddings, HF modeling_roberta.py:75-121 / modeling_bert.py:69-107:
 *   esum = word[ids] (or inputs_embeds) + pos[position_ids] + type[token_type_ids or 0] ; y = dropout(LN(esum)).
 *   pos_mode 1 = RoBERTa position ids (cumsum over non-pad tokens + pad_id), 0 = arange.  T <= 64.
 *   backward: dy = dy_bf16 (optional) + dy_f32 -> desum [rows_pad][H] (= d inputs_embeds when those were given) and the
 *   table gradients; rows whose rowmask ([B][T] f32, optional) is 0 are skipped in the scatter. */
typedef struct vault_lm_embed_args {
  int B, T, H, rows_pad, pos_mode, pad_id; float eps;
  const int64_t* ids; const int64_t* token_type_ids; const float* inputs_embeds;
  const float *word, *pos, *type, *lnw, *lnb;
  int* pos_ids; float* esum; float* mean; float* rstd; float* y; void* y_bf16;
  uint32_t drop_thresh, drop_seed, drop_stream; float drop_scale;
  /* backward */
  const void* dy_bf16; const float* dy_f32; float* desum; const float* rowmask;
  float *g_word, *g_pos, *g_type, *g_lnw, *g_lnb;
} vault_lm_embed_args;
int vault_lm_embed_fwd(const vault_lm_embed_args* args, void* stream);
int vault_lm_embed_bwd(const vault_lm_embed_args* args, void* stream);

/* vault_vilt_text_embed_fwd/bwd: ViLT TextEmbeddings on the text rows of the fused sequence, HF modeling_vilt.py:237-269 as
 * VaultMixin.forward calls it (ref: vault/models/vault/model.py:170-200: the LM's last hidden state arrives as
 * inputs_embeds): vsum = text_src (or word[ids] when NULL) + type[token_type_ids or 0] (+ pos[t] when pos != NULL);
 * x[b*S + t] = LN(vsum[b*T + t]) + mtype0  (modality type 0, modeling_vilt.py:204-207).
 * backward over dx [B*S][H] (the fused-sequence gradient): dvsum [rows_pad][H] (= d text_src), table / LN gradients,
 * g_mtype0; dbeta_scratch [H] f32. */
typedef struct vault_text_embed_args {
  int B, T, S, H, rows_pad; float eps;
  const float* text_src; const int64_t* ids; const int64_t* token_type_ids;
  const float *word, *pos, *type, *lnw, *lnb, *mtype0;
  float* vsum; float* mean; float* rstd; float* x;
  /* backward */
  const float* dx; float* dvsum; float* dbeta_scratch;
  float *g_word, *g_pos, *g_type, *g_lnw, *g_lnb, *g_mtype0;
} vault_text_embed_args;
int vault_vilt_text_embed_fwd(const vault_text_embed_args* args, void* stream);
int vault_vilt_text_embed_bwd(const vault_text_embed_args* args, void* stream);

/* vault_patch_embed_fwd/bwd: ViLT patch embedding on the square pre-training canvas (pixel_mask all ones), HF
 * modeling_vilt.py:290-300 (Conv2d stride = kernel = ps as unfold + GEMM) + visual_embed's position / modality / CLS
 * terms, modeling_vilt.py:160-166,204-215: writes rows b*S + T (CLS) and b*S + T + 1 + p of the fused sequence x.
 * P = (IMG/ps)^2 ; apatch [pad256(B*P)][C*ps*ps] bf16 and addtab [P][H] f32 are saved for backward.
 * backward over dx [B*S][H]: dyp scratch [pad256(B*P)][H] bf16 ; g_w [H][C*ps*ps], g_conv_bias [H], g_pos [(P+1)][H],
 * g_mtype1 [H], g_cls [H].  (Padded batches of differently sized images: the op-level vault_im2col_sel family.) */
typedef struct vault_patch_embed_args {
  int B, C, IMG, ps, T, S, H;
  const float* pixel_values; const void* w_bf16; const float *conv_bias, *pos_emb, *mtype1, *cls;
  void* apatch; float* addtab; float* x; int persist;
  /* backward */
  const float* dx; void* dyp;
  float *g_w, *g_conv_bias, *g_pos, *g_mtype1, *g_cls;
} vault_patch_embed_args;
int vault_patch_embed_fwd(const vault_patch_embed_args* args, void* stream);
int vault_patch_embed_bwd(const vault_patch_embed_args* args, void* stream);

/* vault_head_loss_fwd/bwd: final LayerNorm on the CLS rows, pooler (dense + tanh), classifier and loss: HF
 * modeling_vilt.py ViltModel.forward tail + ViltPooler, ref: vault/models/vault/model.py:547-570, loss as
 * in vault_head_args: cross-entropy on int64 labels, or BCE-with-logits on f32 targets when loss_kind = 1.
 * x [B*S][H] = output of the last encoder layer; h0_bf16 / pre / pooled [pad256(B)][H]; loss [1] is zeroed by the call.
 * backward: zeroes dx_f32 / dx_bf16 [seq_rows_pad][H], then writes the CLS rows' gradient into them; dpre / dh0 scratch
 * [pad256(B)][H] bf16; g_bf_last = bias gradient of the last layer's FFN-out (column sums of dx). */
typedef struct vault_head_loss_args {
  int B, S, H, C, seq_rows_pad; float eps;
  const float* x; const float *lnw, *lnb; const void* wp_bf16; const float* bp; const float *Wc, *bc;
  const int64_t* labels; const float* targets; int loss_kind; float loss_scale, grad_scale;
  void* h0_bf16; float* mean; float* rstd; float* pre; float* pooled; float* logits; float* loss;
  uint32_t drop_thresh, drop_seed, drop_stream; float drop_scale; int persist;
  /* backward */
  void* dpre; void* dh0; float* dx_f32; void* dx_bf16;
  float *g_Wc, *g_bc, *g_wp, *g_bp, *g_lnw, *g_lnb, *g_bf_last;
} vault_head_loss_args;
int vault_head_loss_fwd(const vault_head_loss_args* args, void* stream);
int vault_head_loss_bwd(const vault_head_loss_args* args, void* stream);

/* ---- input pipeline (ABI 5): the image half of ViLT preprocessing for a batch of differently sized uint8 images ------
 * Replaces the per-item CPU call `tokenizer.feature_extractor(image, return_tensors="pt")` of
 * ref: vault/models/vault/dataset.py:337-341 (HF:models/vilt/image_processing_pil_vilt.py:70-98 output size, PIL
 * antialiased bicubic resize, image_transforms.py:118-122 rescale, :417-439 normalise, image_processing_pil_vilt.py:
 * 160-206 pad + pixel_mask) bit-exactly: 8-bit fixed-point resampling in two passes like Pillow's Resample.c.
 * The HOST plans (vault_amd/preprocess.py): output sizes, and per image and axis the taps Pillow's precompute_coeffs
 * gives - `plan` holds, at the descriptor's int32 offsets, bounds [out][2] = (first tap, tap count) and weights
 * [out][ksize] in 22-bit fixed point.  src: the images back to back, [h_in][w_in][3] uint8 each (4-byte aligned base); tmp: [h_in][align4(3 w_out)]
 * uint8 per image (intermediate of the horizontal pass, rows padded to 4-byte multiples, tmp_off multiples of 4); lut [3][256] f32 = value of each 8-bit level per channel after
 * rescale + normalise; pixel_values [B][3][H][W] f32 (zero in the bottom / right padding); pixel_mask [B][H][W] int64
 * and / or f32 (optional); W % 4 == 0 (output sizes are multiples of the size divisor).  max_h_in / max_w_in / max_w_out: maxima
 * over the batch (launch bounds, LDS row buffers) - trusted: an image whose descriptor exceeds them is skipped by the LDS
 * kernel (left unwritten) rather than overrunning its row buffers. */
typedef struct vault_image_desc {
  long long src_off, tmp_off;
  int h_in, w_in, h_out, w_out, ksize_h, ksize_v;
  int hb_off, hk_off, vb_off, vk_off;
} vault_image_desc;
typedef struct vault_preprocess_args {
  const uint8_t* src; uint8_t* tmp; const int* plan; const vault_image_desc* desc; const float* lut;
  float* pixel_values; int64_t* pixel_mask; float* pixel_mask_f32;
  int B, H, W, max_h_in, max_w_out;
  int max_w_in; long long src_bytes;   /* widest source row of the batch; size of src (rows are read as aligned dwords) */
  /* ABI 6, optional: the patch-embedding GEMM's A operand written straight from the vertical pass - the bf16 unfold
   * [B * (H/ps) * (W/ps)][3 ps ps] of the padded canvas (row = image, patch row, patch column; k = channel, y in patch, x in
   * patch: the Conv2d weight's own order, HF:models/vilt/modeling_vilt.py:290-300).  With it pixel_values may be NULL: the
   * f32 NCHW tensor and the separate unfold pass (vault_im2col) are skipped.  H, W multiples of ps; ps % 4 == 0. */
  void* patch_unfold_bf16; int ps;
  /* ABI 12, optional (both zero = unknown): the batch's largest tap count (max of every ksize_h / ksize_v) and the most source
   * rows ONE band of 32 output rows reads (max over images and bands of: first row + tap count of the band's last output row,
   * minus the first row of its first).  With them, and when such a band fits the LDS (480 x 480 -> 384 x 384: 74 KiB), both
   * passes run in ONE launch with the 8-bit intermediate in LDS - `tmp` is then not touched and may be NULL
   * (vault_image_preprocess_is_fused tells, from the maxima alone).  Same bytes out as the two-pass form. */
  int ksize_max, band_rows_max;
} vault_preprocess_args;
int vault_image_preprocess(const vault_preprocess_args* args, void* stream);
int vault_image_preprocess_is_fused(const vault_preprocess_args* args);   /* 1: the one-launch form will run (tmp unused) */

/* Bytes of device memory one forward (+ backward when train) pass over the stages needs for a batch of B items with T text
 * tokens: every activation the stage structs name (saved tensors of all layers, embeddings, head) plus the backward
 * scratch, each buffer rounded up to 256 bytes.  The caller allocates (one arena or many tensors) and hands out the
 * pointers; the library allocates nothing. */
typedef struct vault_model_dims {
  int H, FF, heads, lm_layers, vilt_layers, IMG, ps, C, n_classes;
} vault_model_dims;
long long vault_workspace_bytes(const vault_model_dims* dims, int B, int T, int train);

#ifdef __cplusplus
}
#endif
#endif
