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
 *   - "bf16" = 16-bit brain float, "f32" = IEEE binary32; row-major everywhere.
 */
#ifndef VAULT_HIP_H
#define VAULT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int vault_abi_version(void);

/* ---- GEMM -------------------------------------------------------------------------------
 * C[M,N] = A.B with bf16 operands / f32 accumulation and a fused epilogue.  Replaces every
 * nn.Linear forward/backward of HF:models/vilt/modeling_vilt.py:303-414 and
 * HF:models/roberta/modeling_roberta.py:222-398 (called from ref: vault/models/vault/model.py:190,205)
 * and the Conv2d patch projection (modeling_vilt.py:290-300).
 * a_mode 0: A is [M][K]; 1: A is [K][M].   b_mode 0: B is [N][K]; 1: B is [K][N].
 * epi: 0 bf16 out (+bias) | 1 bf16 gelu(acc+bias) (+ out2 = pre-activation) | 2 bf16 acc*gelu'(aux)
 *      3 f32 out = dropout(acc+bias)+res | 4 f32 patch rows (row remap + addtab) | 5 f32 out += acc.
 * M % 128 == 0, N % 128 == 0, K % 64 == 0; buffers must be allocated to those padded sizes.
 * Rows >= m_valid are not stored.  cfg < 0 selects the tile automatically. */
typedef struct vault_gemm_args {
  const void* A; const void* B; void* out; void* out2;
  const float* bias; const float* res; const void* aux; const float* addtab;
  int M, N, K, lda, ldb, ldo, m_valid;
  int a_mode, b_mode, epi, cfg, splits, accumulate;
  int rpg, gstride, goff;
  uint32_t drop_thresh, drop_seed, drop_stream; float drop_scale;
} vault_gemm_args;
int vault_gemm(const vault_gemm_args* args, void* stream);


/* ---- LayerNorm ----------------------------------------------------------------------------
 * fp32 statistics, one wave per row, H % 256 == 0, H <= 1024.  Replaces nn.LayerNorm at
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
} vault_ln_bwd_args;
int vault_layernorm_bwd(const vault_ln_bwd_args* args, void* stream);

/* out[n] += sum_{r < rows} in_bf16[r][n]   (bias gradients); N % 256 == 0 */
int vault_colsum(const void* in_bf16, int ld, int rows, int N, float* out, void* stream);


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
} vault_attn_args;
int vault_attention_fwd(const vault_attn_args* args, void* stream);
int vault_attention_bwd(const vault_attn_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif
