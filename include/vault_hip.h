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

#ifdef __cplusplus
}
#endif
#endif
