// Helpers shared by the stage-level entries (stage.hip, stage_embed.hip): thin wrappers that fill the op-level argument
// structs of include/vault_hip.h and call the public entry points.
#pragma once
#include "common.h"
#include "../../include/vault_hip.h"

#define CHK(X) { const int rc_ = (X); if (rc_ != 0) return rc_; }

namespace stage {

inline int gemm(const void* A, const void* B, void* out, int M, int N, int K, int lda, int ldb, int ldo, int a_mode, int b_mode, int epi,
         int m_valid, void* st, int persist, const float* bias = nullptr, const float* res = nullptr, const void* aux = nullptr,
         void* out2 = nullptr, float* colsum = nullptr, int splits = 1, int accumulate = 0, int cfg = -1,
         uint32_t dthr = 0, uint32_t dseed = 0, uint32_t dstream = 0, float dscale = 1.f, void* sk_ws = nullptr, long long sk_bytes = 0) {
  vault_gemm_args a{};
  a.splitk_ws = sk_ws; a.splitk_bytes = sk_bytes;
  a.A = A; a.B = B; a.out = out; a.out2 = out2; a.bias = bias; a.res = res; a.aux = aux; a.colsum = colsum;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldo = ldo; a.m_valid = m_valid;
  a.a_mode = a_mode; a.b_mode = b_mode; a.epi = epi; a.cfg = cfg; a.splits = splits; a.accumulate = accumulate;
  a.persist = persist;
  a.drop_thresh = dthr; a.drop_seed = dseed; a.drop_stream = dstream; a.drop_scale = dscale;
  return vault_gemm(&a, st);
}

inline int ln_fwd(const float* x, const float* g, const float* b, float eps, int rows, int H, void* y_bf16, float* y_f32, float* mean,
           float* rstd, void* st) {
  vault_ln_fwd_args a{};
  a.x = x; a.gamma = g; a.beta = b; a.y_bf16 = y_bf16; a.y_f32 = y_f32; a.mean = mean; a.rstd = rstd;
  a.rows = rows; a.H = H; a.eps = eps;
  return vault_layernorm_fwd(&a, st);
}

inline int ln_bwd(const float* x, const float* mean, const float* rstd, const float* gamma, int rows, int H, const void* dy_bf16,
           const float* dy_f32, const float* dres, float* dx_f32, void* dx_bf16, float* dgamma, float* dbeta, float* dbias,
           void* st, uint32_t dthr = 0, uint32_t dseed = 0, uint32_t dstream = 0, float dscale = 1.f,
           const void* dres_bf16 = nullptr) {
  vault_ln_bwd_args a{};
  a.dres_bf16 = dres_bf16;
  a.dy_bf16 = dy_bf16; a.dy_f32 = dy_f32; a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.dres = dres;
  a.dx_f32 = dx_f32; a.dx_bf16 = dx_bf16; a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias;
  a.rows = rows; a.H = H;
  a.drop_thresh = dthr; a.drop_seed = dseed; a.drop_stream = dstream; a.drop_scale = dscale;
  return vault_layernorm_bwd(&a, st);
}


// dY side of the weight gradients: dW += dY^T X, accumulated (atomic epilogue) into the caller's f32 gradient
inline int wgrad(const void* dy, const void* x, float* dw, int Mp, int Nout, int Kin, void* st, int persist) {
  if (!dw) return VAULT_OK;
  const int tiles = (Nout / 256) * (Kin / 256);
  int splits = 1, cfg = 0;
  if (Nout % 256 == 0 && Kin % 256 == 0) {
    cfg = 3;
    splits = 256 / tiles; if (splits > Mp / 128) splits = Mp / 128; if (splits > 16) splits = 16; if (splits < 1) splits = 1;
  } else {
    splits = 768 / ((Nout / 128) * (Kin / 128) > 0 ? (Nout / 128) * (Kin / 128) : 1); if (splits < 1) splits = 1;
    if (splits > Mp / 64) splits = Mp / 64;
  }
  return gemm(dy, x, dw, Nout, Kin, Mp, Nout, Kin, Kin, 1, 1, 5, 0, st, persist, nullptr, nullptr, nullptr, nullptr, nullptr, splits, 1, cfg);
}


}  // namespace stage
