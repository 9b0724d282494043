// Stage-level C ABI: one encoder layer forward / backward per call.  A host in any language binds these
// instead of re-implementing the kernel order; every buffer is the caller's (sizes: vault_layer_workspace_bytes), the
// calls only enqueue on the caller's stream.
//
//   ViLT layer (pre-LN)  HF:models/vilt/modeling_vilt.py:430-451   x' = x + Wo attn(LN1 x) ; out = x' + W2 gelu(W1 LN2 x')
//   LM layer (post-LN)   HF:models/roberta/modeling_roberta.py:421-463 (BERT: modeling_bert.py, same structure)
//                        h1 = drop(Wo attn(y)) + y ; y1 = LN1 h1 ; h2 = drop(W2 gelu(W1 y1)) + y1 ; out = LN2 h2
// The sequencing is the engine's (vault_amd/engine.py calls these for its layers): LayerNorm emits the bf16 GEMM
// operand, QKV is one fused [3H, H] GEMM, the residual adds live in GEMM epilogues, the bias gradients are column sums
// fused into the kernels that produce the corresponding dY.
#include "common.h"
#include "../../include/vault_hip.h"

#include "stage_util.h"

using namespace stage;

namespace {

int attn(const vault_layer_args* L, int bwd, const void* dctx, void* dqkv, void* st) {
  vault_attn_args a{};
  a.qkv = L->qkv; a.keymask = L->keymask; a.ctx = L->ctx; a.lse = L->lse; a.dctx = dctx; a.dqkv = dqkv;
  a.B = L->B; a.S = L->S; a.H = L->H; a.heads = L->heads;
  if (L->attn_drop_thresh) {
    a.drop_thresh = L->attn_drop_thresh; a.drop_seed = L->drop_seed; a.drop_stream = L->drop_stream_base + 2;
    a.drop_scale = L->attn_drop_scale;
  }
  return bwd ? vault_attention_bwd(&a, st) : vault_attention_fwd(&a, st);
}

bool bad(const vault_layer_args* L) {
  return !L || L->B <= 0 || L->S <= 0 || L->H <= 0 || (L->H & 255) || L->FF <= 0 || (L->FF & 127) || L->H != L->heads * 64 ||
         L->rows != L->B * L->S || L->rows_pad < L->rows || (L->rows_pad & 255) || !L->wqkv || !L->wo || !L->wi || !L->wf ||
         !L->qkv || !L->ctx || !L->lse || !L->act;
}

}  // namespace

extern "C" long long vault_layer_workspace_bytes(int B, int S, int H, int FF, int heads, int train, long long* rows_pad_out) {
  if (B <= 0 || S <= 0 || H <= 0 || FF <= 0 || heads <= 0) return -1;
  const long long M = (long long)B * S, Mp = (M + 255) / 256 * 256;
  if (rows_pad_out) *rows_pad_out = Mp;
  // per layer, saved for backward: n1 / yb (bf16 H), qkv (bf16 3H), ctx (bf16 H), xm / h1 (f32 H), n2 / y1b (bf16 H),
  // act (bf16 FF), u = gelu' (bf16 FF, training only), lse (f32 B x heads x S), four row statistics (f32), the layer
  // output x (f32 H) [+ LM: y1 (f32 H), h2 (f32 H), yb_out (bf16 H)]
  long long per_row = 2LL * H + 6LL * H + 2LL * H + 4LL * H + 2LL * H + 2LL * FF + (train ? 2LL * FF : 0) + 4LL * H + 16;
  return Mp * per_row + 4LL * B * heads * S;
}

// ---------------------------------------------------------------- ViLT layer (pre-LN)
extern "C" int vault_vilt_layer_fwd(const vault_layer_args* L, void* st) {
  if (bad(L) || !L->x_in || !L->x_out || !L->n1 || !L->n2 || !L->xm) return VAULT_EINVAL;
  const int M = L->rows, Mp = L->rows_pad, H = L->H, FF = L->FF;
  CHK(ln_fwd(L->x_in, L->ln1w, L->ln1b, L->eps, M, H, L->n1, nullptr, L->m1, L->r1, st));
  CHK(gemm(L->n1, L->wqkv, L->qkv, Mp, 3 * H, H, H, H, 3 * H, 0, 0, 0, M, st, L->persist, L->bqkv));
  CHK(attn(L, 0, nullptr, nullptr, st));
  CHK(gemm(L->ctx, L->wo, L->xm, Mp, H, H, H, H, H, 0, 0, 3, M, st, L->persist, L->bo, L->x_in));
  CHK(ln_fwd(L->xm, L->ln2w, L->ln2b, L->eps, M, H, L->n2, nullptr, L->m2, L->r2, st));
  CHK(gemm(L->n2, L->wi, L->act, Mp, FF, H, H, H, FF, 0, 0, 1, M, st, L->persist, L->bi, nullptr, nullptr, L->u));
  CHK(gemm(L->act, L->wf, L->x_out, Mp, H, FF, FF, FF, H, 0, 0, 3, M, st, L->persist, L->bf, L->xm, nullptr, nullptr, nullptr, 1, 0, -1,
           0, 0, 0, 1.f, L->splitk_ws, L->splitk_bytes));
  return VAULT_OK;
}

extern "C" int vault_vilt_layer_bwd(const vault_layer_bwd_args* G, void* st) {
  if (!G || bad(G->fwd)) return VAULT_EINVAL;
  const vault_layer_args* L = G->fwd;
  if (!L->u || !G->dy_bf16 || !G->dx_bf16 || !G->dU || !G->dN || !G->dmid_bf16 || !G->dctx || !G->dqkv) return VAULT_EINVAL;
  // dy_f32 == NULL: the residual-gradient stream lives in bf16 only (dy_bf16 is stream and FFN-out dY at once; dmid_f32 is
  // not used; dx_f32 is written only if given - the stack's bottom layer, whose consumers read f32)
  const bool bf16_stream = G->dy_f32 == nullptr;
  if (!bf16_stream && (!G->dx_f32 || !G->dmid_f32)) return VAULT_EINVAL;
  const int M = L->rows, Mp = L->rows_pad, H = L->H, FF = L->FF;
  // FFN: dU = (dY . W2) * gelu'  (+ column sums = d b1) ; dN = dU . W1
  if (L->wf_t) { CHK(gemm(G->dy_bf16, L->wf_t, G->dU, Mp, FF, H, H, H, FF, 0, 0, 2, M, st, L->persist, nullptr, nullptr, L->u, nullptr, G->g_bi)); }
  else { CHK(gemm(G->dy_bf16, L->wf, G->dU, Mp, FF, H, H, FF, FF, 0, 1, 2, M, st, L->persist, nullptr, nullptr, L->u, nullptr, G->g_bi)); }
  if (G->do_wgrad) {
    CHK(wgrad(G->dy_bf16, L->act, G->g_wf, Mp, H, FF, st, L->persist));
    CHK(wgrad(G->dU, L->n2, G->g_wi, Mp, FF, H, st, L->persist));
  }
  CHK(gemm(G->dU, L->wi, G->dN, Mp, H, FF, FF, H, H, 0, 1, 0, M, st, L->persist, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, -1,
           0, 0, 0, 1.f, L->splitk_ws, L->splitk_bytes));
  // x' = x + attn-out: LN2 backward adds the residual gradient; its bf16 output is attn-out's dY (column sums = d bo)
  CHK(ln_bwd(L->xm, L->m2, L->r2, L->ln2w, M, H, G->dN, nullptr, G->dy_f32, bf16_stream ? nullptr : G->dmid_f32, G->dmid_bf16, G->g_ln2w,
             G->g_ln2b, G->g_bo, st, 0, 0, 0, 1.f, bf16_stream ? G->dy_bf16 : nullptr));
  if (L->wo_t) { CHK(gemm(G->dmid_bf16, L->wo_t, G->dctx, Mp, H, H, H, H, H, 0, 0, 0, M, st, L->persist)); }
  else { CHK(gemm(G->dmid_bf16, L->wo, G->dctx, Mp, H, H, H, H, H, 0, 1, 0, M, st, L->persist)); }
  if (G->do_wgrad) CHK(wgrad(G->dmid_bf16, L->ctx, G->g_wo, Mp, H, H, st, L->persist));
  CHK(attn(L, 1, G->dctx, G->dqkv, st));
  CHK(gemm(G->dqkv, L->wqkv, G->dN, Mp, H, 3 * H, 3 * H, H, H, 0, 1, 0, M, st, L->persist, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0,
           -1, 0, 0, 0, 1.f, L->splitk_ws, L->splitk_bytes));
  if (G->do_wgrad) CHK(wgrad(G->dqkv, L->n1, G->g_wqkv, Mp, 3 * H, H, st, L->persist));
  if (G->g_bqkv) CHK(vault_colsum(G->dqkv, 3 * H, M, 3 * H, G->g_bqkv, st));
  // d x = LN1 backward (dN) + residual gradient; bf16 copy = dY of the layer below (column sums = its d b2)
  CHK(ln_bwd(L->x_in, L->m1, L->r1, L->ln1w, M, H, G->dN, nullptr, bf16_stream ? nullptr : G->dmid_f32, G->dx_f32, G->dx_bf16, G->g_ln1w,
             G->g_ln1b, G->g_bf_below, st, 0, 0, 0, 1.f, bf16_stream ? G->dmid_bf16 : nullptr));
  return VAULT_OK;
}

// ---------------------------------------------------------------- LM layer (post-LN, dropout on attention probabilities,
// attention output and FFN output in training: the thresholds of vault_layer_args; streams = drop_stream_base + {2, 3, 4})
extern "C" int vault_lm_layer_fwd(const vault_layer_args* L, void* st) {
  if (bad(L) || !L->x_in || !L->x_in_bf16 || !L->x_out || !L->x_out_bf16 || !L->xm || !L->y1 || !L->n2 || !L->h2) return VAULT_EINVAL;
  const int M = L->rows, Mp = L->rows_pad, H = L->H, FF = L->FF;
  CHK(gemm(L->x_in_bf16, L->wqkv, L->qkv, Mp, 3 * H, H, H, H, 3 * H, 0, 0, 0, M, st, L->persist, L->bqkv));
  CHK(attn(L, 0, nullptr, nullptr, st));
  CHK(gemm(L->ctx, L->wo, L->xm, Mp, H, H, H, H, H, 0, 0, 3, M, st, L->persist, L->bo, L->x_in, nullptr, nullptr, nullptr, 1, 0, -1,
           L->hid_drop_thresh, L->drop_seed, L->drop_stream_base + 3, L->hid_drop_scale));
  CHK(ln_fwd(L->xm, L->ln1w, L->ln1b, L->eps, M, H, L->n2, L->y1, L->m1, L->r1, st));
  CHK(gemm(L->n2, L->wi, L->act, Mp, FF, H, H, H, FF, 0, 0, 1, M, st, L->persist, L->bi, nullptr, nullptr, L->u));
  CHK(gemm(L->act, L->wf, L->h2, Mp, H, FF, FF, FF, H, 0, 0, 3, M, st, L->persist, L->bf, L->y1, nullptr, nullptr, nullptr, 1, 0, -1,
           L->hid_drop_thresh, L->drop_seed, L->drop_stream_base + 4, L->hid_drop_scale, L->splitk_ws, L->splitk_bytes));
  CHK(ln_fwd(L->h2, L->ln2w, L->ln2b, L->eps, M, H, L->x_out_bf16, L->x_out, L->m2, L->r2, st));
  return VAULT_OK;
}

extern "C" int vault_lm_layer_bwd(const vault_layer_bwd_args* G, void* st) {
  if (!G || bad(G->fwd)) return VAULT_EINVAL;
  const vault_layer_args* L = G->fwd;
  if (!L->u || !L->h2 || !L->y1 || !G->dy_f32 || !G->dx_f32 || !G->dx_bf16 || !G->dU || !G->dN || !G->dmid_f32 || !G->dmid_bf16 ||
      !G->dctx || !G->dqkv || !G->dh1_bf16)
    return VAULT_EINVAL;
  const int M = L->rows, Mp = L->rows_pad, H = L->H, FF = L->FF;
  // out = LN2(h2): d h2 (f32 -> dmid_f32, bf16 (dropout-masked) -> dmid_bf16 = FFN-out's dY ; column sums = d b2)
  CHK(ln_bwd(L->h2, L->m2, L->r2, L->ln2w, M, H, G->dy_bf16, G->dy_f32, nullptr, G->dmid_f32, G->dmid_bf16, G->g_ln2w, G->g_ln2b, G->g_bf,
             st, L->hid_drop_thresh, L->drop_seed, L->drop_stream_base + 4, L->hid_drop_scale));
  if (L->wf_t) { CHK(gemm(G->dmid_bf16, L->wf_t, G->dU, Mp, FF, H, H, H, FF, 0, 0, 2, M, st, L->persist, nullptr, nullptr, L->u, nullptr, G->g_bi)); }
  else { CHK(gemm(G->dmid_bf16, L->wf, G->dU, Mp, FF, H, H, FF, FF, 0, 1, 2, M, st, L->persist, nullptr, nullptr, L->u, nullptr, G->g_bi)); }
  if (G->do_wgrad) {
    CHK(wgrad(G->dmid_bf16, L->act, G->g_wf, Mp, H, FF, st, L->persist));
    CHK(wgrad(G->dU, L->n2, G->g_wi, Mp, FF, H, st, L->persist));
  }
  CHK(gemm(G->dU, L->wi, G->dN, Mp, H, FF, FF, H, H, 0, 1, 0, M, st, L->persist, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, -1,
           0, 0, 0, 1.f, L->splitk_ws, L->splitk_bytes));
  // y1 = LN1(h1): d y1 = dN (bf16) + d h2 (f32, the residual) ; d h1 -> dx_f32 (the f32 part of d y: h1 = ... + y),
  // its bf16 (dropout-masked) copy = attn-out's dY
  CHK(ln_bwd(L->xm, L->m1, L->r1, L->ln1w, M, H, G->dN, G->dmid_f32, nullptr, G->dx_f32, G->dh1_bf16, G->g_ln1w, G->g_ln1b, G->g_bo, st,
             L->hid_drop_thresh, L->drop_seed, L->drop_stream_base + 3, L->hid_drop_scale));
  if (L->wo_t) { CHK(gemm(G->dh1_bf16, L->wo_t, G->dctx, Mp, H, H, H, H, H, 0, 0, 0, M, st, L->persist)); }
  else { CHK(gemm(G->dh1_bf16, L->wo, G->dctx, Mp, H, H, H, H, H, 0, 1, 0, M, st, L->persist)); }
  if (G->do_wgrad) CHK(wgrad(G->dh1_bf16, L->ctx, G->g_wo, Mp, H, H, st, L->persist));
  CHK(attn(L, 1, G->dctx, G->dqkv, st));
  // d y (layer input) = dqkv . Wqkv (bf16 -> dx_bf16) + d h1 (f32 -> dx_f32, the residual): consumed by the LN2 backward
  // of the layer below
  CHK(gemm(G->dqkv, L->wqkv, G->dx_bf16, Mp, H, 3 * H, 3 * H, H, H, 0, 1, 0, M, st, L->persist, nullptr, nullptr, nullptr, nullptr, nullptr, 1,
           0, -1, 0, 0, 0, 1.f, L->splitk_ws, L->splitk_bytes));
  if (G->do_wgrad) CHK(wgrad(G->dqkv, L->x_in_bf16, G->g_wqkv, Mp, 3 * H, H, st, L->persist));
  if (G->g_bqkv) CHK(vault_colsum(G->dqkv, 3 * H, M, 3 * H, G->g_bqkv, st));
  return VAULT_OK;
}
