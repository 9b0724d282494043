// Stage-level C ABI, part 2 (ABI 5): the stages around the encoder layers - LM embeddings, ViLT text embeddings, patch
// embedding, head + loss - each forward / backward as ONE call that sequences the op-level entry points in the
// engine's order (vault_amd/engine.py), and the workspace size query.
//
//   vault_lm_embed_*         HF modeling_roberta.py:75-121 / modeling_bert.py:69-107 (embeddings + LayerNorm + dropout)
//   vault_vilt_text_embed_*  HF modeling_vilt.py:237-269 on inputs_embeds (ref: vault/models/vault/model.py:170-200) + modality type 0
//   vault_patch_embed_*      HF modeling_vilt.py:290-300 (Conv2d as unfold + GEMM) + visual_embed constants (160-166, 204-215)
//   vault_head_loss_*        final LayerNorm (CLS rows) + ViltPooler + classifier + loss (ref: model.py:547-570)
#include "stage_util.h"

using namespace stage;

namespace {

inline long long pad256(long long n) { return (n + 255) / 256 * 256; }

int gather(const float* src, float* out, const float* t0, const void* i0, int f0, const float* t1, const void* i1, int f1,
           const float* t2, const void* i2, int f2, int is64_0, int is64_1, int is64_2, int rows, int H, int period, void* st) {
  vault_gather_args a{};
  a.src = src; a.out = out; a.rows = rows; a.H = H; a.period = period;
  a.tab[0] = t0; a.idx[0] = i0; a.fixed[0] = f0; a.is64[0] = is64_0;
  a.tab[1] = t1; a.idx[1] = i1; a.fixed[1] = f1; a.is64[1] = is64_1;
  a.tab[2] = t2; a.idx[2] = i2; a.fixed[2] = f2; a.is64[2] = is64_2;
  return vault_gather_sum(&a, st);
}

int scatter(const float* src, float* t0, const void* i0, int f0, float* t1, const void* i1, int f1, float* t2, const void* i2,
            int f2, int is64_0, int is64_1, int is64_2, int rows, int H, int period, const float* rowmask, void* st) {
  vault_gather_args a{};
  a.src = src; a.rows = rows; a.H = H; a.period = period; a.rowmask = rowmask;
  a.tab[0] = t0; a.idx[0] = i0; a.fixed[0] = f0; a.is64[0] = is64_0;
  a.tab[1] = t1; a.idx[1] = i1; a.fixed[1] = f1; a.is64[1] = is64_1;
  a.tab[2] = t2; a.idx[2] = i2; a.fixed[2] = f2; a.is64[2] = is64_2;
  if (!t0 && !t1 && !t2) return VAULT_OK;
  return vault_scatter_add(&a, st);
}

}  // namespace

// ---------------------------------------------------------------- LM embeddings
extern "C" int vault_lm_embed_fwd(const vault_lm_embed_args* E, void* st) {
  if (!E || E->B <= 0 || E->T <= 0 || E->T > 64 || (E->H & 255) || E->rows_pad < E->B * E->T || !E->ids || !E->pos || !E->type ||
      !E->lnw || !E->lnb || !E->pos_ids || !E->esum || !E->mean || !E->rstd || !E->y || (!E->word && !E->inputs_embeds))
    return VAULT_EINVAL;
  const int rows = E->B * E->T;
  CHK(vault_position_ids(E->ids, E->pos_ids, E->B, E->T, E->pos_mode, E->pad_id, st));
  // table 0: word embeddings (skipped when inputs_embeds stand in), 1: positions, 2: token types (ids or row 0)
  CHK(gather(E->inputs_embeds, E->esum, E->inputs_embeds ? nullptr : E->word, E->ids, 0, E->pos, E->pos_ids, 0, E->type,
             E->token_type_ids, 0, 1, 0, 1, rows, E->H, 1, st));
  vault_ln_fwd_args a{};
  a.x = E->esum; a.gamma = E->lnw; a.beta = E->lnb; a.y_f32 = E->y; a.y_bf16 = E->y_bf16; a.mean = E->mean; a.rstd = E->rstd;
  a.rows = rows; a.H = E->H; a.eps = E->eps;
  a.drop_thresh = E->drop_thresh; a.drop_seed = E->drop_seed; a.drop_stream = E->drop_stream; a.drop_scale = E->drop_scale;
  return vault_layernorm_fwd(&a, st);
}

extern "C" int vault_lm_embed_bwd(const vault_lm_embed_args* E, void* st) {
  if (!E || E->B <= 0 || E->T <= 0 || (E->H & 255) || !E->esum || !E->mean || !E->rstd || !E->lnw || !E->desum || !E->pos_ids ||
      (!E->dy_bf16 && !E->dy_f32))
    return VAULT_EINVAL;
  const int rows = E->B * E->T;
  vault_ln_bwd_args a{};
  a.dy_bf16 = E->dy_bf16; a.dy_f32 = E->dy_f32; a.x = E->esum; a.mean = E->mean; a.rstd = E->rstd; a.gamma = E->lnw;
  a.dx_f32 = E->desum; a.dgamma = E->g_lnw; a.dbeta = E->g_lnb; a.rows = rows; a.H = E->H;
  a.drop_thresh = E->drop_thresh; a.drop_seed = E->drop_seed; a.drop_stream = E->drop_stream; a.drop_scale = E->drop_scale;
  a.drop_on_dy = 1;      // y = dropout(LN(esum))
  CHK(vault_layernorm_bwd(&a, st));
  return scatter(E->desum, E->inputs_embeds ? nullptr : E->g_word, E->ids, 0, E->g_pos, E->pos_ids, 0, E->g_type, E->token_type_ids, 0,
                 1, 0, 1, rows, E->H, 1, E->rowmask, st);
}

// ---------------------------------------------------------------- ViLT text embeddings (text rows of the fused sequence)
extern "C" int vault_vilt_text_embed_fwd(const vault_text_embed_args* E, void* st) {
  if (!E || E->B <= 0 || E->T <= 0 || E->S <= E->T || (E->H & 255) || E->rows_pad < E->B * E->T || !E->type || !E->lnw || !E->lnb ||
      !E->mtype0 || !E->vsum || !E->mean || !E->rstd || !E->x || (!E->text_src && (!E->word || !E->ids)))
    return VAULT_EINVAL;
  const int rows = E->B * E->T;
  // table 0: token types, 1: word embeddings (only without text_src), 2: positions t = row % T (optional)
  CHK(gather(E->text_src, E->vsum, E->type, E->token_type_ids, 0, E->text_src ? nullptr : E->word, E->ids, 0, E->pos, nullptr, -2,
             1, 1, 0, rows, E->H, E->T, st));
  vault_ln_fwd_args a{};
  a.x = E->vsum; a.gamma = E->lnw; a.beta = E->lnb; a.post_add = E->mtype0; a.y_f32 = E->x; a.mean = E->mean; a.rstd = E->rstd;
  a.rows = rows; a.H = E->H; a.eps = E->eps;
  a.y_rpg = E->T; a.y_gstride = E->S; a.y_goff = 0;
  return vault_layernorm_fwd(&a, st);
}

extern "C" int vault_vilt_text_embed_bwd(const vault_text_embed_args* E, void* st) {
  if (!E || E->B <= 0 || E->T <= 0 || E->S <= E->T || (E->H & 255) || !E->vsum || !E->mean || !E->rstd || !E->lnw || !E->dx ||
      !E->dvsum || !E->dbeta_scratch)
    return VAULT_EINVAL;
  const int rows = E->B * E->T;
  hipStream_t hs = reinterpret_cast<hipStream_t>(st);
  // out = LN(.) + mtype0: d mtype0 = sum dy = this call's d beta - taken through the scratch vector (g_lnb accumulates)
  if (hipMemsetAsync(E->dbeta_scratch, 0, sizeof(float) * E->H, hs) != hipSuccess) return VAULT_EINVAL;
  vault_ln_bwd_args a{};
  a.dy_f32 = E->dx; a.x = E->vsum; a.mean = E->mean; a.rstd = E->rstd; a.gamma = E->lnw; a.dx_f32 = E->dvsum;
  a.dgamma = E->g_lnw; a.dbeta = E->dbeta_scratch; a.rows = rows; a.H = E->H;
  a.dy_rpg = E->T; a.dy_gstride = E->S; a.dy_goff = 0;
  CHK(vault_layernorm_bwd(&a, st));
  if (E->g_lnb) CHK(vault_axpy_f32(E->g_lnb, E->dbeta_scratch, 1.f, E->H, st));
  if (E->g_mtype0) CHK(vault_axpy_f32(E->g_mtype0, E->dbeta_scratch, 1.f, E->H, st));
  return scatter(E->dvsum, E->g_type, E->token_type_ids, 0, E->text_src ? nullptr : E->g_word, E->ids, 0, E->pos ? E->g_pos : nullptr,
                 nullptr, -2, 1, 1, 0, rows, E->H, E->T, nullptr, st);
}

// ---------------------------------------------------------------- patch embedding (square canvas, all pixels valid)
extern "C" int vault_patch_embed_fwd(const vault_patch_embed_args* E, void* st) {
  if (!E || E->B <= 0 || E->C <= 0 || E->ps <= 0 || (E->ps & 7) || E->IMG % E->ps || (E->H & 255) || !E->pixel_values || !E->w_bf16 ||
      !E->conv_bias || !E->pos_emb || !E->mtype1 || !E->cls || !E->apatch || !E->addtab || !E->x)
    return VAULT_EINVAL;
  const int G = E->IMG / E->ps, P = G * G, Kp = E->C * E->ps * E->ps;
  if (E->S != E->T + 1 + P || (Kp & 63)) return VAULT_EINVAL;
  const int Mpp = (int)pad256((long long)E->B * P);
  CHK(vault_im2col(E->pixel_values, E->apatch, E->B, E->C, E->IMG, E->ps, 0, st));
  CHK(vault_image_consts(E->conv_bias, E->pos_emb, E->mtype1, E->cls, E->addtab, E->x, P, E->H, E->B, E->S, E->T, st));
  vault_gemm_args a{};
  a.A = E->apatch; a.B = E->w_bf16; a.out = E->x; a.addtab = E->addtab;
  a.M = Mpp; a.N = E->H; a.K = Kp; a.lda = Kp; a.ldb = Kp; a.ldo = E->H; a.m_valid = E->B * P;
  a.epi = 4; a.cfg = -1; a.splits = 1; a.rpg = P; a.gstride = E->S; a.goff = E->T + 1; a.persist = E->persist;
  return vault_gemm(&a, st);
}

extern "C" int vault_patch_embed_bwd(const vault_patch_embed_args* E, void* st) {
  if (!E || E->B <= 0 || E->C <= 0 || E->ps <= 0 || E->IMG % E->ps || (E->H & 255) || !E->dx || !E->dyp || !E->apatch || !E->g_pos ||
      !E->g_mtype1 || !E->g_cls || !E->g_conv_bias)
    return VAULT_EINVAL;
  const int G = E->IMG / E->ps, P = G * G, Kp = E->C * E->ps * E->ps;
  const int Mpp = (int)pad256((long long)E->B * P);
  CHK(vault_image_rows_bwd(E->dx, E->g_pos, E->g_mtype1, E->g_cls, E->g_conv_bias, E->dyp, P, E->H, E->B, E->S, E->T, st));
  return wgrad(E->dyp, E->apatch, E->g_w, Mpp, E->H, Kp, st, E->persist);
}

// ---------------------------------------------------------------- final LayerNorm (CLS rows) + pooler + classifier + loss
extern "C" int vault_head_loss_fwd(const vault_head_loss_args* E, void* st) {
  if (!E || E->B <= 0 || E->S <= 0 || (E->H & 255) || E->C < 0 || !E->x || !E->lnw || !E->lnb || !E->wp_bf16 || !E->bp || !E->h0_bf16 ||
      !E->mean || !E->rstd || !E->pre || !E->pooled)
    return VAULT_EINVAL;
  const int Bp = (int)pad256(E->B);
  hipStream_t hs = reinterpret_cast<hipStream_t>(st);
  vault_ln_fwd_args a{};
  a.x = E->x; a.gamma = E->lnw; a.beta = E->lnb; a.y_bf16 = E->h0_bf16; a.mean = E->mean; a.rstd = E->rstd;
  a.rows = E->B; a.H = E->H; a.eps = E->eps; a.x_rpg = 1; a.x_gstride = E->S; a.x_goff = 0;
  CHK(vault_layernorm_fwd(&a, st));
  CHK(gemm(E->h0_bf16, E->wp_bf16, E->pre, Bp, E->H, E->H, E->H, E->H, E->H, 0, 0, 3, E->B, st, E->persist, E->bp));
  vault_head_args h{};
  h.pre = E->pre; h.Wc = E->Wc; h.bc = E->bc; h.labels = E->labels; h.targets = E->targets; h.loss_kind = E->loss_kind;
  h.pooled = E->pooled; h.logits = E->logits; h.B = E->B; h.H = E->H; h.C = E->C; h.loss_scale = E->loss_scale;
  h.drop_thresh = E->drop_thresh; h.drop_seed = E->drop_seed; h.drop_stream = E->drop_stream; h.drop_scale = E->drop_scale;
  if (E->loss && (E->labels || E->targets)) {
    if (hipMemsetAsync(E->loss, 0, sizeof(float), hs) != hipSuccess) return VAULT_EINVAL;
    h.loss_sum = E->loss;
  }
  return vault_head_fwd(&h, st);
}

extern "C" int vault_head_loss_bwd(const vault_head_loss_args* E, void* st) {
  if (!E || E->B <= 0 || E->S <= 0 || (E->H & 255) || E->C <= 0 || !E->x || !E->lnw || !E->wp_bf16 || !E->h0_bf16 || !E->mean ||
      !E->rstd || !E->pooled || !E->logits || !E->Wc || !E->dpre || !E->dh0 || !E->dx_f32 || !E->dx_bf16 || !E->g_Wc || !E->g_bc ||
      E->seq_rows_pad < E->B * E->S)
    return VAULT_EINVAL;
  const int Bp = (int)pad256(E->B);
  hipStream_t hs = reinterpret_cast<hipStream_t>(st);
  if (hipMemsetAsync(E->dx_f32, 0, sizeof(float) * (size_t)E->seq_rows_pad * E->H, hs) != hipSuccess) return VAULT_EINVAL;
  if (hipMemsetAsync(E->dx_bf16, 0, 2 * (size_t)E->seq_rows_pad * E->H, hs) != hipSuccess) return VAULT_EINVAL;
  if (hipMemsetAsync(E->dpre, 0, 2 * (size_t)Bp * E->H, hs) != hipSuccess) return VAULT_EINVAL;
  vault_head_args h{};
  h.pooled = E->pooled; h.logits = E->logits; h.labels = E->labels; h.targets = E->targets; h.loss_kind = E->loss_kind;
  h.Wc = E->Wc; h.dWc = E->g_Wc; h.dbc = E->g_bc; h.dpre_bf16 = E->dpre; h.B = E->B; h.H = E->H; h.C = E->C;
  h.grad_scale = E->grad_scale;
  h.drop_thresh = E->drop_thresh; h.drop_seed = E->drop_seed; h.drop_stream = E->drop_stream; h.drop_scale = E->drop_scale;
  CHK(vault_head_bwd(&h, st));
  CHK(wgrad(E->dpre, E->h0_bf16, E->g_wp, Bp, E->H, E->H, st, E->persist));
  if (E->g_bp) CHK(vault_colsum(E->dpre, E->H, E->B, E->H, E->g_bp, st));
  CHK(gemm(E->dpre, E->wp_bf16, E->dh0, Bp, E->H, E->H, E->H, E->H, E->H, 0, 1, 0, E->B, st, E->persist));
  vault_ln_bwd_args a{};
  a.dy_bf16 = E->dh0; a.x = E->x; a.mean = E->mean; a.rstd = E->rstd; a.gamma = E->lnw; a.dx_f32 = E->dx_f32; a.dx_bf16 = E->dx_bf16;
  a.dgamma = E->g_lnw; a.dbeta = E->g_lnb; a.dbias = E->g_bf_last; a.rows = E->B; a.H = E->H;
  a.x_rpg = 1; a.x_gstride = E->S; a.x_goff = 0; a.dx_rpg = 1; a.dx_gstride = E->S; a.dx_goff = 0;
  return vault_layernorm_bwd(&a, st);
}

// ---------------------------------------------------------------- workspace size of a whole pass
extern "C" long long vault_workspace_bytes(const vault_model_dims* D, int B, int T, int train) {
  if (!D || B <= 0 || T <= 0 || D->H <= 0 || D->FF <= 0 || D->heads <= 0 || D->vilt_layers <= 0 || D->lm_layers < 0 || D->ps <= 0 ||
      D->IMG % D->ps)
    return -1;
  const long long H = D->H, FF = D->FF;
  const long long P = (long long)(D->IMG / D->ps) * (D->IMG / D->ps), S = T + 1 + P;
  const long long Mp = pad256((long long)B * S), Mlp = pad256((long long)B * T), Mpp = pad256((long long)B * P), Bp = pad256(B);
  const long long Kp = (long long)D->C * D->ps * D->ps;
  auto r = [](long long b) { return (b + 255) / 256 * 256; };
  long long rp = 0, total = 0;
  // encoder layers (saved activations of every layer when training, of one layer otherwise) + the first layer's input
  total += (train ? D->vilt_layers : 1) * vault_layer_workspace_bytes(B, (int)S, D->H, D->FF, D->heads, train, &rp) + r(4 * Mp * H);
  if (D->lm_layers > 0)
    total += (train ? D->lm_layers : 1) * vault_layer_workspace_bytes(B, T, D->H, D->FF, D->heads, train, &rp) + r(6 * Mlp * H);
  // LM embeddings: pos_ids, esum, mean, rstd ; ViLT text embeddings: vsum, mean, rstd ; patch: apatch, addtab
  if (D->lm_layers > 0) total += r(4LL * B * T) + r(4 * Mlp * H) + 2 * r(4 * Mlp);
  total += r(4 * Mlp * H) + 2 * r(4 * Mlp) + r(2 * Mpp * Kp) + r(4 * P * H);
  // head: h0, mean, rstd, pre, pooled, logits, loss
  total += r(2 * Bp * H) + 2 * r(4 * Bp) + 2 * r(4 * Bp * H) + r(4LL * B * (D->n_classes > 0 ? D->n_classes : 1)) + 256;
  if (train) {
    // backward scratch: two f32 + two bf16 residual-gradient streams, dU, dN, dctx, dqkv, dmid (ViLT rows; the LM's fit inside)
    total += 2 * r(4 * Mp * H) + 2 * r(2 * Mp * H) + r(2 * Mp * FF) + 2 * r(2 * Mp * H) + r(2 * Mp * 3 * H) + r(2 * Mp * H) + r(4 * Mp * H);
    // embeddings / head backward: dyp, dvsum, dbeta scratch, desum, dpre, dh0
    total += r(2 * Mpp * H) + 2 * r(4 * Mlp * H) + r(4 * H) + 2 * r(2 * Bp * H);
  }
  return total;
}
