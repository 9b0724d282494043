// C-ABI wrapper of the GEMM (see include/vault_hip.h).
#include "common.h"
#include "gemm.h"
#include "../../include/vault_hip.h"

int vault_gemm_mx8_launch(const GemmParams& p, const void* a_scale, const void* b_scale, int epi, int cfg, hipStream_t st);

static GemmParams params_of(const vault_gemm_args* a) {
  GemmParams p{};
  p.A = reinterpret_cast<const h16*>(a->A);
  p.B = reinterpret_cast<const h16*>(a->B);
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.lda = a->lda; p.ldb = a->ldb; p.ldo = a->ldo;
  p.m_valid = a->m_valid > 0 ? a->m_valid : a->M;
  p.splits = a->splits; p.accumulate = a->accumulate;
  p.out = a->out; p.out2 = a->out2;
  p.bias = a->bias; p.res = a->res;
  p.aux = reinterpret_cast<const h16*>(a->aux);
  p.addtab = a->addtab; p.colsum = a->colsum; p.split3 = a->split3; p.rpg = a->rpg; p.gstride = a->gstride; p.goff = a->goff;
  p.drop_thresh = a->drop_thresh; p.drop_seed = a->drop_seed; p.drop_stream = a->drop_stream;
  p.drop_scale = a->drop_scale;
  p.gn = a->gn;
  p.persist = a->persist;
  p.batch = a->batch; p.batch_a = a->batch_a; p.batch_b = a->batch_b; p.batch_o = a->batch_o;
  p.aux_u8 = a->aux_u8;
  p.out_hm = a->out_hm; p.a_hm = a->a_hm;
  p.out_q = a->out_q; p.out_scale = a->out_scale;
  p.sk_ws = a->splitk_ws; p.sk_bytes = a->splitk_ws ? a->splitk_bytes : 0;
  return p;
}

extern "C" int vault_gemm(const vault_gemm_args* a, void* stream) {
  if (a == nullptr) return VAULT_EINVAL;
  return vault_gemm_launch(params_of(a), a->a_mode, a->b_mode, a->epi, a->cfg, reinterpret_cast<hipStream_t>(stream));
}

int vault_gemm_resolve(GemmParams& p, int a_mode, int b_mode, int epi, int cfg);

extern "C" int vault_gemm_plan(const vault_gemm_args* a) {
  if (a == nullptr) return -VAULT_EINVAL;
  GemmParams p = params_of(a);
  return vault_gemm_resolve(p, a->a_mode, a->b_mode, a->epi, a->cfg);
}

extern "C" int vault_gemm_mxfp8(const vault_gemm_args* a, const void* a_scale, const void* b_scale, void* stream) {
  if (a == nullptr || a->a_mode != 0 || a->b_mode != 0 || a->splits > 1) return VAULT_EINVAL;
  return vault_gemm_mx8_launch(params_of(a), a_scale, b_scale, a->epi, a->cfg, reinterpret_cast<hipStream_t>(stream));
}

int vault_gemm_mx8_resolve(const GemmParams& p, const void* a_scale, const void* b_scale, int epi, int cfg, GemmParams& q);

extern "C" int vault_gemm_mxfp8_plan(const vault_gemm_args* a) {
  if (a == nullptr || a->a_mode != 0 || a->b_mode != 0 || a->splits > 1) return -VAULT_EINVAL;
  GemmParams q;
  return vault_gemm_mx8_resolve(params_of(a), a->A, a->B, a->epi, a->cfg, q);   // (any non-null pointers stand in for the scales)
}

extern "C" int vault_wgrad_grouped(const vault_wgrad_grouped_args* a, void* stream) {
  if (a == nullptr || a->nseg < 1 || a->nseg > 3 || a->tokens <= 0 || (a->tokens & 63)) return VAULT_EINVAL;
  GemmParams p{};
  p.K = a->tokens;
  p.splits = a->splits > 0 ? a->splits : 1;
  p.accumulate = a->accumulate;
  p.persist = a->persist;
  p.nseg = a->nseg;
  for (int k = 0; k < a->nseg; ++k) {
    const vault_wgrad_seg& g = a->seg[k];
    if (g.n_out <= 0 || g.n_in <= 0 || (g.n_out & 255) || (g.n_in & 255) || g.batch < 1) return VAULT_EINVAL;
    GemmParams::Seg& t = p.seg[k];
    t.A = reinterpret_cast<const h16*>(g.dy); t.B = reinterpret_cast<const h16*>(g.x); t.out = g.dw;
    t.tiles_n = g.n_in / 256; t.tiles = (g.n_out / 256) * t.tiles_n;
    t.lda = g.ld_dy; t.ldb = g.ld_x; t.ldo = g.ld_dw;
    t.first = g.first; t.count = g.count;
    if (g.first < 0 || g.count < 1 || g.first + g.count > t.tiles * g.batch) return VAULT_EINVAL;
    t.batch_a = g.batch_dy; t.batch_b = g.batch_x; t.batch_o = g.batch_dw;
    t.a_hm = g.dy_hm;
    if (g.dy_hm < 0 || (g.dy_hm > 0 && g.dy_hm < a->tokens)) return VAULT_EINVAL;
    if (g.dy_hm > 0) t.lda = 64;
  }
  return vault_gemm256_grouped_launch(p, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int vault_abi_version(void) { return 12; }
#ifdef VAULT_F16
extern "C" int vault_operand_format(void) { return 1; }
#else
extern "C" int vault_operand_format(void) { return 0; }
#endif
