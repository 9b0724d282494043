"""Task heads of the HIP engine besides the TMSC classifier: the MLP head (HF ViltForQuestionAnswering /
ViltForImagesAndTextClassification ``classifier``) and the MLM head (HF ViltMLMHead), forward and backward
(ref: vault/models/vault/model.py:375-509)."""
from __future__ import annotations

import torch

from . import ops
from .params import _in_format, _pad


class HeadsMixin:
    # ---- MLP task head (HF ViltForQuestionAnswering / ViltForImagesAndTextClassification .classifier) ------------
    def _mlp_forward(self, ws: dict, x_f32: torch.Tensor, B: int) -> torch.Tensor:
        """logits = Linear(GELU(LayerNorm(Linear(x))))  for x [>= B rows, H_in] f32; Linear(H_in, H_mid) -
        LayerNorm(H_mid, eps 1e-5) - GELU - Linear(H_mid, L).  The output projection runs with L padded to 256 columns
        (readable slack behind the parameter buffers).  Buffers live in ``ws``."""
        spec, P = self.spec, self.params
        Hin, Hm = spec.mlp_dims
        L = spec.n_classes
        Lp, Bp = _pad(L), _pad(B)
        bf = self.hdt
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        xb = buf("mlp_xb", (Bp, Hin), bf)
        ops.cast_bf16(x_f32, xb, B * Hin)
        h1 = buf("mlp_h1", (Bp, Hm))
        self._linear(xb, "classifier.0.weight", h1, Bp, Hm, Hin, ops.EPI_F32_RES, B, bias=P.w("classifier.0.bias"))
        n1 = buf("mlp_n1", (Bp, Hm))
        ops.layernorm_fwd(h1, P.w("classifier.1.weight"), P.w("classifier.1.bias"), 1e-5, B, Hm, y_f32=n1,
                          mean=buf("mlp_mean", (Bp,)), rstd=buf("mlp_rstd", (Bp,)))
        a1 = buf("mlp_a1", (Bp, Hm), bf)
        ops.gelu_fwd(n1, a1, Bp * Hm)
        lg = buf("mlp_logits", (Bp, Lp))
        ops.gemm(a1, P.wb("classifier.3.weight", n_elems=Lp * Hm, shape=(Lp, Hm)), lg, Bp, Lp, Hm, Hm, Hm, Lp, 0, 0,
                 ops.EPI_F32_RES, m_valid=B, bias=P.w("classifier.3.bias", n_elems=Lp, shape=(Lp,)))
        return lg[:B, :L]

    def _mlp_backward(self, ws: dict, dlogits: torch.Tensor, B: int, scale: float = 1.0) -> torch.Tensor:
        """Parameter gradients of the MLP head (+=) and d/dx [Bp, H_in] f32 of the last :meth:`_mlp_forward` on ``ws``.
        ``scale``: the gradient scale of the operand format, applied to ``dlogits`` (every result is scaled by it)."""
        spec, P = self.spec, self.params
        Hin, Hm = spec.mlp_dims
        L = spec.n_classes
        Lp, Bp = _pad(L), _pad(B)
        bf = self.hdt
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        dl32 = buf("mlp_dlogits", (Bp, Lp))
        ops.pycall(dl32.zero_)
        dl32[:B, :L].copy_(dlogits.reshape(B, L))
        if scale != 1.0:
            ops.scale(dl32.view(-1), scale, Bp * Lp)
        dlb = buf("mlp_dlogits_b", (Bp, Lp), bf)
        ops.cast_bf16(dl32, dlb, Bp * Lp)
        # output projection: weight-gradient rows >= L are never written (m_valid); its bias gradient is the column sum
        # (the padded columns are zero and land in the slack behind the gradient buffer)
        self._wgrad(dlb, ws["mlp_a1"], "classifier.3.weight", "classifier.3.bias", Bp, Lp, Hm, B, out_rows=L)
        da1 = buf("mlp_da1", (Bp, Hm))
        ops.gemm(dlb, P.wb("classifier.3.weight", n_elems=Lp * Hm, shape=(Lp, Hm)), da1, Bp, Hm, Lp, Lp, Hm, Hm, 0, 1,
                 ops.EPI_F32_RES, m_valid=B)
        dn1 = buf("mlp_dn1", (Bp, Hm))
        ops.gelu_bwd(ws["mlp_n1"], da1, dn1, Bp * Hm)
        dh1b = buf("mlp_dh1b", (Bp, Hm), bf)
        ops.layernorm_bwd(ws["mlp_h1"], ws["mlp_mean"], ws["mlp_rstd"], P.w("classifier.1.weight"), B, Hm, dy_f32=dn1,
                          dx_bf16=dh1b, dgamma=P.gr("classifier.1.weight"), dbeta=P.gr("classifier.1.bias"),
                          dbias=P.gr("classifier.0.bias"))
        self._wgrad(dh1b, ws["mlp_xb"], "classifier.0.weight", None, Bp, Hm, Hin, B)
        dx = buf("mlp_dx", (Bp, Hin))
        ops.gemm(dh1b, P.wb("classifier.0.weight", shape=(Hm, Hin)), dx, Bp, Hin, Hm, Hm, Hin, Hin, 0, 1, ops.EPI_F32_RES,
                 m_valid=B)
        return dx

    # ---- MLM head (HF ViltMLMHead): dense(H, H) - GELU - LayerNorm - decoder tied to ViLT's word embeddings + bias --------
    @_in_format
    def mlm_head_forward(self, x_f32: torch.Tensor) -> torch.Tensor:
        """x [R, H] f32 (text rows of last_hidden_state) -> logits [R, V]."""
        with torch.cuda.device(self.device):
            spec, P = self.spec, self.params
            v = spec.vilt
            H, V = v.hidden_size, v.vocab_size
            R = x_f32.shape[0]
            Rp, Vp = _pad(R), _pad(V)
            ws = self._ws.setdefault(("mlm_head", R), {})
            bf = self.hdt
            buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
            xin = buf("x", (Rp, H))
            xin[:R].copy_(x_f32)
            xb = buf("xb", (Rp, H), bf)
            ops.cast_bf16(xin, xb, Rp * H)
            h1 = buf("h1", (Rp, H))
            self._linear(xb, "mlm_score.transform.dense.weight", h1, Rp, H, H, ops.EPI_F32_RES, R,
                         bias=P.w("mlm_score.transform.dense.bias"))
            a = buf("a", (Rp, H))
            ops.gelu_fwd_f32(h1, a, Rp * H)
            nb = buf("nb", (Rp, H), bf)
            ops.layernorm_fwd(a, P.w("mlm_score.transform.LayerNorm.weight"), P.w("mlm_score.transform.LayerNorm.bias"),
                              v.layer_norm_eps, R, H, y_bf16=nb, mean=buf("mean", (Rp,)), rstd=buf("rstd", (Rp,)))
            lg = buf("logits", (Rp, Vp))
            wn = "embeddings.text_embeddings.word_embeddings.weight"
            ops.gemm(nb, P.wb(wn, n_elems=Vp * H, shape=(Vp, H)), lg, Rp, Vp, H, H, H, Vp, 0, 0, ops.EPI_F32_RES, m_valid=R,
                     bias=P.w("mlm_score.bias", n_elems=Vp, shape=(Vp,)))
            return lg[:R, :V]

    @_in_format
    def mlm_head_backward(self, dlogits: torch.Tensor) -> torch.Tensor:
        self._api_backward_begins()
        with torch.cuda.device(self.device), self._grads_scaled():
            spec, P = self.spec, self.params
            v = spec.vilt
            H, V = v.hidden_size, v.vocab_size
            R = dlogits.shape[0]
            Rp, Vp = _pad(R), _pad(V)
            ws = self._ws[("mlm_head", R)]
            bf = self.hdt
            buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
            dl32 = buf("dlogits", (Rp, Vp))
            dl32.zero_()
            dl32[:R, :V].copy_(dlogits.reshape(R, V))
            if self.grad_scale != 1.0:
                ops.scale(dl32.view(-1), self.grad_scale, Rp * Vp)
            dlb = buf("dlogits_b", (Rp, Vp), bf)
            ops.cast_bf16(dl32, dlb, Rp * Vp)
            wn = "embeddings.text_embeddings.word_embeddings.weight"
            self._wgrad(dlb, ws["nb"], wn, "mlm_score.bias", Rp, Vp, H, R, out_rows=V)
            dn = buf("dn", (Rp, H))
            ops.gemm(dlb, P.wb(wn, n_elems=Vp * H, shape=(Vp, H)), dn, Rp, H, Vp, Vp, H, H, 0, 1, ops.EPI_F32_RES, m_valid=R)
            da = buf("da", (Rp, H))
            ops.layernorm_bwd(ws["a"], ws["mean"], ws["rstd"], P.w("mlm_score.transform.LayerNorm.weight"), R, H, dy_f32=dn,
                              dx_f32=da, dgamma=P.gr("mlm_score.transform.LayerNorm.weight"),
                              dbeta=P.gr("mlm_score.transform.LayerNorm.bias"))
            dh1 = buf("dh1", (Rp, H))
            ops.gelu_bwd(ws["h1"], da, dh1, Rp * H)
            dh1b = buf("dh1b", (Rp, H), bf)
            ops.cast_bf16(dh1, dh1b, Rp * H)
            self._wgrad(dh1b, ws["xb"], "mlm_score.transform.dense.weight", "mlm_score.transform.dense.bias", Rp, H, H, R)
            dx = buf("dx", (Rp, H))
            ops.gemm(dh1b, P.wb("mlm_score.transform.dense.weight", shape=(H, H)), dx, Rp, H, H, H, H, H, 0, 1,
                     ops.EPI_F32_RES, m_valid=R)
            if self.grad_scale != 1.0:
                ops.scale(dx.view(-1), 1.0 / self.grad_scale, dx.numel())
            return dx[:R]

    @_in_format
    def mlp_head_forward(self, x_f32: torch.Tensor, train: bool = True) -> torch.Tensor:
        """The MLP head on an external input [B, H_in] (concatenated pooled outputs of several encoder passes)."""
        with torch.cuda.device(self.device):
            B = x_f32.shape[0]
            ws = self._ws.setdefault(("mlp_head", B), {})
            xin = self._buf(ws, "mlp_xin", (_pad(B), self.spec.mlp_dims[0]), torch.float32)
            xin[:B].copy_(x_f32)
            return self._mlp_forward(ws, xin, B)

    @_in_format
    def mlp_head_backward(self, dlogits: torch.Tensor) -> torch.Tensor:
        self._api_backward_begins()
        with torch.cuda.device(self.device), self._grads_scaled():
            B = dlogits.shape[0]
            dx = self._mlp_backward(self._ws[("mlp_head", B)], dlogits.contiguous().float(), B, scale=self.grad_scale)
            if self.grad_scale != 1.0:
                ops.scale(dx.view(-1), 1.0 / self.grad_scale, dx.numel())
            return dx[:B]

    # ---- backward ---------------------------------------------------------------------------
