"""CPU oracle for the stacked BERT -> ViLT hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file; the product (``vault_amd``) never does and fails loudly without its
HIP library.

What it is: a plain fp32 ``torch`` restatement (functional; no ``transformers``, no
reference code) of the arithmetic the reference reaches through HuggingFace:

  * LM embeddings   HF:models/roberta/modeling_roberta.py:75-155 / models/bert/modeling_bert.py:69-107
  * LM layer        HF:models/roberta/modeling_roberta.py:158-250,329-463 (post-LN, erf-GELU)
  * ViLT text embed HF:models/vilt/modeling_vilt.py:237-269 (with ``inputs_embeds``)
  * patch embed     HF:models/vilt/modeling_vilt.py:290-300 (Conv2d k=s=patch)
  * visual_embed    HF:models/vilt/modeling_vilt.py:92-178 for full pixel masks, static patch
                    order (the reference shuffles valid patches with ``torch.multinomial``;
                    attention is permutation-equivariant so pooler/logits are unchanged,
                    SURVEY §8c D3)
  * modality add    HF:models/vilt/modeling_vilt.py:204-219
  * ViLT layer      HF:models/vilt/modeling_vilt.py:303-451 (pre-LN, additive finfo.min mask)
  * tail            HF:models/vilt/modeling_vilt.py:636-663 (final LN, pooler)
  * glue            ref: vault/models/vault/model.py:151-218 (LM output -> inputs_embeds),
                    model.py:547-570 (Dropout -> Linear head), D1: ViLT text position
                    embeddings are skipped unless ``use_vilt_position_embeddings``
                    (model.py:78-79,113-116 under transformers 4.48)
  * loss            ref: vault/tmsc_utils/trainer.py:241-242 (mean CrossEntropy)
  * optimizer       transformers==4.48 ``AdamW`` as called at trainer.py:244-254
                    (``correct_bias=False``), schedule trainer.py:274-278

Pinning: the reference ships no tests or golden vectors for this path, so the
oracle is pinned against outputs of the reference itself, run in the build container via
``oracle/make_goldens.py`` and committed under ``tests/golden/`` (see
tests/test_oracle_golden.py).  The optimizer has no importable reference under
transformers 5.15 (``transformers.optimization.AdamW`` was removed) and is pinned by
formula against a float64 restatement: that part is "parity unpinned" by the reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def to_torch_state(state: Dict[str, np.ndarray], requires_grad: bool = False) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in state.items():
        t = _t(v).clone().float()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


# Optional emulation of the HIP path's number format: every matmul operand (activations and
# weights of each Linear, Q/K/V/P of attention) is rounded to bf16 before an exact fp32 product,
# exactly where the kernels round.  Off by default (pure fp32 = the reference's arithmetic).
# ``backward=True`` also restates the number format of the HIP BACKWARD pass (what autograd does for
# ref: vault/models/vault/model.py:557-570 in fp32): bf16 dY operands of every data / weight gradient
# GEMM, bf16 data-gradient outputs, the saved bf16 operands, gelu' as stored (bf16, or the 8-bit grid
# q = rne(200 g + 26) of the ViLT FFN at large batch), the recomputed attention probabilities with bf16
# dS / P operands, and the bf16 residual-gradient stream of the pre-LN ViLT stack.
_EMULATE_BF16 = False
_EMULATE_BWD = False
_EMULATE_GELU8 = False
# mutation check of the parity bounds (tests only): (weight-name substring, factor) scales the DATA gradient of the matching
# Linear in the emulated backward - the tests assert that their bounds notice a 1 % error of one data-gradient GEMM
_INJECT_DGRAD = None


# the 16-bit operand type being emulated (the HIP library exists for both: csrc/common.h `h16`) and, for fp16, the static
# power-of-two scale the HIP backward's gradients carry while they are 16-bit tensors (engine.VaultEngine.grad_scale)
_EMU_DTYPE = torch.bfloat16
_EMU_GSCALE = 1.0


class emulate_bf16:
    """``with emulate_bf16():`` - run the oracle with bf16-rounded matmul operands; ``backward=True``: gradients in the
    HIP backward's number format too (``gelu8``: the ViLT FFN keeps gelu' on the 8-bit grid)."""
    dtype, gscale = torch.bfloat16, 1.0

    def __init__(self, backward: bool = False, gelu8: bool = False):
        self.backward, self.gelu8 = backward, gelu8

    def __enter__(self):
        global _EMULATE_BF16, _EMULATE_BWD, _EMULATE_GELU8, _EMU_DTYPE, _EMU_GSCALE
        self._old = (_EMULATE_BF16, _EMULATE_BWD, _EMULATE_GELU8, _EMU_DTYPE, _EMU_GSCALE)
        _EMULATE_BF16, _EMULATE_BWD, _EMULATE_GELU8 = True, self.backward, self.gelu8
        _EMU_DTYPE, _EMU_GSCALE = self.dtype, float(self.gscale)

    def __exit__(self, *a):
        global _EMULATE_BF16, _EMULATE_BWD, _EMULATE_GELU8, _EMU_DTYPE, _EMU_GSCALE
        _EMULATE_BF16, _EMULATE_BWD, _EMULATE_GELU8, _EMU_DTYPE, _EMU_GSCALE = self._old


class emulate_fp16(emulate_bf16):
    """The same emulation for the fp16 build of the HIP library (libvault_hip_f16.so): IEEE half operands, conversions
    saturating at +-65504, and - ``backward=True`` - 16-bit gradient tensors rounded under the static power-of-two gradient
    scale ``grad_scale`` (a tensor the HIP backward stores as fp16(S g) is emulated as fp16(S g) / S)."""
    dtype = torch.float16

    def __init__(self, backward: bool = False, gelu8: bool = False, grad_scale: float = 4096.0):
        super().__init__(backward, gelu8)
        self.gscale = grad_scale


def _rb(x):
    """Round a forward value to the emulated operand type."""
    if _EMU_DTYPE is torch.float16:
        return x.clamp(-65504.0, 65504.0).half().float()
    return x.bfloat16().float()


def _rg(g):
    """Round a gradient tensor the way the HIP backward stores it (under the gradient scale in the fp16 build)."""
    if _EMU_DTYPE is torch.float16:
        return (g * _EMU_GSCALE).clamp(-65504.0, 65504.0).half().float() / _EMU_GSCALE
    return g.bfloat16().float()


class _RoundST(torch.autograd.Function):
    """y = bf16(x); the gradient passes unchanged (the kernels round where they store, not where they differentiate)."""

    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """y = x; the (accumulated) gradient is rounded to bf16 once: a tensor the HIP backward keeps in bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rg(g)


class _LinearEmu(torch.autograd.Function):
    """y = bf16(x) bf16(W)^T + b with the HIP backward: dX = bf16(dY) bf16(W) (stored bf16 when ``round_dx``),
    dW = bf16(dY)^T bf16(x) in fp32, db = column sums of dY."""

    @staticmethod
    def forward(ctx, x, w, b, round_dx, scale_dx=1.0):
        xr, wr = _rb(x), _rb(w)
        ctx.save_for_backward(xr, wr)
        ctx.round_dx, ctx.has_b, ctx.scale_dx = round_dx, b is not None, scale_dx
        return F.linear(xr, wr, b)

    @staticmethod
    def backward(ctx, gy):
        xr, wr = ctx.saved_tensors
        gr = _rg(gy)
        dx = torch.matmul(gr, wr) * ctx.scale_dx
        if ctx.round_dx:
            dx = _rg(dx)
        dw = torch.matmul(gr.reshape(-1, gr.shape[-1]).t(), xr.reshape(-1, xr.shape[-1]))
        db = gy.reshape(-1, gy.shape[-1]).sum(0) if ctx.has_b else None
        return dx, dw, db, None, None


class _GeluEmu(torch.autograd.Function):
    """act = gelu(z) on the fp32 pre-activation; backward multiplies by gelu'(z) AS STORED by the forward epilogue (bf16, or
    the 8-bit grid (rne(200 g + 26) - 26) / 200 saturating at 0 / 255) and stores the product in bf16."""

    @staticmethod
    def forward(ctx, z, u8):
        cdf = 0.5 * (1.0 + torch.erf(z * 0.7071067811865476))
        gp = cdf + z * torch.exp(-0.5 * z * z) * 0.3989422804014327
        if u8:
            gp = torch.clamp(torch.round(gp * 200.0 + 26.0), 0.0, 255.0) * 0.005 - 0.13   # (torch.round: half to even)
        else:
            gp = _rb(gp)
        ctx.save_for_backward(gp)
        return z * cdf

    @staticmethod
    def backward(ctx, g):
        (gp,) = ctx.saved_tensors
        return _rg(g * gp), None


class _AttnEmu(torch.autograd.Function):
    """softmax(q k^T / sqrt(d) + mask) v on bf16 q, k, v with bf16 probabilities in the PV product; backward as the HIP
    kernels: P recomputed in fp32, D = rowsum(dO o O) on the stored bf16 O, dS and P rounded to bf16 as operands of
    dQ = dS K, dK = dS^T Q, dV = P^T dO, results stored in bf16."""

    @staticmethod
    def forward(ctx, q, k, v, mask_add, scale):
        s = torch.matmul(q, k.transpose(-1, -2)) * scale + mask_add
        # the forward kernel feeds the UNNORMALISED exponentials exp(s - max) to the matrix pipe in bf16 and divides the
        # output by their fp32 row sum afterwards; backward recomputes the normalised probabilities from the log-sum-exp
        e = torch.exp(s - s.max(dim=-1, keepdim=True).values)
        lsum = e.sum(-1, keepdim=True)
        o = torch.matmul(_rb(e), v) / lsum
        p = e / lsum
        ctx.save_for_backward(q, k, v, p, _rb(o))
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, go):
        q, k, v, p, o = ctx.saved_tensors
        go = _rg(go)
        dp = torch.matmul(go, v.transpose(-1, -2))
        dsum = (go * o).sum(-1, keepdim=True)
        ds = _rg(p * (dp - dsum))
        dq = torch.matmul(ds, k) * ctx.scale
        dk = torch.matmul(ds.transpose(-1, -2), q) * ctx.scale
        dv = torch.matmul(_rb(p).transpose(-1, -2), go)
        return _rg(dq), _rg(dk), _rg(dv), None, None


def _r(x):
    if _EMULATE_BWD:
        return _RoundST.apply(x)
    return _rb(x) if _EMULATE_BF16 else x


def _lin(x, w, b=None, round_dx=True, name=""):
    if _EMULATE_BWD:
        sc = _INJECT_DGRAD[1] if (_INJECT_DGRAD is not None and _INJECT_DGRAD[0] in name) else 1.0
        return _LinearEmu.apply(x, w, b, round_dx, sc)
    return F.linear(_r(x), _r(w), b)


def _mm(a, b):
    return torch.matmul(_r(a), _r(b))


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _ffn(x, wi, bi, wf, bf_, u8=False):
    """Linear - exact-erf GELU - Linear (HF ViltIntermediate / ViltOutput.dense, RobertaIntermediate / Output.dense)."""
    if _EMULATE_BWD:
        act = _GeluEmu.apply(_lin(x, wi, bi), u8)
        return _lin(act, wf, bf_, round_dx=False)      # (its data gradient is rounded after the gelu' product)
    return _lin(F.gelu(_r(_lin(x, wi, bi))), wf, bf_)


def _grad_bf16(x):
    """Marks a tensor whose gradient the HIP backward holds in bf16 only (the pre-LN ViLT residual-gradient stream)."""
    return _RoundGrad.apply(x) if _EMULATE_BWD else x


def _mha(x, mask_add, P, pre, heads, att_name):
    """softmax(QK^T/sqrt(d) + mask) V ; returns context [B,S,H] (before output dense)."""
    B, S, H = x.shape
    d = H // heads
    if _EMULATE_BWD:
        # one fused [3H, H] projection like the kernels: its data gradient sums the three parts in fp32, one rounding
        w = torch.cat([P[f"{pre}.{att_name}.{n}.weight"] for n in ("query", "key", "value")], dim=0)
        b = torch.cat([P[f"{pre}.{att_name}.{n}.bias"] for n in ("query", "key", "value")], dim=0)
        qkv = _RoundST.apply(_lin(x, w, b, name=f"{pre}.{att_name}.qkv")).view(B, S, 3, heads, d).permute(2, 0, 3, 1, 4)
        c = _AttnEmu.apply(qkv[0], qkv[1], qkv[2], mask_add, 1.0 / math.sqrt(d))
        return c.permute(0, 2, 1, 3).reshape(B, S, H)
    q = _r(_lin(x, P[f"{pre}.{att_name}.query.weight"], P[f"{pre}.{att_name}.query.bias"]))
    k = _r(_lin(x, P[f"{pre}.{att_name}.key.weight"], P[f"{pre}.{att_name}.key.bias"]))
    v = _r(_lin(x, P[f"{pre}.{att_name}.value.weight"], P[f"{pre}.{att_name}.value.bias"]))
    q = q.view(B, S, heads, d).transpose(1, 2)
    k = k.view(B, S, heads, d).transpose(1, 2)
    v = v.view(B, S, heads, d).transpose(1, 2)
    s = _mm(q, k.transpose(-1, -2)) / math.sqrt(d)
    s = s + mask_add
    p = torch.softmax(s, dim=-1)
    c = _mm(p, v)
    return c.permute(0, 2, 1, 3).reshape(B, S, H)


def lm_position_ids(input_ids: torch.Tensor, lm) -> torch.Tensor:
    if lm.kind == "roberta":
        m = input_ids.ne(lm.pad_token_id).int()
        return (torch.cumsum(m, dim=1).type_as(m) * m).long() + lm.pad_token_id
    return torch.arange(input_ids.shape[1]).unsqueeze(0).expand_as(input_ids)


def lm_forward(P, spec, input_ids, attention_mask, token_type_ids=None, taps: Optional[dict] = None,
               inputs_embeds: Optional[torch.Tensor] = None):
    """``inputs_embeds`` [B,T,H] instead of ids (ref model.py:170-190 passes them on to the LM): the word-embedding
    lookup is skipped and the position ids count every position (HF:models/roberta/modeling_roberta.py
    ``create_position_ids_from_inputs_embeds``: pad + 1 .. pad + T; BERT: arange)."""
    lm = spec.lm
    if inputs_embeds is not None:
        B_, T_ = inputs_embeds.shape[:2]
        base = lm.pad_token_id + 1 if lm.kind == "roberta" else 0
        pos = (torch.arange(T_) + base).unsqueeze(0).expand(B_, T_)
        words = inputs_embeds
        like = torch.zeros((B_, T_), dtype=torch.long)
    else:
        pos = lm_position_ids(input_ids, lm)
        words = P["bert.embeddings.word_embeddings.weight"][input_ids]
        like = input_ids
    if token_type_ids is None or lm.type_vocab_size < 2:
        token_type_ids = torch.zeros_like(like)   # ref: model.py:174-180
    x = (words
         + P["bert.embeddings.token_type_embeddings.weight"][token_type_ids]
         + P["bert.embeddings.position_embeddings.weight"][pos])
    x = _ln(x, P["bert.embeddings.LayerNorm.weight"], P["bert.embeddings.LayerNorm.bias"], lm.layer_norm_eps)
    if taps is not None:
        taps["lm_embed"] = x
    mask_add = (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for i in range(lm.num_hidden_layers):
        pre = f"bert.encoder.layer.{i}"
        c = _mha(x, mask_add, P, pre, lm.num_attention_heads, "attention.self")
        a = _lin(c, P[f"{pre}.attention.output.dense.weight"], P[f"{pre}.attention.output.dense.bias"])
        x = _ln(a + x, P[f"{pre}.attention.output.LayerNorm.weight"], P[f"{pre}.attention.output.LayerNorm.bias"],
                lm.layer_norm_eps)
        o = _ffn(x, P[f"{pre}.intermediate.dense.weight"], P[f"{pre}.intermediate.dense.bias"],
                 P[f"{pre}.output.dense.weight"], P[f"{pre}.output.dense.bias"])
        x = _ln(o + x, P[f"{pre}.output.LayerNorm.weight"], P[f"{pre}.output.LayerNorm.bias"], lm.layer_norm_eps)
        if taps is not None:
            taps[f"lm_layer{i}"] = x
    return x


def _select_patches(pixel_mask: torch.Tensor, gh: int, gw: int):
    """The oracle's OWN restatement of the patch bookkeeping of ``ViltEmbeddings.visual_embed``
    (HF:models/vilt/modeling_vilt.py:96-101 mask interpolation, :130-160 valid / non-valid indices and the padding
    draw) - independent of the product's ``vault_amd.spec.select_patches``.  Where the reference samples
    (``torch.multinomial`` over valid patches for the order, over non-valid ones for the padding), the deterministic
    member of its outcome set is taken: valid patches in ``nonzero`` (row-major) order, padding = the non-valid patches
    repeated in order.  Returns numpy sel [B, L] (slot on the gh x gw grid), valid [B, L], hw [B, 2] and L."""
    x_mask = F.interpolate(pixel_mask[:, None, :, :].float(), size=(gh, gw)).long()     # nearest, like HF :97
    x_h = x_mask[:, 0].sum(dim=1)[:, 0]
    x_w = x_mask[:, 0].sum(dim=2)[:, 0]
    eff = x_h * x_w
    L = int(eff.max())
    flat = x_mask.flatten(1)                                                            # [B, gh*gw]
    sel_rows, valid_rows = [], []
    for b in range(flat.shape[0]):
        v_idx = flat[b].nonzero(as_tuple=False)[:, 0]
        nv_idx = (1 - flat[b]).nonzero(as_tuple=False)[:, 0]
        pad = L - int(v_idx.numel())
        if pad <= 0:
            sel_rows.append(v_idx[:L]); valid_rows.append(torch.ones(L, dtype=torch.long))
        else:
            reps = nv_idx.repeat((pad + nv_idx.numel() - 1) // nv_idx.numel())[:pad]
            sel_rows.append(torch.cat([v_idx, reps]))
            valid_rows.append(torch.cat([torch.ones(v_idx.numel(), dtype=torch.long), torch.zeros(pad, dtype=torch.long)]))
    sel = torch.stack(sel_rows).numpy().astype(np.int32)
    valid = torch.stack(valid_rows).numpy().astype(np.int32)
    hw = torch.stack([x_h, x_w], dim=1).numpy().astype(np.int32)
    return sel, valid, hw, L


def vilt_embed(P, spec, text_in, attention_mask, token_type_ids, pixel_values, taps=None, pixel_mask=None,
               image_type_idx: int = 1, image_embeds: Optional[torch.Tensor] = None):
    """text_in: LM output [B,T,H] (inputs_embeds) or int64 ids [B,T] when no LM is used."""
    v = spec.vilt
    B = attention_mask.shape[0]
    T = attention_mask.shape[1]
    if token_type_ids is None:
        token_type_ids = torch.zeros((B, T), dtype=torch.long)
    if text_in.dtype in (torch.int64, torch.int32):
        e = P["embeddings.text_embeddings.word_embeddings.weight"][text_in]
        use_pos = True
    else:
        e = text_in
        use_pos = spec.use_vilt_position_embeddings or spec.lm is None
    e = e + P["embeddings.text_embeddings.token_type_embeddings.weight"][token_type_ids]
    if use_pos:
        e = e + P["embeddings.text_embeddings.position_embeddings.weight"][:T].unsqueeze(0)
    e = _ln(e, P["embeddings.text_embeddings.LayerNorm.weight"], P["embeddings.text_embeddings.LayerNorm.bias"],
            v.layer_norm_eps)
    mt = P["embeddings.token_type_embeddings.weight"]
    text = e + mt[0]
    if image_embeds is not None:
        img = image_embeds + mt[image_type_idx]
        img_mask = (torch.ones(img.shape[:2], dtype=attention_mask.dtype) if pixel_mask is None
                    else pixel_mask.flatten(1).to(attention_mask.dtype))
        return torch.cat([text, img], dim=1), torch.cat([attention_mask, img_mask], dim=1)
    if _EMULATE_BWD:
        # the same convolution as an unfold + Linear (the kernels' im2col GEMM): bf16 dY / patch operands in its weight gradient
        ps = v.patch_size
        gh, gw = pixel_values.shape[2] // ps, pixel_values.shape[3] // ps
        cols = F.unfold(pixel_values, kernel_size=ps, stride=ps).transpose(1, 2)          # [B, gh*gw, C*ps*ps]
        wp = P["embeddings.patch_embeddings.projection.weight"]
        pe = _lin(cols, wp.reshape(wp.shape[0], -1), P["embeddings.patch_embeddings.projection.bias"])
    else:
        pe = F.conv2d(_r(pixel_values), _r(P["embeddings.patch_embeddings.projection.weight"]),
                      P["embeddings.patch_embeddings.projection.bias"], stride=v.patch_size)
        gh, gw = pe.shape[2], pe.shape[3]
        pe = pe.flatten(2).transpose(1, 2)                      # [B, gh*gw, H], row-major patch order
    pos = P["embeddings.position_embeddings"]               # [1, 1+g*g, H]
    g = v.image_size // v.patch_size
    full = pixel_mask is None or (gh == g and gw == g and bool((pixel_mask != 0).all()))
    if full:
        img = torch.cat([P["embeddings.cls_token"].expand(B, -1, -1), pe], dim=1) + pos
        img_mask = torch.ones((B, img.shape[1]), dtype=attention_mask.dtype)
    else:
        # padded / non-square images (HF:models/vilt/modeling_vilt.py:92-178): per-image bilinear resize of the
        # g x g position table to the image's own h x w patch grid (align_corners=True, zero outside), patch
        # selection by _select_patches above (the deterministic member of the reference's random outcomes)
        sel, valid, hw, L = _select_patches(pixel_mask, gh, gw)
        spatial = pos[:, 1:, :].transpose(1, 2).reshape(1, -1, g, g)
        rows = []
        for b in range(B):
            h, w = int(hw[b, 0]), int(hw[b, 1])
            pr = F.interpolate(spatial, size=(h, w), mode="bilinear", align_corners=True)
            pr = F.pad(pr, (0, gw - w, 0, gh - h)).flatten(2).transpose(1, 2)[0]      # [gh*gw, H]
            idx = torch.from_numpy(sel[b].astype(np.int64))
            rows.append(pe[b][idx] + pr[idx])
        patches = torch.stack(rows, dim=0)                                              # [B, L, H]
        cls = P["embeddings.cls_token"].expand(B, -1, -1) + pos[:, :1, :]
        img = torch.cat([cls, patches], dim=1)
        img_mask = torch.cat([torch.ones((B, 1), dtype=attention_mask.dtype),
                              torch.from_numpy(valid.astype(np.int64)).to(attention_mask.dtype)], dim=1)
    img = img + mt[image_type_idx]
    x = torch.cat([text, img], dim=1)
    if taps is not None:
        taps["vilt_embed"] = x
        taps["image_mask"] = img_mask
    mask = torch.cat([attention_mask, img_mask], dim=1)
    return x, mask


def vilt_encoder(P, spec, x, mask, taps=None):
    v = spec.vilt
    mask_add = (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for i in range(v.num_hidden_layers):
        pre = f"encoder.layer.{i}"
        if i > 0:
            x = _grad_bf16(x)      # (the bottom layer also writes its input gradient in f32, for the embedding backward)
        n1 = _ln(x, P[f"{pre}.layernorm_before.weight"], P[f"{pre}.layernorm_before.bias"], v.layer_norm_eps)
        c = _mha(n1, mask_add, P, pre, v.num_attention_heads, "attention.attention")
        a = _lin(c, P[f"{pre}.attention.output.dense.weight"], P[f"{pre}.attention.output.dense.bias"])
        x = _grad_bf16(a + x)
        n2 = _ln(x, P[f"{pre}.layernorm_after.weight"], P[f"{pre}.layernorm_after.bias"], v.layer_norm_eps)
        x = _ffn(n2, P[f"{pre}.intermediate.dense.weight"], P[f"{pre}.intermediate.dense.bias"],
                 P[f"{pre}.output.dense.weight"], P[f"{pre}.output.dense.bias"], u8=_EMULATE_GELU8) + x
        if taps is not None:
            taps[f"vilt_layer{i}"] = x
    return _grad_bf16(x)


def vault_forward(P, spec, batch: Dict[str, torch.Tensor], taps: Optional[dict] = None,
                  classifier_keep_mask: Optional[torch.Tensor] = None, classifier_p: float = 0.0):
    """Returns dict(last_hidden_state, pooler_output[, logits]).  Eval-mode arithmetic
    (all dropouts off) unless ``classifier_keep_mask`` is given (train-mode head dropout
    with an explicit mask so a GPU run with the same mask can be compared)."""
    ids = batch.get("input_ids")
    am = batch["attention_mask"]
    tt = batch.get("token_type_ids")
    pix = batch.get("pixel_values")
    temb = batch.get("inputs_embeds")
    if spec.lm is not None:
        text_in = lm_forward(P, spec, ids, am, tt, taps, inputs_embeds=temb)
    else:
        text_in = ids if temb is None else temb
    if batch.get("image_embeds") is not None:
        # HF ViltEmbeddings.forward with image_embeds (modeling_vilt.py:190-207): no patch projection / CLS / position
        # table; image_masks = pixel_mask.flatten(1); only the modality type is added
        x, mask = vilt_embed(P, spec, text_in, am, tt, None, taps, pixel_mask=batch.get("pixel_mask"),
                             image_embeds=batch["image_embeds"])
        x = vilt_encoder(P, spec, x, mask, taps)
        x = _ln(x, P["layernorm.weight"], P["layernorm.bias"], spec.vilt.layer_norm_eps)
        out = {"last_hidden_state": x}
        if spec.add_pooling_layer:
            out["pooler_output"] = torch.tanh(_lin(x[:, 0], P["pooler.dense.weight"], P["pooler.dense.bias"]))
            if spec.n_classes > 0 and getattr(spec, "head", "linear") == "linear":
                out["logits"] = F.linear(out["pooler_output"], P["classifier.1.weight"], P["classifier.1.bias"]).squeeze(-1)
        return out
    if getattr(spec, "num_images", 1) > 1:
        # HF ViltForImagesAndTextClassification.forward (modeling_vilt.py): one encoder pass per image with modality
        # type i + 1, pooled outputs concatenated, MLP classifier; the text goes through the LM once (ref
        # VaultMixin.lm_preprocess, model.py:151-202)
        pooled = []
        pm = batch.get("pixel_mask")
        for i in range(spec.num_images):
            x, mask = vilt_embed(P, spec, text_in, am, tt, pix[:, i], None, pixel_mask=None if pm is None else pm[:, i],
                                 image_type_idx=i + 1)
            x = vilt_encoder(P, spec, x, mask, None)
            x = _ln(x, P["layernorm.weight"], P["layernorm.bias"], spec.vilt.layer_norm_eps)
            pooled.append(torch.tanh(_lin(x[:, 0], P["pooler.dense.weight"], P["pooler.dense.bias"])))
        z = torch.cat(pooled, dim=-1)
        h = _lin(z, P["classifier.0.weight"], P["classifier.0.bias"])
        h = _ln(h, P["classifier.1.weight"], P["classifier.1.bias"], 1e-5)
        return {"pooler_output": z, "logits": _lin(F.gelu(_r(h)), P["classifier.3.weight"], P["classifier.3.bias"])}
    x, mask = vilt_embed(P, spec, text_in, am, tt, pix, taps, pixel_mask=batch.get("pixel_mask"))
    x = vilt_encoder(P, spec, x, mask, taps)
    x = _ln(x, P["layernorm.weight"], P["layernorm.bias"], spec.vilt.layer_norm_eps)
    out = {"last_hidden_state": x}
    if spec.add_pooling_layer:
        pooled = torch.tanh(_lin(x[:, 0], P["pooler.dense.weight"], P["pooler.dense.bias"]))
        out["pooler_output"] = pooled
        if getattr(spec, "head", "linear") == "mlm":
            # HF ViltMLMHead on the text rows: dense - GELU - LayerNorm - decoder (tied to ViLT's word embeddings) + bias
            T = ids.shape[1]
            t = F.gelu(_r(_lin(x[:, :T], P["mlm_score.transform.dense.weight"], P["mlm_score.transform.dense.bias"])))
            t = _ln(t, P["mlm_score.transform.LayerNorm.weight"], P["mlm_score.transform.LayerNorm.bias"],
                    spec.vilt.layer_norm_eps)
            out["logits"] = _lin(t, P["embeddings.text_embeddings.word_embeddings.weight"], P["mlm_score.bias"])
        elif spec.n_classes > 0 and getattr(spec, "head", "linear") == "mlp":
            # HF ViltForQuestionAnswering.classifier (modeling_vilt.py): Linear(H, 2H) - LayerNorm(2H) - GELU - Linear
            h = _lin(pooled, P["classifier.0.weight"], P["classifier.0.bias"])
            h = _ln(h, P["classifier.1.weight"], P["classifier.1.bias"], 1e-5)
            out["logits"] = _lin(F.gelu(_r(h)), P["classifier.3.weight"], P["classifier.3.bias"])
        elif spec.n_classes > 0:
            z = pooled
            if classifier_keep_mask is not None:
                z = z * classifier_keep_mask / (1.0 - classifier_p)
            out["logits"] = F.linear(z, P["classifier.1.weight"], P["classifier.1.bias"]).squeeze(-1)
    return out


def vault_loss(P, spec, batch, **kw):
    out = vault_forward(P, spec, batch, **kw)
    if batch["labels"].dtype.is_floating_point:
        # single-logit fine-tune (Bloomberg): logits.squeeze(-1) (ref: models/vault/model.py:569) under
        # nn.BCEWithLogitsLoss (ref: models/vault/trainer.py:55-56)
        loss = F.binary_cross_entropy_with_logits(out["logits"].squeeze(-1), batch["labels"])
    else:
        loss = F.cross_entropy(out["logits"], batch["labels"])
    return loss, out


def torch_batch(batch: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: _t(v) for k, v in batch.items()}


# --------------------------------------------------------------------------------------
# optimizer / schedule (formula restatement, float64 capable)
# --------------------------------------------------------------------------------------
def hf_adamw_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                  correct_bias=False):
    """One transformers-4.48 ``AdamW.step`` on arrays (in place on p, m, v; returns them).

    exp_avg.mul_(b1).add_(g, alpha=1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2)
    denom = exp_avg_sq.sqrt().add_(eps); step_size = lr [* sqrt(1-b2^t)/(1-b1^t)]
    p.addcdiv_(exp_avg, denom, value=-step_size); then p.add_(p, alpha=-lr*wd) if wd > 0.
    """
    m *= beta1
    m += (1.0 - beta1) * g
    v *= beta2
    v += (1.0 - beta2) * g * g
    denom = np.sqrt(v) + eps if isinstance(v, np.ndarray) else v.sqrt() + eps
    step_size = lr
    if correct_bias:
        step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p -= step_size * (m / denom)
    if weight_decay > 0.0:
        p -= lr * weight_decay * p
    return p, m, v


def linear_schedule_lr(base_lr: float, step: int, warmup_steps: int, total_steps: int) -> float:
    """``get_linear_schedule_with_warmup`` multiplier x base lr; ``step`` = number of
    scheduler.step() calls so far (the lr used by optimizer step number ``step``, 0-based)."""
    if step < warmup_steps:
        return base_lr * float(step) / float(max(1, warmup_steps))
    return base_lr * max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))
