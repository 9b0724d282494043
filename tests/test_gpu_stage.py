"""GPU parity of the stage-level C ABI (include/vault_hip.h: vault_vilt_layer_fwd/bwd, vault_lm_layer_fwd/bwd): ONE C call
per encoder layer forward / backward, bound through ctypes exactly as a foreign host would, against the fp32 oracle's
layer arithmetic (HF ViltLayer, modeling_vilt.py:430-451; RobertaLayer, modeling_roberta.py:421-463) and autograd."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import vault_oracle as O
from vault_amd import lib as L, ops
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, build_state

pytestmark = pytest.mark.gpu


def _pad(n, m=256):
    return (n + m - 1) // m * m


def _call(name, args):
    fn = getattr(L.load(), name)
    L.check(fn(C.byref(args), C.c_void_p(torch.cuda.current_stream().cuda_stream)), name)


def _layer_weights(state, pre, att, dev):
    w = lambda n: torch.from_numpy(state[f"{pre}.{n}"]).to(dev)  # noqa: E731
    qkv_w = torch.cat([w(f"{att}.query.weight"), w(f"{att}.key.weight"), w(f"{att}.value.weight")], 0)
    qkv_b = torch.cat([w(f"{att}.query.bias"), w(f"{att}.key.bias"), w(f"{att}.value.bias")], 0)
    return dict(wqkv=qkv_w.bfloat16().contiguous(), bqkv=qkv_b.contiguous(),
                wo=w("attention.output.dense.weight").bfloat16(), bo=w("attention.output.dense.bias"),
                wi=w("intermediate.dense.weight").bfloat16(), bi=w("intermediate.dense.bias"),
                wf=w("output.dense.weight").bfloat16(), bf=w("output.dense.bias"))


@pytest.mark.parametrize("transposed", [False, True])
def test_vilt_layer_one_c_call_forward_and_backward(transposed):
    spec = VaultSpec(vilt=ViltSpec(hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512),
                     lm=None, n_classes=0)
    v = spec.vilt
    state = build_state(spec, 3)
    B, S, H, FF, heads = 3, 185, v.hidden_size, v.intermediate_size, v.num_attention_heads
    M, Mp = B * S, _pad(B * S)
    dev = "cuda"
    pre = "encoder.layer.0"
    W = _layer_weights(state, pre, "attention.attention", dev)
    lnp = {k: torch.from_numpy(state[f"{pre}.{n}"]).to(dev) for k, n in
           (("ln1w", "layernorm_before.weight"), ("ln1b", "layernorm_before.bias"), ("ln2w", "layernorm_after.weight"),
            ("ln2b", "layernorm_after.bias"))}
    g = torch.Generator().manual_seed(0)
    x = torch.zeros(Mp, H); x[:M] = torch.randn(M, H, generator=g)
    keymask = torch.ones(B, S); keymask[1, 20:33] = 0; keymask[2, S - 11:] = 0
    x_d, km_d = x.to(dev), keymask.to(dev)
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
    bfl = torch.bfloat16
    bufs = dict(x_out=z(Mp, H), n1=z(Mp, H, dt=bfl), qkv=z(Mp, 3 * H, dt=bfl), ctx=z(Mp, H, dt=bfl), lse=z(B, heads, S),
                xm=z(Mp, H), n2=z(Mp, H, dt=bfl), act=z(Mp, FF, dt=bfl), u=z(Mp, FF, dt=bfl), m1=z(Mp), r1=z(Mp), m2=z(Mp),
                r2=z(Mp))
    extra = {}
    if transposed:
        extra = dict(wo_t=W["wo"].t().contiguous(), wf_t=W["wf"].t().contiguous())
    need = L.load().vault_layer_workspace_bytes
    need.restype = C.c_longlong
    rp = C.c_longlong(0)
    nbytes = need(B, S, H, FF, heads, 1, C.byref(rp))
    assert rp.value == Mp and nbytes >= sum(t.numel() * t.element_size() for t in bufs.values())
    a = ops.layer_args(B=B, S=S, H=H, FF=FF, heads=heads, rows=M, rows_pad=Mp, eps=v.layer_norm_eps, x_in=x_d, keymask=km_d,
                       **W, **lnp, **bufs, **extra)
    _call("vault_vilt_layer_fwd", a)
    # ---- oracle: the same layer in fp32 with autograd
    P = O.to_torch_state(state, requires_grad=True)
    xr = x[:M].view(B, S, H).clone().requires_grad_(True)
    y = O.vilt_encoder(P, spec, xr, keymask.long())
    torch.cuda.synchronize()
    out = bufs["x_out"][:M].view(B, S, H).cpu()
    valid = keymask.bool()
    assert (out[valid] - y.detach()[valid]).abs().max().item() < 2e-2 * y.detach().abs().max().item()
    assert float(bufs["x_out"][M:].abs().max()) == 0.0                          # pad rows stay zero
    # ---- backward: gradient at the output, valid rows only (masked keys' rows are garbage in, never read out)
    dy = torch.zeros(Mp, H); dy[:M] = torch.randn(M, H, generator=g) * valid.view(M, 1)
    dy_d = dy.to(dev)
    grads = {k: z(*W[k if k != "g_bf_below" else "bo"].shape) for k in ()}
    gw = dict(g_wqkv=z(3 * H, H), g_bqkv=z(3 * H), g_wo=z(H, H), g_bo=z(H), g_wi=z(FF, H), g_bi=z(FF), g_wf=z(H, FF),
              g_ln1w=z(H), g_ln1b=z(H), g_ln2w=z(H), g_ln2b=z(H), g_bf_below=z(H))
    sc = dict(dx_f32=z(Mp, H), dx_bf16=z(Mp, H, dt=bfl), dU=z(Mp, FF, dt=bfl), dN=z(Mp, H, dt=bfl), dctx=z(Mp, H, dt=bfl),
              dqkv=z(Mp, 3 * H, dt=bfl), dmid_bf16=z(Mp, H, dt=bfl), dmid_f32=z(Mp, H))
    gb = ops.layer_bwd_args(a, dy_bf16=dy_d.bfloat16(), dy_f32=dy_d, do_wgrad=1, **gw, **sc)
    _call("vault_vilt_layer_bwd", gb)
    y.backward(dy[:M].view(B, S, H))
    torch.cuda.synchronize()
    dxr = xr.grad[valid]
    dxm = sc["dx_f32"][:M].view(B, S, H).cpu()[valid]
    assert (dxm - dxr).norm() < 2e-2 * dxr.norm()
    att = f"{pre}.attention.attention"
    ref_qkv = torch.cat([P[f"{att}.query.weight"].grad, P[f"{att}.key.weight"].grad, P[f"{att}.value.weight"].grad], 0)
    for mine, ref in ((gw["g_wqkv"], ref_qkv), (gw["g_wo"], P[f"{pre}.attention.output.dense.weight"].grad),
                      (gw["g_wi"], P[f"{pre}.intermediate.dense.weight"].grad), (gw["g_wf"], P[f"{pre}.output.dense.weight"].grad),
                      (gw["g_bi"], P[f"{pre}.intermediate.dense.bias"].grad), (gw["g_bo"], P[f"{pre}.attention.output.dense.bias"].grad),
                      (gw["g_ln1w"], P[f"{pre}.layernorm_before.weight"].grad), (gw["g_ln2b"], P[f"{pre}.layernorm_after.bias"].grad)):
        assert (mine.cpu() - ref).norm() < 3e-2 * ref.norm() + 1e-6
    # column sums of the gradient at the layer input = bias gradient of the FFN-out below
    assert (gw["g_bf_below"].cpu() - sc["dx_bf16"][:M].float().sum(0).cpu()).abs().max().item() < 2e-2 * float(gw["g_bf_below"].abs().max()) + 1e-3


def test_lm_layer_one_c_call_forward_and_backward():
    lm = LMSpec(vocab_size=64, max_position_embeddings=50, hidden_size=256, num_hidden_layers=1, num_attention_heads=4,
                intermediate_size=512, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    spec = VaultSpec(vilt=ViltSpec(hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
                                   vocab_size=64), lm=lm, n_classes=0)
    state = build_state(spec, 5)
    B, S, H, FF, heads = 5, 40, 256, 512, 4
    M, Mp = B * S, _pad(B * S)
    dev = "cuda"
    pre = "bert.encoder.layer.0"
    W = _layer_weights(state, pre, "attention.self", dev)
    lnp = {k: torch.from_numpy(state[f"{pre}.{n}"]).to(dev) for k, n in
           (("ln1w", "attention.output.LayerNorm.weight"), ("ln1b", "attention.output.LayerNorm.bias"),
            ("ln2w", "output.LayerNorm.weight"), ("ln2b", "output.LayerNorm.bias"))}
    g = torch.Generator().manual_seed(1)
    x = torch.zeros(Mp, H); x[:M] = torch.randn(M, H, generator=g)
    keymask = torch.ones(B, S); keymask[0, 25:] = 0; keymask[3, 9:] = 0
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
    bfl = torch.bfloat16
    x_d = x.to(dev)
    bufs = dict(x_out=z(Mp, H), x_out_bf16=z(Mp, H, dt=bfl), qkv=z(Mp, 3 * H, dt=bfl), ctx=z(Mp, H, dt=bfl), lse=z(B, heads, S),
                xm=z(Mp, H), y1=z(Mp, H), n2=z(Mp, H, dt=bfl), act=z(Mp, FF, dt=bfl), u=z(Mp, FF, dt=bfl), h2=z(Mp, H),
                m1=z(Mp), r1=z(Mp), m2=z(Mp), r2=z(Mp))
    a = ops.layer_args(B=B, S=S, H=H, FF=FF, heads=heads, rows=M, rows_pad=Mp, eps=lm.layer_norm_eps, x_in=x_d,
                       x_in_bf16=x_d.bfloat16(), keymask=keymask.to(dev), **W, **lnp, **bufs)
    _call("vault_lm_layer_fwd", a)
    # ---- oracle: one post-LN layer = lm_forward's loop body
    P = O.to_torch_state(state, requires_grad=True)
    xr = x[:M].view(B, S, H).clone().requires_grad_(True)
    mask_add = (1.0 - keymask[:, None, None, :]) * torch.finfo(torch.float32).min
    c = O._mha(xr, mask_add, P, pre, heads, "attention.self")
    a1 = O._lin(c, P[f"{pre}.attention.output.dense.weight"], P[f"{pre}.attention.output.dense.bias"])
    y1 = O._ln(a1 + xr, P[f"{pre}.attention.output.LayerNorm.weight"], P[f"{pre}.attention.output.LayerNorm.bias"], lm.layer_norm_eps)
    h = torch.nn.functional.gelu(O._lin(y1, P[f"{pre}.intermediate.dense.weight"], P[f"{pre}.intermediate.dense.bias"]))
    o = O._lin(h, P[f"{pre}.output.dense.weight"], P[f"{pre}.output.dense.bias"])
    y = O._ln(o + y1, P[f"{pre}.output.LayerNorm.weight"], P[f"{pre}.output.LayerNorm.bias"], lm.layer_norm_eps)
    torch.cuda.synchronize()
    valid = keymask.bool()
    out = bufs["x_out"][:M].view(B, S, H).cpu()
    assert (out[valid] - y.detach()[valid]).abs().max().item() < 2e-2 * y.detach().abs().max().item()
    assert torch.equal(bufs["x_out_bf16"][:M], bufs["x_out"][:M].bfloat16())
    dy = torch.zeros(Mp, H); dy[:M] = torch.randn(M, H, generator=g) * valid.view(M, 1)
    gw = dict(g_wqkv=z(3 * H, H), g_bqkv=z(3 * H), g_wo=z(H, H), g_bo=z(H), g_wi=z(FF, H), g_bi=z(FF), g_wf=z(H, FF), g_bf=z(H),
              g_ln1w=z(H), g_ln1b=z(H), g_ln2w=z(H), g_ln2b=z(H))
    sc = dict(dx_f32=z(Mp, H), dx_bf16=z(Mp, H, dt=bfl), dU=z(Mp, FF, dt=bfl), dN=z(Mp, H, dt=bfl), dctx=z(Mp, H, dt=bfl),
              dqkv=z(Mp, 3 * H, dt=bfl), dmid_bf16=z(Mp, H, dt=bfl), dh1_bf16=z(Mp, H, dt=bfl), dmid_f32=z(Mp, H))
    gb = ops.layer_bwd_args(a, dy_f32=dy.to(dev), do_wgrad=1, **gw, **sc)
    _call("vault_lm_layer_bwd", gb)
    y.backward(dy[:M].view(B, S, H))
    torch.cuda.synchronize()
    dxm = (sc["dx_f32"][:M] + sc["dx_bf16"][:M].float()).view(B, S, H).cpu()[valid]     # the two parts of d y
    dxr = xr.grad[valid]
    assert (dxm - dxr).norm() < 2e-2 * dxr.norm()
    for mine, ref in ((gw["g_wo"], P[f"{pre}.attention.output.dense.weight"].grad), (gw["g_wi"], P[f"{pre}.intermediate.dense.weight"].grad),
                      (gw["g_wf"], P[f"{pre}.output.dense.weight"].grad), (gw["g_bf"], P[f"{pre}.output.dense.bias"].grad),
                      (gw["g_bo"], P[f"{pre}.attention.output.dense.bias"].grad), (gw["g_ln2w"], P[f"{pre}.output.LayerNorm.weight"].grad),
                      (gw["g_ln1b"], P[f"{pre}.attention.output.LayerNorm.bias"].grad)):
        assert (mine.cpu() - ref).norm() < 3e-2 * ref.norm() + 1e-6


def test_embedding_and_head_stages_one_c_call_each_match_the_engine():
    """ABI 5: vault_lm_embed / vault_vilt_text_embed / vault_patch_embed / vault_head_loss forward and backward, each ONE C
    call bound through ctypes, on the inputs and parameters of an engine pass whose op-by-op results are pinned against the
    oracle elsewhere (tests/test_gpu_model.py, test_gpu_train.py): forward outputs bit-identical (same kernels, same order),
    gradients equal up to the order of the float atomics."""
    from vault_amd.engine import VaultEngine
    from vault_amd.spec import synthetic_batch
    spec = VaultSpec.tiny(3, "bert")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    lm, v = spec.lm, spec.vilt
    B = 4
    bn = synthetic_batch(spec, B, seed=5, n_classes=3)
    bn["token_type_ids"] = (np.arange(bn["input_ids"].shape[1])[None, :] >= 20).astype(np.int64).repeat(B, 0)
    state = build_state(spec, 9)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    db = {k: torch.from_numpy(val).cuda() for k, val in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    out = eng.forward(db, train=True, labels=labels)
    eng.zero_grad(); eng.backward()
    torch.cuda.synchronize()
    ws, P = eng.last, eng.params
    T, S, H, NP = ws["T"], ws["S"], ws["H"], ws["NP"]
    Ml, Mlp, M, Mp = ws["Ml"], ws["Mlp"], ws["M"], ws["Mp"]
    nl, nv = lm.num_hidden_layers, v.num_hidden_layers
    dev = "cuda"
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
    bfl = torch.bfloat16

    def close(mine, ref, tol=2e-5):
        ref = ref.reshape(mine.shape).float()
        assert (mine.float() - ref).norm().item() <= tol * ref.norm().item() + 1e-9

    # ---- LM embeddings ------------------------------------------------------------------------------------------
    e = ops.stage_args(
        ops.LmEmbedArgs, B=B, T=T, H=H, rows_pad=Mlp, pos_mode=0, pad_id=lm.pad_token_id, eps=lm.layer_norm_eps, ids=ws["ids"],
        token_type_ids=ws["tt"], word=P.w("bert.embeddings.word_embeddings.weight"), pos=P.w("bert.embeddings.position_embeddings.weight"),
        type=P.w("bert.embeddings.token_type_embeddings.weight"), lnw=P.w("bert.embeddings.LayerNorm.weight"),
        lnb=P.w("bert.embeddings.LayerNorm.bias"), pos_ids=z(B, T, dt=torch.int32), esum=z(Mlp, H), mean=z(Mlp), rstd=z(Mlp),
        y=z(Mlp, H), y_bf16=z(Mlp, H, dt=bfl))
    _call("vault_lm_embed_fwd", e)
    torch.cuda.synchronize()
    y_mine, yb_mine = e._keep[-2], e._keep[-1]
    assert torch.equal(y_mine[:Ml], ws["lm_y"][0][:Ml]) and torch.equal(yb_mine[:Ml], ws["lm_yb"][0][:Ml])
    gE = dict(g_word=z(lm.vocab_size, H), g_pos=z(lm.max_position_embeddings, H), g_type=z(lm.type_vocab_size, H), g_lnw=z(H), g_lnb=z(H))
    desum = z(Mlp, H)
    for k, val in dict(dy_bf16=ws["lm_dN"], dy_f32=ws["lm_dh1"], desum=desum, rowmask=ws["amf"], **gE).items():
        setattr(e, k, val.data_ptr())
    _call("vault_lm_embed_bwd", e)
    torch.cuda.synchronize()
    assert torch.equal(desum[:Ml], ws["lm_desum"][:Ml])
    close(gE["g_word"], P.gr("bert.embeddings.word_embeddings.weight"))
    close(gE["g_pos"], P.gr("bert.embeddings.position_embeddings.weight"))
    close(gE["g_type"], P.gr("bert.embeddings.token_type_embeddings.weight"))
    close(gE["g_lnw"], P.gr("bert.embeddings.LayerNorm.weight")); close(gE["g_lnb"], P.gr("bert.embeddings.LayerNorm.bias"))

    # ---- ViLT text embeddings on the LM output -----------------------------------------------------------------
    x_mine = z(Mp, H)
    mt = P.w("embeddings.token_type_embeddings.weight")
    use_pos = ws["use_pos"]
    t = ops.stage_args(
        ops.TextEmbedArgs, B=B, T=T, S=S, H=H, rows_pad=Mlp, eps=v.layer_norm_eps, text_src=ws["lm_y"][nl], ids=ws["ids"],
        token_type_ids=ws["tt"], type=P.w("embeddings.text_embeddings.token_type_embeddings.weight"),
        lnw=P.w("embeddings.text_embeddings.LayerNorm.weight"), lnb=P.w("embeddings.text_embeddings.LayerNorm.bias"), mtype0=mt[0],
        vsum=z(Mlp, H), mean=z(Mlp), rstd=z(Mlp), x=x_mine,
        **(dict(pos=P.w("embeddings.text_embeddings.position_embeddings.weight")) if use_pos else {}))
    _call("vault_vilt_text_embed_fwd", t)
    # ---- patch embedding into the same fused sequence ----------------------------------------------------------
    Kp = v.num_channels * v.patch_size * v.patch_size
    wpn = "embeddings.patch_embeddings.projection.weight"
    pe = ops.stage_args(
        ops.PatchEmbedArgs, B=B, C=v.num_channels, IMG=v.image_size, ps=v.patch_size, T=T, S=S, H=H, pixel_values=ws["pix"],
        w_bf16=P.wb(wpn, shape=(H, Kp)), conv_bias=P.w("embeddings.patch_embeddings.projection.bias"),
        pos_emb=P.w("embeddings.position_embeddings"), mtype1=mt[1], cls=P.w("embeddings.cls_token"),
        apatch=z(_pad(B * NP), Kp, dt=bfl), addtab=z(NP, H), x=x_mine)
    _call("vault_patch_embed_fwd", pe)
    torch.cuda.synchronize()
    assert torch.equal(x_mine[:M], ws["x"][0][:M])           # text rows, CLS rows and patch rows of every item
    # backward of both over the engine's gradient at the fused sequence
    dx0 = ws["dx_a"]
    gT = dict(g_type=z(v.type_vocab_size, H), g_lnw=z(H), g_lnb=z(H), g_mtype0=z(H))
    if use_pos:
        gT["g_pos"] = z(v.max_position_embeddings, H)
    dvsum = z(Mlp, H)
    for k, val in dict(dx=dx0, dvsum=dvsum, dbeta_scratch=z(H), **gT).items():
        setattr(t, k, val.data_ptr())
    _call("vault_vilt_text_embed_bwd", t)
    gP = dict(g_w=z(H, Kp), g_conv_bias=z(H), g_pos=z(NP + 1, H), g_mtype1=z(H), g_cls=z(H))
    for k, val in dict(dx=dx0, dyp=z(_pad(B * NP), H, dt=bfl), **gP).items():
        setattr(pe, k, val.data_ptr())
    _call("vault_patch_embed_bwd", pe)
    torch.cuda.synchronize()
    assert torch.equal(dvsum[:Ml], ws["d_vt_sum"][:Ml])
    close(gT["g_type"], P.gr("embeddings.text_embeddings.token_type_embeddings.weight"))
    close(gT["g_lnw"], P.gr("embeddings.text_embeddings.LayerNorm.weight")); close(gT["g_lnb"], P.gr("embeddings.text_embeddings.LayerNorm.bias"))
    gmt = P.gr("embeddings.token_type_embeddings.weight")
    close(gT["g_mtype0"], gmt[0]); close(gP["g_mtype1"], gmt[1])
    close(gP["g_w"], P.gr(wpn)); close(gP["g_conv_bias"], P.gr("embeddings.patch_embeddings.projection.bias"))
    close(gP["g_pos"], P.gr("embeddings.position_embeddings")); close(gP["g_cls"], P.gr("embeddings.cls_token"))

    # ---- head + loss ---------------------------------------------------------------------------------------------
    Bp, Cn = _pad(B), spec.n_classes
    h = ops.stage_args(
        ops.HeadLossArgs, B=B, S=S, H=H, C=Cn, seq_rows_pad=Mp, eps=v.layer_norm_eps, x=ws["x"][nv], lnw=P.w("layernorm.weight"),
        lnb=P.w("layernorm.bias"), wp_bf16=P.wb("pooler.dense.weight"), bp=P.w("pooler.dense.bias"), Wc=P.w("classifier.1.weight"),
        bc=P.w("classifier.1.bias"), labels=ws["labels"], loss_kind=0, loss_scale=1.0 / B, grad_scale=1.0 / B,
        h0_bf16=z(Bp, H, dt=bfl), mean=z(Bp), rstd=z(Bp), pre=z(Bp, H), pooled=z(Bp, H), logits=z(B, Cn), loss=z(1) + 7.0)
    _call("vault_head_loss_fwd", h)
    torch.cuda.synchronize()
    pooled, logits, loss = h._keep[-3], h._keep[-2], h._keep[-1]
    assert torch.equal(logits, out["logits"]) and torch.equal(pooled[:B], out["pooler_output"])
    assert abs(float(loss) - float(out["loss"])) < 1e-6
    gH = dict(g_Wc=z(Cn, H), g_bc=z(Cn), g_wp=z(H, H), g_bp=z(H), g_lnw=z(H), g_lnb=z(H), g_bf_last=z(H))
    dxf, dxb = z(Mp, H) + 3.0, z(Mp, H, dt=bfl) + 3.0          # (the call zeroes them)
    for k, val in dict(dpre=z(Bp, H, dt=bfl), dh0=z(Bp, H, dt=bfl), dx_f32=dxf, dx_bf16=dxb, **gH).items():
        setattr(h, k, val.data_ptr())
    _call("vault_head_loss_bwd", h)
    torch.cuda.synchronize()
    close(gH["g_Wc"], P.gr("classifier.1.weight")); close(gH["g_bc"], P.gr("classifier.1.bias"))
    close(gH["g_wp"], P.gr("pooler.dense.weight")); close(gH["g_bp"], P.gr("pooler.dense.bias"))
    close(gH["g_lnw"], P.gr("layernorm.weight")); close(gH["g_lnb"], P.gr("layernorm.bias"))
    rows = torch.arange(B, device=dev) * S
    other = torch.ones(Mp, dtype=torch.bool, device=dev); other[rows] = False
    assert float(dxf[other].abs().max()) == 0.0 and float(dxf[rows].abs().max()) > 0.0
    assert torch.equal(dxb, dxf.bfloat16())

    # ---- workspace query ------------------------------------------------------------------------------------------
    need = lambda b, tr: ops.workspace_bytes(768, 3072, 12, 12, 12, 384, 32, 3, 3, b, 40, tr)  # noqa: E731
    layer = L.load().vault_layer_workspace_bytes
    layer.restype = C.c_longlong
    assert need(256, True) > 12 * layer(256, 185, 768, 3072, 12, 1, None) + 12 * layer(256, 40, 768, 3072, 12, 1, None)
    assert need(8, False) < need(8, True) < need(64, True) < need(256, True) < 288e9
    assert ops.workspace_bytes(768, 3072, 12, 12, 12, 384, 31, 3, 3, 8, 40, True) == -1
