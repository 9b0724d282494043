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
