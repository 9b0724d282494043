"""GPU parity of the LayerNorm / column-sum / attention kernels (through the C ABI) against plain
fp32 torch references of the same ops on the same (bf16-rounded) inputs."""
import math

import pytest
import torch

from vault_amd import ops

pytestmark = pytest.mark.gpu


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("H", [256, 768])
@pytest.mark.parametrize("eps", [1e-12, 1e-5])
def test_layernorm_fwd_bwd(H, eps):
    rows = 333
    x = _rand(rows, H, seed=1) * 2 + 0.3
    gam = 1 + 0.1 * _rand(H, seed=2)
    bet = 0.1 * _rand(H, seed=3)
    post = _rand(H, seed=4)
    y16 = torch.zeros(rows, H, dtype=torch.bfloat16, device="cuda")
    y32 = torch.zeros(rows, H, device="cuda")
    mean = torch.zeros(rows, device="cuda"); rstd = torch.zeros(rows, device="cuda")
    ops.layernorm_fwd(x, gam, bet, eps, rows, H, y_bf16=y16, y_f32=y32, mean=mean, rstd=rstd, post_add=post)
    xr = x.clone().requires_grad_(True)
    gr = gam.clone().requires_grad_(True)
    br = bet.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (H,), gr, br, eps) + post
    torch.cuda.synchronize()
    assert (y32 - ref).abs().max().item() < 1e-4
    assert (y16.float() - ref).abs().max().item() < ref.abs().max().item() * 2 ** -8
    # backward: dy = bf16 part + f32 part, plus residual gradient
    dy16 = _rand(rows, H, seed=5).bfloat16()
    dy32 = _rand(rows, H, seed=6)
    dres = _rand(rows, H, seed=7)
    dx32 = torch.zeros(rows, H, device="cuda"); dx16 = torch.zeros(rows, H, dtype=torch.bfloat16, device="cuda")
    dg = torch.zeros(H, device="cuda"); db = torch.zeros(H, device="cuda"); dbias = torch.ones(H, device="cuda")
    ops.layernorm_bwd(x, mean, rstd, gam, rows, H, dy_bf16=dy16, dy_f32=dy32, dres=dres, dx_f32=dx32, dx_bf16=dx16,
                      dgamma=dg, dbeta=db, dbias=dbias)
    ref.backward(dy16.float() + dy32)
    torch.cuda.synchronize()
    tol = 5e-5 * max(1.0, xr.grad.abs().max().item())
    assert (dx32 - (xr.grad + dres)).abs().max().item() < tol
    assert (dx16.float() - dx32).abs().max().item() <= dx32.abs().max().item() * 2 ** -8
    assert (dg - gr.grad).abs().max().item() < 2e-4 * gr.grad.abs().max().item() + 1e-4
    assert (db - br.grad).abs().max().item() < 2e-4 * br.grad.abs().max().item() + 1e-4
    # fused bias gradient of the Linear fed by the bf16 branch: column sums of dx
    assert (dbias - (1 + dx32.sum(0))).abs().max().item() < 2e-3


def test_layernorm_rowmaps():
    # read rows b*S + t (t < T) of a fused sequence, write compact
    B, S, T, H = 3, 185, 40, 256
    xfull = _rand(B * S, H, seed=8)
    gam = torch.ones(H, device="cuda"); bet = torch.zeros(H, device="cuda")
    y = torch.zeros(B * T, H, device="cuda")
    ops.layernorm_fwd(xfull, gam, bet, 1e-12, B * T, H, y_f32=y, xmap=(T, S, 0))
    ref = torch.nn.functional.layer_norm(xfull.view(B, S, H)[:, :T].reshape(B * T, H), (H,))
    torch.cuda.synchronize()
    assert (y - ref).abs().max().item() < 1e-4
    y2 = torch.zeros(B * S, H, device="cuda")
    ops.layernorm_fwd(ref, gam, bet, 1e-12, B * T, H, y_f32=y2, ymap=(T, S, 0))
    torch.cuda.synchronize()
    assert y2.view(B, S, H)[:, T:].abs().max().item() == 0.0
    assert (y2.view(B, S, H)[:, :T].reshape(B * T, H) - torch.nn.functional.layer_norm(ref, (H,))).abs().max().item() < 1e-4


def test_colsum():
    rows, N = 1000, 768
    x = _rand(1024, N, seed=9).bfloat16()
    out = torch.ones(N, device="cuda")
    ops.colsum(x, N, rows, N, out)
    torch.cuda.synchronize()
    ref = 1 + x[:rows].float().sum(0)
    assert (out - ref).abs().max().item() < 1e-3


def _attn_ref(qkv, keymask, B, S, H, heads):
    q, k, v = qkv.float().view(B, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) / 8.0
    s = s + (1.0 - keymask)[:, None, None, :] * torch.finfo(torch.float32).min
    p = torch.softmax(s, dim=-1)
    o = p @ v
    return o.permute(0, 2, 1, 3).reshape(B * S, H), s


# (S = 70 / 129 / 65: waves of the single-pass backward whose 16-row tile lies entirely behind the sequence - they issue no
#  stores, the other branch of its counted waits; S = 192 / 64: no padding rows at all; S = 17: a single tile)
@pytest.mark.parametrize("S,heads", [(185, 12), (40, 12), (185, 4), (33, 2), (70, 3), (129, 5), (192, 2), (65, 1), (64, 2), (17, 1),
                                     (193, 2), (281, 3), (288, 1), (289, 2), (320, 1)])     # 72 KiB and 80 KiB K / V images (padded image batches)
def test_attention_fwd_bwd(S, heads):
    B, H = 3, heads * 64
    M = B * S
    qkv = (_rand(M, 3 * H, seed=10) * 1.5).bfloat16()
    keymask = torch.ones(B, S, device="cuda")
    keymask[1, 5:min(17, S - 1)] = 0
    keymask[2, S - min(9, S - 2):] = 0
    ctx = torch.zeros(M, H, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, heads, S, device="cuda")
    ops.attention_fwd(qkv, keymask, ctx, lse, B, S, H, heads)
    qr = qkv.float().clone().requires_grad_(True)
    ref, s = _attn_ref(qr, keymask, B, S, H, heads)
    torch.cuda.synchronize()
    assert (ctx.float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item()
    lse_ref = torch.logsumexp(s, dim=-1)
    assert (lse - lse_ref).abs().max().item() < 1e-3
    # backward
    dctx = _rand(M, H, seed=11).bfloat16()
    dqkv = torch.zeros(M, 3 * H, dtype=torch.bfloat16, device="cuda")
    ops.attention_bwd(qkv, keymask, ctx, lse, dctx, dqkv, B, S, H, heads)
    ref.backward(dctx.float())
    torch.cuda.synchronize()
    g = qr.grad
    err = (dqkv.float() - g).abs().max().item()
    assert err < 3e-2 * g.abs().max().item(), (err, g.abs().max().item())
    # relative Frobenius error is the tighter, bf16-sized bound
    rel = ((dqkv.float() - g).norm() / g.norm()).item()
    assert rel < 1e-2, rel


def test_attention_dropout_consistency():
    """Dropout on the probabilities: forward and backward regenerate the same mask (finite-difference
    free check: d<ctx, dctx>/dV equals P_drop^T dctx, so dV from bwd must match a V-gradient computed by
    running fwd with one-hot perturbations -> instead use linearity in V: ctx(V1+V2) = ctx(V1)+ctx(V2))."""
    B, S, heads = 2, 40, 2
    H = heads * 64
    M = B * S
    qkv = _rand(M, 3 * H, seed=12).bfloat16()
    keymask = torch.ones(B, S, device="cuda")
    drop = ops.Drop(0.1, seed=7, stream=3)
    ctx = torch.zeros(M, H, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, heads, S, device="cuda")
    ops.attention_fwd(qkv, keymask, ctx, lse, B, S, H, heads, drop=drop)
    ctx2 = torch.zeros_like(ctx)
    ops.attention_fwd(qkv, keymask, ctx2, lse, B, S, H, heads, drop=drop)
    torch.cuda.synchronize()
    assert torch.equal(ctx, ctx2)  # deterministic mask
    ctx0 = torch.zeros_like(ctx)
    ops.attention_fwd(qkv, keymask, ctx0, lse, B, S, H, heads)
    torch.cuda.synchronize()
    assert not torch.equal(ctx, ctx0)
    # E[dropout(P)] = P: means agree loosely
    assert abs(ctx.float().mean().item() - ctx0.float().mean().item()) < 0.02
    # backward consistency: dV = P_drop^T dO  <=>  <dV, V'> = <dO, ctx(V')> for any V' (ctx is linear in V)
    dctx = _rand(M, H, seed=13).bfloat16()
    dqkv = torch.zeros(M, 3 * H, dtype=torch.bfloat16, device="cuda")
    ops.attention_bwd(qkv, keymask, ctx, lse, dctx, dqkv, B, S, H, heads, drop=drop)
    qkv2 = qkv.clone()
    vprime = _rand(M, H, seed=14).bfloat16()
    qkv2[:, 2 * H:] = vprime
    ctxp = torch.zeros_like(ctx)
    ops.attention_fwd(qkv2, keymask, ctxp, lse.clone(), B, S, H, heads, drop=drop)
    torch.cuda.synchronize()
    lhs = (dqkv[:, 2 * H:].float() * vprime.float()).sum().item()
    rhs = (dctx.float() * ctxp.float()).sum().item()
    assert abs(lhs - rhs) < 2e-2 * max(1.0, abs(rhs)), (lhs, rhs)


@pytest.mark.parametrize("B,S,heads", [(50, 185, 12), (64, 185, 12), (300, 40, 12), (100, 70, 12), (130, 24, 12), (60, 129, 7)])
def test_attention_fwd_bwd_many_items(B, S, heads):
    """More (batch, head) items than the 256 persistent workgroups of the resident attention backward (B = 50 x 12 heads =
    600, B = 64: 768 = three full rounds): every workgroup walks several items with the next item's operands prefetched
    under the current one's arithmetic - the code path of the B = 256 bench (3,072 items).  Key masks on a few samples,
    three runs (a race in the item hand-off shows as run-to-run differences)."""
    H = heads * 64
    M = B * S
    Mp = ((M + 255) // 256) * 256
    qkv = torch.zeros(Mp, 3 * H, dtype=torch.bfloat16, device="cuda")
    qkv[:M] = (_rand(M, 3 * H, seed=20) * 1.5).bfloat16()
    keymask = torch.ones(B, S, device="cuda")
    g = torch.Generator().manual_seed(5)
    for b in range(0, B, 7):
        n = int(torch.randint(1, S // 2, (1,), generator=g))
        keymask[b, S - n:] = 0
    keymask[3, 5:17] = 0
    ctx = torch.zeros(Mp, H, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, heads, S, device="cuda")
    ops.attention_fwd(qkv, keymask, ctx, lse, B, S, H, heads)
    qr = qkv[:M].float().clone().requires_grad_(True)
    ref, s = _attn_ref(qr, keymask, B, S, H, heads)
    torch.cuda.synchronize()
    assert (ctx[:M].float() - ref).abs().max().item() < 2e-2 * ref.abs().max().item()
    assert (lse - torch.logsumexp(s, dim=-1)).abs().max().item() < 1e-3
    dctx = torch.zeros(Mp, H, dtype=torch.bfloat16, device="cuda")
    dctx[:M] = _rand(M, H, seed=21).bfloat16()
    ref.backward(dctx[:M].float())
    gref = qr.grad
    first = None
    for rep in range(3):
        dqkv = torch.zeros(Mp, 3 * H, dtype=torch.bfloat16, device="cuda")
        ops.attention_bwd(qkv, keymask, ctx, lse, dctx, dqkv, B, S, H, heads)
        torch.cuda.synchronize()
        d = dqkv[:M].float()
        assert (d - gref).abs().max().item() < 3e-2 * gref.abs().max().item()
        # per-sample relative error: one wrong item among hundreds must not hide in a global norm
        per = ((d - gref).view(B, -1).norm(dim=1) / gref.view(B, -1).norm(dim=1)).max().item()
        assert per < 1.5e-2, per
        if first is None:
            first = dqkv.clone()
        else:
            assert torch.equal(dqkv, first)


@pytest.mark.parametrize("rows,H,tt_kind", [(2560, 768, "zeros"), (333, 256, "mixed"), (40, 768, "ones"), (1000, 512, "int32")])
def test_scatter_add_embedding_gradients(rows, H, tt_kind):
    """Embedding-table gradients (ref: the autograd backward of nn.Embedding in HF modeling_roberta.py:86-120 /
    modeling_vilt.py:249-262): word rows by id, positions by row % period, token types by an index that is the same for
    (nearly) every row - the kernel sums rows that repeat the first index of their block in registers; masked rows add
    nothing; the tables ACCUMULATE."""
    g = torch.Generator(device="cpu").manual_seed(rows)
    d = torch.randn(rows, H, generator=g).cuda()
    V, period = 500, 40
    ids = torch.randint(0, V, (rows,), generator=g).cuda()
    ids[: rows // 3] = 7                                           # long runs of one id, too
    if tt_kind == "zeros":
        tt = torch.zeros(rows, dtype=torch.int64, device="cuda")
    elif tt_kind == "ones":
        tt = torch.ones(rows, dtype=torch.int64, device="cuda")
    elif tt_kind == "int32":
        tt = (torch.arange(rows, device="cuda") % 2).to(torch.int32)
    else:
        tt = (torch.rand(rows, generator=g) < 0.2).long().cuda()
    mask = (torch.rand(rows, generator=g) < 0.8).float().cuda()
    for rowmask in (None, mask):
        word = torch.full((V, H), 0.5, device="cuda"); pos = torch.full((period, H), -0.25, device="cuda")
        typ = torch.full((2, H), 1.0, device="cuda")
        ops.scatter_add(d, [(word, ids), (pos, "mod"), (typ, tt)], rows, H, period=period, rowmask=rowmask)
        dm = d if rowmask is None else d * rowmask[:, None]
        rw = torch.full((V, H), 0.5, device="cuda").index_add_(0, ids, dm)
        rp = torch.full((period, H), -0.25, device="cuda").index_add_(0, torch.arange(rows, device="cuda") % period, dm)
        rt = torch.full((2, H), 1.0, device="cuda").index_add_(0, tt.long(), dm)
        torch.cuda.synchronize()
        for got, ref in ((word, rw), (pos, rp), (typ, rt)):
            assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    # a fixed destination row and an absent table
    typ = torch.zeros(2, H, device="cuda")
    ops.scatter_add(d, [None, None, (typ, 1)], rows, H)
    torch.cuda.synchronize()
    assert (typ[1] - d.sum(0)).abs().max().item() <= 1e-4 * d.sum(0).abs().max().item() and float(typ[0].abs().max()) == 0.0


@pytest.mark.parametrize("rows,ld,N,batch", [(2560, 2304, 2304, 6), (11840, 2304, 768, 6), (300, 768, 256, 1), (47360, 2304, 768, 2)])
def test_colsum_batched_bias_gradients_of_a_group_of_layers(rows, ld, N, batch):
    """The QKV bias gradients of `batch` layers in one launch (db = sum over token rows of dY, ref: autograd of nn.Linear's
    bias, HF modeling_vilt.py:303-313): matrix z of a stack sums into vector z at a uniform stride of the flat gradient
    buffer, accumulating; rows beyond `rows` and columns beyond N are not read."""
    pad = ((rows + 255) // 256) * 256
    x = torch.randn(batch, pad, ld, device="cuda").bfloat16()
    stride_o = 4096
    out = torch.full((batch * stride_o,), 0.25, device="cuda")
    ops.colsum_batched(x[0], ld, rows, N, out, batch, x.stride(0), stride_o)
    ref = x[:, :rows, :N].float().sum(1)
    torch.cuda.synchronize()
    got = out.view(batch, stride_o)
    assert (got[:, :N] - 0.25 - ref).abs().max().item() <= 2e-5 * rows ** 0.5 * ref.abs().max().item() + 1e-3
    assert float((got[:, N:] - 0.25).abs().max()) == 0.0
    single = torch.zeros(N, device="cuda")
    ops.colsum(x[batch - 1], ld, rows, N, single)
    torch.cuda.synchronize()
    assert (single - ref[batch - 1]).abs().max().item() <= 2e-5 * rows ** 0.5 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("n,offset", [(4096, 0), (12, 0), (9, 0), (1027, 1), (3, 2), (1 << 20, 3)])
def test_scale_is_exact_for_powers_of_two_at_any_length_and_alignment(n, offset):
    """vault_scale_f32 (the fp16 build's gradient scale in and out of the flat f32 gradient buffer): x * 4096 / 4096 is the
    identity bit for bit, for 16-byte-aligned middles and ragged heads / tails alike; the neighbours are untouched."""
    buf = torch.randn(n + offset + 5, device="cuda")
    x = buf[offset:offset + n]
    want = x.clone()
    guard = (buf[:offset].clone(), buf[offset + n:].clone())
    ops.scale(x, 4096.0, n)
    torch.cuda.synchronize()
    assert torch.equal(x, want * 4096.0)
    ops.scale(x, 1.0 / 4096.0, n)
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    assert torch.equal(buf[:offset], guard[0]) and torch.equal(buf[offset + n:], guard[1])


def _to_head_major(x, R):
    """[M, N] -> [N / 64, R, 64] (R >= M rows per plane, zero padded): the head-major layout of qkv / dqkv."""
    M, N = x.shape
    out = torch.zeros(N // 64, R, 64, dtype=x.dtype, device=x.device)
    out[:, :M] = x.view(M, N // 64, 64).permute(1, 0, 2)
    return out


def _from_head_major(y, M):
    P, R, _ = y.shape
    return y[:, :M].permute(1, 0, 2).reshape(M, P * 64)


@pytest.mark.parametrize("B,S,heads,drop_p", [(3, 185, 12, 0.0), (50, 185, 12, 0.0), (9, 40, 12, 0.1), (300, 40, 12, 0.0), (5, 129, 5, 0.0),
                                              (4, 64, 2, 0.0), (2, 17, 1, 0.0)])
def test_attention_head_major_layout_is_bit_identical(B, S, heads, drop_p):
    """``qkv_hm``: qkv / dqkv as [3][heads][R][64] (a (batch, head) item's rows contiguous) instead of [tokens][3H]: the same
    arithmetic on other addresses - context, log-sum-exp and the gradient (brought back to row-major) equal the row-major
    run bit for bit, also with more items than workgroups and with dropout."""
    H = heads * 64
    M = B * S
    R = ((M + 255) // 256) * 256 + 256
    qkv = (_rand(M, 3 * H, seed=30) * 1.5).bfloat16()
    keymask = torch.ones(B, S, device="cuda")
    keymask[1, 5:min(17, S - 1)] = 0
    keymask[B - 1, S - min(9, S - 2):] = 0
    drop = ops.Drop(drop_p, seed=3, stream=2) if drop_p else ops.NO_DROP
    ctx = torch.zeros(M, H, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, heads, S, device="cuda")
    ops.attention_fwd(qkv, keymask, ctx, lse, B, S, H, heads, drop=drop)
    dctx = _rand(M, H, seed=31).bfloat16()
    dqkv = torch.zeros(M, 3 * H, dtype=torch.bfloat16, device="cuda")
    ops.attention_bwd(qkv, keymask, ctx, lse, dctx, dqkv, B, S, H, heads, drop=drop)
    qh = _to_head_major(qkv, R)
    ctx2 = torch.zeros_like(ctx); lse2 = torch.zeros_like(lse)
    ops.attention_fwd(qh, keymask, ctx2, lse2, B, S, H, heads, drop=drop, qkv_hm=R)
    dqh = torch.full((3 * heads, R, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.attention_bwd(qh, keymask, ctx2, lse2, dctx, dqh, B, S, H, heads, drop=drop, qkv_hm=R)
    torch.cuda.synchronize()
    assert torch.equal(ctx, ctx2) and torch.equal(lse, lse2)
    assert torch.equal(_from_head_major(dqh, M), dqkv)
    assert float((dqh[:, M:] - 7.0).abs().max()) == 0.0              # rows beyond the tokens are never written
    # the layouts the kernels do not serve refuse
    if S <= 64:
        return
    with pytest.raises(RuntimeError):
        ops.attention_fwd(qh, keymask, ctx2, lse2, B, S, H, heads, qkv_hm=M - 1)


def test_colsum_head_major():
    rows, R, planes, batch = 11840, 12032, 12, 3
    x = torch.randn(batch, 36, R, 64, device="cuda").bfloat16()
    stride_o = 4096
    out = torch.full((batch * stride_o,), 0.5, device="cuda")
    ops.colsum_hm(x[0], rows, R, planes, out, batch, x.stride(0), stride_o)
    torch.cuda.synchronize()
    ref = x[:, :planes, :rows].float().sum(2).reshape(batch, planes * 64)
    got = out.view(batch, stride_o)
    assert (got[:, :planes * 64] - 0.5 - ref).abs().max().item() <= 2e-5 * rows ** 0.5 * ref.abs().max().item() + 1e-3
    assert float((got[:, planes * 64:] - 0.5).abs().max()) == 0.0


@pytest.mark.parametrize("B,S,heads,drop_p,thirds,hm", [(50, 185, 12, 0.0, 1, True), (300, 185, 12, 0.0, 1, True), (3, 185, 12, 0.0, 1, False),
                                                         (9, 40, 12, 0.1, 3, False), (300, 40, 12, 0.1, 3, True), (5, 129, 5, 0.0, 1, False),
                                                         (2, 17, 1, 0.0, 3, False)])
def test_attention_backward_leaves_the_qkv_bias_gradient_as_partial_sums(B, S, heads, drop_p, thirds, hm):
    """``vault_attn_args.bias_partials``: every workgroup of the backward launch writes the column sums of dq (| dk | dv) over
    its (batch, head) items; ``vault_colsum_partials`` adds the rows - equal to the column sums of the dqkv the same launch
    stores (f32 sums of values the kernel rounds to 16 bits when storing: tolerance of that rounding), dqkv itself bit-identical
    to a launch without the partials, in both layouts, with more items than workgroups, with dropout."""
    H = heads * 64
    M = B * S
    R = ((M + 255) // 256) * 256
    qkv = (_rand(M, 3 * H, seed=40) * 1.5).bfloat16()
    keymask = torch.ones(B, S, device="cuda")
    keymask[B - 1, S - min(9, S - 2):] = 0
    drop = ops.Drop(drop_p, seed=3, stream=2) if drop_p else ops.NO_DROP
    q_in = _to_head_major(qkv, R) if hm else qkv
    kw = dict(qkv_hm=R) if hm else {}
    ctx = torch.zeros(M, H, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, heads, S, device="cuda")
    ops.attention_fwd(q_in, keymask, ctx, lse, B, S, H, heads, drop=drop, **kw)
    dctx = _rand(M, H, seed=41).bfloat16()
    shape = (3 * heads, R, 64) if hm else (M, 3 * H)
    d0 = torch.zeros(shape, dtype=torch.bfloat16, device="cuda"); d1 = torch.zeros(shape, dtype=torch.bfloat16, device="cuda")
    ops.attention_bwd(q_in, keymask, ctx, lse, dctx, d0, B, S, H, heads, drop=drop, **kw)
    n = ops.attention_bwd_partials(B, S, H, heads, thirds)
    assert n == min(B * heads, 768 if S <= 64 else 256)
    part = torch.full((n, thirds * H), 3.0, device="cuda")             # (every row is written whole: no zeroing needed)
    ops.attention_bwd(q_in, keymask, ctx, lse, dctx, d1, B, S, H, heads, drop=drop, bias_partials=part, bias_thirds=thirds, **kw)
    out = torch.full((thirds * H + 64,), 0.25, device="cuda")
    ops.colsum_partials(part, n, thirds * H, out)
    torch.cuda.synchronize()
    assert torch.equal(d0, d1)
    dq = (_from_head_major(d1, M) if hm else d1).float()
    ref = dq[:, :thirds * H].sum(0)
    got = out[:thirds * H] - 0.25
    tol = 2.0 ** -8 * float(dq.abs().max()) * M ** 0.5 + 1e-4           # un-rounded sums against sums of bf16-rounded values
    assert float((got - ref).abs().max()) <= tol, (float((got - ref).abs().max()), tol)
    assert float((got - ref).norm() / ref.norm()) < 5e-3
    assert float((out[thirds * H:] - 0.25).abs().max()) == 0.0
    if S > 64:       # all three thirds exist in the S <= 64 kernel only (LDS), none beyond the single-pass kernels
        assert ops.attention_bwd_partials(B, S, H, heads, 3) == 0 and ops.attention_bwd_partials(B, 200, H, heads, 1) == 0
        with pytest.raises(RuntimeError):
            ops.attention_bwd(q_in, keymask, ctx, lse, dctx, d1, B, S, H, heads, bias_partials=part, bias_thirds=3, **kw)
