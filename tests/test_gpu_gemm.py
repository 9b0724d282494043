"""GPU parity of the bf16 MFMA GEMM (through the C ABI) against a plain fp32 torch matmul of the
same bf16-rounded operands.  Tolerance: fp32 accumulation order only -> 2e-3 * scale absolute for
f32 outputs, plus one bf16 rounding (2^-8 relative) for bf16 outputs."""
import ctypes as C
import math

import pytest
import torch

from vault_amd import lib as L

pytestmark = pytest.mark.gpu

EPI_BF16, EPI_GELU, EPI_DGELU, EPI_RES, EPI_PATCH, EPI_ATOMIC = range(6)


def _gemm(A, B, out, M, N, K, lda, ldb, ldo, a_mode, b_mode, epi, cfg=-1, m_valid=0, splits=1, bias=None,
          res=None, aux=None, out2=None, addtab=None, rpg=0, gstride=0, goff=0, accumulate=0, colsum=None, persist=0, batch=0,
          batch_a=0, batch_b=0, batch_o=0, aux_u8=0, plan_only=False, out_hm=0, a_hm=0, splitk_ws=None):
    lib = L.load()
    a = L.GemmArgs()
    if splitk_ws is not None:
        a.splitk_ws, a.splitk_bytes = splitk_ws.data_ptr(), splitk_ws.numel() * splitk_ws.element_size()
    a.aux_u8 = aux_u8
    a.out_hm, a.a_hm = out_hm, a_hm
    a.A, a.B, a.out = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.out2 = out2.data_ptr() if out2 is not None else None
    a.bias = bias.data_ptr() if bias is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.aux = aux.data_ptr() if aux is not None else None
    a.addtab = addtab.data_ptr() if addtab is not None else None
    a.colsum = colsum.data_ptr() if colsum is not None else None
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, lda, ldb, ldo, m_valid
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.accumulate = a_mode, b_mode, epi, cfg, splits, accumulate
    a.rpg, a.gstride, a.goff = rpg, gstride, goff
    a.persist = persist
    a.batch, a.batch_a, a.batch_b, a.batch_o = batch, batch_a, batch_b, batch_o
    if plan_only:
        return int(lib.vault_gemm_plan(C.byref(a)))
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.vault_gemm(C.byref(a), C.c_void_p(st)), "vault_gemm")


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 7])
@pytest.mark.parametrize("shape", [(512, 768, 768), (256, 2304, 768), (768, 768, 3072), (256, 256, 64), (512, 256, 192), (256, 512, 320)])
def test_forward_nt_bias(cfg, shape):
    M, N, K = shape
    A = _rand(M, K, seed=1).bfloat16()
    W = _rand(N, K, scale=0.05, seed=2).bfloat16()
    bias = _rand(N, seed=3)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    m_valid = M - 37
    _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, m_valid=m_valid)
    ref = A.float() @ W.float().t() + bias
    torch.cuda.synchronize()
    err = (out[:m_valid].float() - ref[:m_valid]).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= scale * 2 ** -7, (err, scale)
    assert out[m_valid:].abs().max().item() == 0.0  # masked rows untouched


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 7])
def test_forward_gelu_and_residual(cfg):
    M, N, K = 512, 1024, 256
    A = _rand(M, K, seed=4).bfloat16()
    W = _rand(N, K, scale=0.1, seed=5).bfloat16()
    bias = _rand(N, seed=6)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    pre = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_GELU, cfg=cfg, bias=bias, out2=pre)
    z = A.float() @ W.float().t() + bias
    ref = torch.nn.functional.gelu(z)
    torch.cuda.synchronize()
    gprime = 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    assert (pre.float() - gprime).abs().max().item() <= 2 ** -7        # out2 = gelu'(pre-activation)
    assert (out.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7
    # f32 residual epilogue
    res = _rand(M, N, seed=7)
    o32 = torch.zeros(M, N, device="cuda")
    _gemm(A, W, o32, M, N, K, K, K, N, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res)
    torch.cuda.synchronize()
    assert (o32 - (z + res)).abs().max().item() <= 2e-4 * z.abs().max().item()


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 7])
def test_dgrad_nn_and_dgelu(cfg):
    # dX[M,Kin] = dY[M,Nout] . W[Nout,Kin]  (A mode 0, B mode 1)
    M, Nout, Kin = 512, 768, 1024
    dY = _rand(M, Nout, seed=8).bfloat16()
    W = _rand(Nout, Kin, scale=0.05, seed=9).bfloat16()
    dX = torch.zeros(M, Kin, dtype=torch.bfloat16, device="cuda")
    _gemm(dY, W, dX, M, Kin, Nout, Nout, Kin, Kin, 0, 1, EPI_BF16, cfg=cfg)
    ref = dY.float() @ W.float()
    torch.cuda.synchronize()
    assert (dX.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7
    u = _rand(M, Kin, seed=10).bfloat16()     # stands for the stored gelu'
    _gemm(dY, W, dX, M, Kin, Nout, Nout, Kin, Kin, 0, 1, EPI_DGELU, cfg=cfg, aux=u)
    ref2 = ref * u.float()
    torch.cuda.synchronize()
    assert (dX.float() - ref2).abs().max().item() <= ref2.abs().max().item() * 2 ** -7


@pytest.mark.parametrize("cfg", [0, 2, 3])
@pytest.mark.parametrize("splits", [1, 3, 5])
def test_wgrad_tn_splitk(cfg, splits):
    # dW[Nout,Kin] = dY[Mtok,Nout]^T . X[Mtok,Kin]  (A mode 1, B mode 1), contraction over tokens
    Mtok, Nout, Kin = 1024, 768, 512
    dY = _rand(Mtok, Nout, seed=11).bfloat16()
    X = _rand(Mtok, Kin, seed=12).bfloat16()
    dY[1000:] = 0  # pad rows of a token buffer are zero
    dW = torch.zeros(Nout, Kin, device="cuda")
    _gemm(dY, X, dW, Nout, Kin, Mtok, Nout, Kin, Kin, 1, 1, EPI_ATOMIC, cfg=cfg, splits=splits)
    ref = dY.float().t() @ X.float()
    torch.cuda.synchronize()
    assert (dW - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-3
    # accumulate on top
    _gemm(dY, X, dW, Nout, Kin, Mtok, Nout, Kin, Kin, 1, 1, EPI_ATOMIC, cfg=cfg, splits=splits, accumulate=1)
    torch.cuda.synchronize()
    assert (dW - 2 * ref).abs().max().item() <= 4e-4 * ref.abs().max().item() + 2e-3


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("splits", [1, 3])
def test_wgrad_batched_layers(cfg, splits):
    """`batch` weight gradients of one shape in one launch (ABI 3): problem b reads slice b of the stacked dY / X
    tensors and accumulates into out + b * batch_o (the layers of a stack in the flat gradient buffer, with other
    tensors in between); the 192-wide ring form and the non-atomic epilogues refuse batches."""
    L_, Mtok, Nout, Kin = 3, 768, 512, 256
    dY = _rand(L_, Mtok, Nout, seed=21).bfloat16()
    X = _rand(L_, Mtok, Kin, seed=22).bfloat16()
    stride_o = Nout * Kin + 4096                      # gap between the layers' dW (other parameters of a layer)
    flat = torch.ones(L_ * stride_o, device="cuda")
    _gemm(dY, X, flat, Nout, Kin, Mtok, Nout, Kin, Kin, 1, 1, EPI_ATOMIC, cfg=cfg, splits=splits, accumulate=1,
          batch=L_, batch_a=Mtok * Nout, batch_b=Mtok * Kin, batch_o=stride_o)
    torch.cuda.synchronize()
    for b in range(L_):
        ref = 1.0 + dY[b].float().t() @ X[b].float()
        got = flat[b * stride_o:b * stride_o + Nout * Kin].view(Nout, Kin)
        assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-3, b
        assert torch.equal(flat[b * stride_o + Nout * Kin:(b + 1) * stride_o], torch.ones(4096, device="cuda"))
    with pytest.raises(RuntimeError):
        _gemm(dY, X, flat, Nout, Kin, Mtok, Nout, Kin, Kin, 1, 1, EPI_ATOMIC, cfg=4, splits=1, accumulate=1,
              batch=L_, batch_a=Mtok * Nout, batch_b=Mtok * Kin, batch_o=stride_o)


@pytest.mark.parametrize("persist", [0, 1])
def test_wgrad_batched_ring_many_items(persist):
    """Ring kernel with more work items than blocks (12 problems x 9 tiles x 3 splits = 324 > 256): every block walks
    several (problem, split, tile) items, statically or through the ticket scheduler."""
    L_, Mtok, Nout, Kin = 12, 1024, 768, 768
    dY = _rand(L_, Mtok, Nout, seed=23).bfloat16()
    X = _rand(L_, Mtok, Kin, seed=24).bfloat16()
    stride_o = Nout * Kin + 768
    flat = torch.zeros(L_ * stride_o, device="cuda")
    for _ in range(2):   # (two launches: the dynamic scheduler's double-buffered ticket counters)
        _gemm(dY, X, flat, Nout, Kin, Mtok, Nout, Kin, Kin, 1, 1, EPI_ATOMIC, cfg=3, splits=3, accumulate=1,
              batch=L_, batch_a=Mtok * Nout, batch_b=Mtok * Kin, batch_o=stride_o, persist=persist)
    torch.cuda.synchronize()
    for b in range(L_):
        ref = 2.0 * (dY[b].float().t() @ X[b].float())
        got = flat[b * stride_o:b * stride_o + Nout * Kin].view(Nout, Kin)
        assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 2e-3, b
        assert float(flat[b * stride_o + Nout * Kin:(b + 1) * stride_o].abs().max()) == 0.0


def test_patch_epilogue_rowmap():
    # rows of group b (rpg patches) land at b*gstride + goff + p, plus a per-patch additive table
    B_, rpg, N, K = 4, 144, 256, 192
    M = 640  # 576 valid rows padded to a tile multiple
    A = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    A[: B_ * rpg] = _rand(B_ * rpg, K, seed=13).bfloat16()
    W = _rand(N, K, scale=0.1, seed=14).bfloat16()
    tab = _rand(rpg, N, seed=15)
    S = 185
    out = torch.zeros(B_ * S, N, device="cuda")
    _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_PATCH, cfg=0, m_valid=B_ * rpg, addtab=tab, rpg=rpg,
          gstride=S, goff=41)
    ref = (A[: B_ * rpg].float() @ W.float().t()).view(B_, rpg, N) + tab
    torch.cuda.synchronize()
    got = out.view(B_, S, N)[:, 41:41 + rpg]
    assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-4
    assert out.view(B_, S, N)[:, :41].abs().max().item() == 0.0


def test_einval_on_bad_shapes():
    A = torch.zeros(128, 64, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(128, 128, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        _gemm(A, A, out, 100, 128, 64, 64, 64, 128, 0, 0, EPI_BF16)


@pytest.mark.parametrize("mode", ["nt", "nn", "tn"])
def test_phase_kernel_repeatability_and_large_k(mode):
    """The 8-phase kernel (cfg 3) relies on counted waits: run a long-K problem several times and demand
    bit-identical results (a race shows up as run-to-run differences) that also match the reference."""
    M, N, K = 512, 768, 3072
    A = _rand(M, K, seed=21).bfloat16()
    Bm = _rand(N, K, scale=0.05, seed=22).bfloat16()
    ref = A.float() @ Bm.float().t()
    outs = []
    for _ in range(5):
        if mode == "nt":
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            _gemm(A, Bm, out, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=3)
        elif mode == "nn":
            Bt = Bm.t().contiguous()          # [K][N]
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            _gemm(A, Bt, out, M, N, K, K, N, N, 0, 1, EPI_BF16, cfg=3)
        else:
            At, Bt = A.t().contiguous(), Bm.t().contiguous()   # [K][M], [K][N]
            out = torch.zeros(M, N, device="cuda")
            _gemm(At, Bt, out, M, N, K, M, N, N, 1, 1, EPI_ATOMIC, cfg=3, splits=1)
        torch.cuda.synchronize()
        outs.append(out.float().clone())
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert (outs[0] - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 7])
def test_fused_bias_gradient_colsum(cfg):
    """bf16 epilogues can accumulate the column sums of what they store (rows < m_valid): the bias
    gradient of the Linear whose dY this GEMM produces."""
    M, Nout, Kin = 512, 768, 1024
    dY = _rand(M, Nout, seed=31).bfloat16()
    W = _rand(Nout, Kin, scale=0.05, seed=32).bfloat16()
    u = _rand(M, Kin, seed=33).bfloat16()
    dX = torch.zeros(M, Kin, dtype=torch.bfloat16, device="cuda")
    cs = torch.full((Kin,), 2.0, device="cuda")
    m_valid = 500
    _gemm(dY, W, dX, M, Kin, Nout, Nout, Kin, Kin, 0, 1, EPI_DGELU, cfg=cfg, aux=u, colsum=cs, m_valid=m_valid)
    ref = (dY.float() @ W.float()) * u.float()
    torch.cuda.synchronize()
    want = 2.0 + ref[:m_valid].sum(0)
    assert (cs - want).abs().max().item() <= 2e-3 * want.abs().max().item() + 1e-2


@pytest.mark.parametrize("K", [64, 192, 768, 3072])
@pytest.mark.parametrize("mode", ["nt", "nn"])
def test_ring_kernel_192_wide_tiles(mode, K):
    """cfg 4 = the 8-phase ring kernel with 256x192 block tiles (used for N = 768): every epilogue it serves."""
    M, N = 512, 768
    m_valid = 477
    A = _rand(M, K, seed=41).bfloat16()
    Wnk = _rand(N, K, scale=0.05, seed=42).bfloat16()
    ref = A.float() @ Wnk.float().t()
    if mode == "nt":
        Bop, ldb, bm = Wnk, K, 0
    else:
        Bop, ldb, bm = Wnk.t().contiguous(), N, 1
    bias = _rand(N, seed=43)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    _gemm(A, Bop, out, M, N, K, K, ldb, N, 0, bm, EPI_BF16, cfg=4, bias=bias, m_valid=m_valid)
    torch.cuda.synchronize()
    r = ref + bias
    assert (out[:m_valid].float() - r[:m_valid]).abs().max().item() <= r.abs().max().item() * 2 ** -7
    assert out[m_valid:].abs().max().item() == 0.0
    res = _rand(M, N, seed=44)
    o32 = torch.zeros(M, N, device="cuda")
    _gemm(A, Bop, o32, M, N, K, K, ldb, N, 0, bm, EPI_RES, cfg=4, bias=bias, res=res)
    torch.cuda.synchronize()
    assert (o32 - (r + res)).abs().max().item() <= 2e-4 * r.abs().max().item() + 1e-4
    g = _rand(M, N, seed=45).bfloat16()
    cs = torch.zeros(N, device="cuda")
    _gemm(A, Bop, out, M, N, K, K, ldb, N, 0, bm, EPI_DGELU, cfg=4, aux=g, colsum=cs, m_valid=m_valid)
    torch.cuda.synchronize()
    r2 = ref * g.float()
    assert (out[:m_valid].float() - r2[:m_valid]).abs().max().item() <= r2.abs().max().item() * 2 ** -7
    assert (cs - r2[:m_valid].sum(0)).abs().max().item() <= 2e-3 * r2[:m_valid].sum(0).abs().max().item() + 1e-2
    pre = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    _gemm(A, Bop, out, M, N, K, K, ldb, N, 0, bm, EPI_GELU, cfg=4, bias=bias, out2=pre)
    torch.cuda.synchronize()
    assert (out.float() - torch.nn.functional.gelu(r)).abs().max().item() <= r.abs().max().item() * 2 ** -7


# ---- persistent ring kernel: > 256 work items, so blocks walk several tiles and the epilogue of one tile
#      overlaps the staging / first phases of the next (counted waits that allow for the stores in flight)
@pytest.mark.parametrize("cfg", [3, 4])
@pytest.mark.parametrize("K", [64, 192, 256, 832])
@pytest.mark.parametrize("epi", ["bf16", "gelu", "res"])
def test_ring_kernel_persistent_overlap_forward(cfg, K, epi):
    M, N = 256 * 43, (1536 if cfg == 4 else 2048)      # 43 x 8 = 344 tiles > 256 blocks
    A = _rand(M, K, seed=31).bfloat16()
    W = _rand(N, K, scale=0.05, seed=32).bfloat16()
    bias = _rand(N, seed=33)
    m_valid = M - 300                                   # last TWO tile rows are partial / one fully masked row block
    z = A.float() @ W.float().t() + bias
    for rep in range(3):                                # repeat: a racy wait shows as run-to-run differences
        if epi == "bf16":
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, m_valid=m_valid)
            ref, tol = z, z.abs().max().item() * 2 ** -7
        elif epi == "gelu":
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            pre = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_GELU, cfg=cfg, bias=bias, out2=pre, m_valid=m_valid)
            ref = torch.nn.functional.gelu(z); tol = ref.abs().max().item() * 2 ** -7
            gprime = 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
            assert (pre[:m_valid].float() - gprime[:m_valid]).abs().max().item() <= 2 ** -7
        else:
            res = _rand(M, N, seed=34)
            out = torch.zeros(M, N, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res, m_valid=m_valid)
            ref, tol = z + res, 2e-4 * z.abs().max().item()
        torch.cuda.synchronize()
        assert (out[:m_valid].float() - ref[:m_valid]).abs().max().item() <= tol
        assert out[m_valid:].abs().max().item() == 0.0


@pytest.mark.parametrize("cfg", [3, 4])
def test_ring_kernel_persistent_overlap_dgrad_colsum(cfg):
    M, K, N = 256 * 43, 512, (1536 if cfg == 4 else 2048)
    dY = _rand(M, K, seed=41).bfloat16()
    W = _rand(K, N, scale=0.05, seed=42).bfloat16()     # [K][N]: b_mode 1
    aux = _rand(M, N, seed=43).bfloat16()
    ref = (dY.float() @ W.float()) * aux.float()
    for rep in range(2):
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        cs = torch.zeros(N, device="cuda")
        _gemm(dY, W, out, M, N, K, K, N, N, 0, 1, EPI_DGELU, cfg=cfg, aux=aux, colsum=cs)
        torch.cuda.synchronize()
        assert (out.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7
        csr = out.float().sum(0)
        assert (cs - csr).abs().max().item() <= 2e-3 * csr.abs().max().item() + 1e-2


@pytest.mark.parametrize("splits", [1, 3, 8])
def test_ring_kernel_persistent_wgrad(splits):
    # out[M][N] (+)= A[K][M]^T B[K][N]: 12 x 12 = 144 tiles x splits work items
    M, N, K = 3072, 3072, 1280
    A = _rand(K, M, seed=51).bfloat16()
    B = _rand(K, N, scale=0.05, seed=52).bfloat16()
    ref = A.float().t() @ B.float()
    out = torch.zeros(M, N, device="cuda")
    _gemm(A, B, out, M, N, K, M, N, N, 1, 1, EPI_ATOMIC, cfg=3, splits=splits)
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()



@pytest.mark.parametrize("cfg", [3, 4])
@pytest.mark.parametrize("epi", ["bf16", "res"])
def test_ring_kernel_dynamic_scheduler(cfg, epi):
    """persist = 1: work items handed out by per-XCD ticket counters (with stealing) instead of the static
    block -> items walk; many back-to-back launches also exercise the self-reset of the counters."""
    M, N, K = 256 * 43, (1536 if cfg == 4 else 2048), 320
    A = _rand(M, K, seed=61).bfloat16()
    W = _rand(N, K, scale=0.05, seed=62).bfloat16()
    bias = _rand(N, seed=63)
    z = A.float() @ W.float().t() + bias
    res = _rand(M, N, seed=64)
    for rep in range(6):
        if epi == "bf16":
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, persist=1)
            ref, tol = z, z.abs().max().item() * 2 ** -7
        else:
            out = torch.zeros(M, N, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res, persist=1)
            ref, tol = z + res, 2e-4 * z.abs().max().item()
        # a small launch in between: fewer work items than blocks (every block exits after its static item)
        o2 = torch.zeros(512, 512, dtype=torch.bfloat16, device="cuda")
        _gemm(A[:512], W[:512], o2, 512, 512, K, K, K, 512, 0, 0, EPI_BF16, cfg=3, persist=1)
        torch.cuda.synchronize()
        assert (out.float() - ref).abs().max().item() <= tol
        r2 = A[:512].float() @ W[:512].float().t()
        assert (o2.float() - r2).abs().max().item() <= r2.abs().max().item() * 2 ** -7


def test_ring_kernel_dynamic_scheduler_wgrad_splits():
    M, N, K = 3072, 3072, 1280
    A = _rand(K, M, seed=71).bfloat16()
    B = _rand(K, N, scale=0.05, seed=72).bfloat16()
    ref = A.float().t() @ B.float()
    for splits in (1, 3, 8):
        out = torch.zeros(M, N, device="cuda")
        _gemm(A, B, out, M, N, K, M, N, N, 1, 1, EPI_ATOMIC, cfg=3, splits=splits, persist=1)
        torch.cuda.synchronize()
        assert (out - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


# ---- 8-wave kernel with register-direct epilogue (cfg 5: 256-wide tiles, cfg 6: 192-wide; gemm8w.hip): forward-form
#      operands (A [M][K], B [N][K]), swapped-operand MFMA with permuted weight rows, loads running ahead across tiles
def _gprime(z):
    return 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("cfg", [5, 6])
@pytest.mark.parametrize("shape", [(256, 768, 128), (512, 768, 768), (256, 2304, 192), (768, 768, 3072), (1280, 1536, 320),
                                   (256 * 43, 3072, 832), (256 * 150, 768, 768), (512, 1536, 256), (256 * 70, 768, 256)])
@pytest.mark.parametrize("epi", ["bf16", "gelu", "gelu_inf", "res", "dgelu"])
def test_8wave_kernel_every_epilogue(cfg, shape, epi):
    """One tile row, fewer tiles than the 256 resident blocks, several tiles per block, odd and even K tile counts (K < 256: the
    two-slot pipeline; K = 256: the shortest contraction of the three-A-slot one, whose look-ahead then always reaches into the
    next work item), partial and fully masked row blocks; three runs each (a racy wait shows as run-to-run differences)."""
    M, N, K = shape
    A = _rand(M, K, seed=81).bfloat16()
    W = _rand(N, K, scale=0.05, seed=82).bfloat16()
    bias = _rand(N, seed=83)
    m_valid = M - 150 if M > 256 else M - 37
    z = A.float() @ W.float().t() + bias
    first = None
    for rep in range(3):
        if epi == "bf16":
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            cs = torch.full((N,), 1.0, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, m_valid=m_valid, colsum=cs)
            ref, tol = z, z.abs().max().item() * 2 ** -7
            torch.cuda.synchronize()
            want = 1.0 + ref[:m_valid].sum(0)            # (the kernel sums its f32 values, before the 16-bit rounding of the output)
            assert (cs - want).abs().max().item() <= 2e-3 * want.abs().max().item() + 1e-2
        elif epi in ("gelu", "gelu_inf"):
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            pre = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda") if epi == "gelu" else None
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_GELU, cfg=cfg, bias=bias, out2=pre, m_valid=m_valid)
            ref = torch.nn.functional.gelu(z); tol = ref.abs().max().item() * 2 ** -7
            if pre is not None:
                torch.cuda.synchronize()
                assert (pre[:m_valid].float() - _gprime(z)[:m_valid]).abs().max().item() <= 2 ** -7
                assert pre[m_valid:].abs().max().item() == 0.0
        elif epi == "res":
            res = _rand(M, N, seed=84)
            out = torch.zeros(M, N, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res, m_valid=m_valid)
            ref, tol = z + res, 2e-4 * z.abs().max().item() + 1e-5
        else:
            aux = _rand(M, N, seed=85).bfloat16()
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            cs = torch.zeros(N, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_DGELU, cfg=cfg, aux=aux, colsum=cs, m_valid=m_valid)
            ref = (z - bias) * aux.float(); tol = ref.abs().max().item() * 2 ** -7
            torch.cuda.synchronize()
            want = ref[:m_valid].sum(0)
            assert (cs - want).abs().max().item() <= 2e-3 * want.abs().max().item() + 1e-2
        torch.cuda.synchronize()
        assert (out[:m_valid].float() - ref[:m_valid]).abs().max().item() <= tol
        assert out[m_valid:].abs().max().item() == 0.0          # masked rows untouched
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(out, first)


def test_8wave_kernel_refuses_what_it_does_not_take():
    A = torch.zeros(256, 256, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(256, 256, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):      # k-strided weights (dgrad form): the engine feeds it the transposed shadow
        _gemm(A, A, out, 256, 256, 128, 256, 256, 256, 0, 1, EPI_BF16, cfg=5)
    with pytest.raises(RuntimeError):      # a single K tile
        _gemm(A, A, out, 256, 256, 64, 256, 256, 256, 0, 0, EPI_BF16, cfg=5)
    with pytest.raises(RuntimeError):      # N not a multiple of 192
        _gemm(A, A, out, 256, 256, 128, 256, 256, 256, 0, 0, EPI_BF16, cfg=6)


@pytest.mark.parametrize("cfg", [5, 6])
@pytest.mark.parametrize("shape", [(512, 768, 256, 500), (1024, 1536, 768, 1024), (2560, 768, 192, 2309)])
def test_8wave_kernel_8bit_gelu_prime_round_trip(cfg, shape):
    """aux_u8: the FFN-in forward (epi 1) leaves gelu' as an 8-bit tile-native image, the gelu'-product dgrad (epi 2) of the same
    M, N and kernel reads it back: dX = (dY W2) * gelu'(z) against the exact derivative, within bf16 rounding + half a
    quantisation step (0.0025) of the product; the activation output is bit-identical to the bf16-gelu' form; column sums
    (bias gradient) follow the stored values."""
    M, N, K, mv = shape
    A = _rand(M, K, seed=41).bfloat16()
    W = _rand(N, K, scale=0.08, seed=42).bfloat16()
    bias = _rand(N, seed=43)
    act8 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); act16 = torch.zeros_like(act8)
    u8 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")      # (2 M N bytes: covers M N (256-wide) and 4/3 M N (192-wide))
    u16 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    assert _gemm(A, W, act8, M, N, K, K, K, N, 0, 0, EPI_GELU, cfg=cfg, bias=bias, out2=u8, aux_u8=1, plan_only=True) == cfg
    _gemm(A, W, act8, M, N, K, K, K, N, 0, 0, EPI_GELU, cfg=cfg, m_valid=mv, bias=bias, out2=u8, aux_u8=1)
    _gemm(A, W, act16, M, N, K, K, K, N, 0, 0, EPI_GELU, cfg=cfg, m_valid=mv, bias=bias, out2=u16)
    torch.cuda.synchronize()
    assert torch.equal(act8, act16)
    z = A.float() @ W.float().t() + bias
    gprime = 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    # backward GEMM of the same output shape: dX[M, N] = dY[M, K2] . W2t[N, K2]^T, times gelu'
    K2 = 256
    dY = _rand(M, K2, seed=44).bfloat16()
    W2t = _rand(N, K2, scale=0.05, seed=45).bfloat16()
    dX8 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda"); dX16 = torch.zeros_like(dX8)
    cs8 = torch.zeros(N, device="cuda"); cs16 = torch.zeros(N, device="cuda")
    _gemm(dY, W2t, dX8, M, N, K2, K2, K2, N, 0, 0, EPI_DGELU, cfg=cfg, m_valid=mv, aux=u8, aux_u8=1, colsum=cs8)
    _gemm(dY, W2t, dX16, M, N, K2, K2, K2, N, 0, 0, EPI_DGELU, cfg=cfg, m_valid=mv, aux=u16, colsum=cs16)
    torch.cuda.synchronize()
    pre = dY.float() @ W2t.float().t()
    ref = (pre * gprime)[:mv]
    tol = ref.abs().max().item() * 2 ** -7 + 0.0026 * pre.abs().max().item()
    assert (dX8[:mv].float() - ref).abs().max().item() <= tol
    assert (dX16[:mv].float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7 + 2 ** -8 * pre.abs().max().item()
    assert dX8[mv:].abs().max().item() == 0.0 if mv < M else True
    # quantisation error is unbiased to first order: the mean product error stays far below the step
    assert abs((dX8[:mv].float() - ref).mean().item()) < 2e-4 * pre.abs().max().item()
    want = dX8[:mv].float().sum(0)
    assert (cs8 - want).abs().max().item() <= 2e-3 * want.abs().max().item() + 2e-2
    # the 8-bit form exists in the 8-wave kernel's tile order only
    assert _gemm(A, W, act8, M, N, K, K, K, N, 0, 0, EPI_GELU, cfg=3, bias=bias, out2=u8, aux_u8=1, plan_only=True) < 0
    with pytest.raises(RuntimeError):
        _gemm(A, W, act8, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, aux_u8=1)


def test_small_launches_take_64_row_tiles():
    """Launches of at most 256 blocks of 64 x 128 resolve to cfg 7 (the LM stack at per-GPU batch <= 64, both stacks at batch 8),
    one more block and they stay on 128 x 128; same results as the 128 x 128 form (same K order: bit-identical)."""
    for (M, N, K, want) in [(2560, 768, 3072, 7), (2560, 768, 768, 7), (2816, 768, 3072, 0), (512, 3072, 768, 7), (768, 3072, 768, 0),
                            (512, 2304, 768, 7)]:
        A = _rand(M, K, seed=1).bfloat16()
        W = _rand(N, K, scale=0.05, seed=2).bfloat16()
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        got = _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_BF16, plan_only=True)
        assert (got == 7) == (want == 7), (M, N, K, got)
        ref = torch.zeros_like(out)
        _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_BF16)
        _gemm(A, W, ref, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=0)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    # not for weight gradients, split-K or a_mode 1
    dY = _rand(512, 768, seed=3).bfloat16(); X = _rand(512, 768, seed=4).bfloat16()
    dW = torch.zeros(768, 768, device="cuda")
    assert _gemm(dY, X, dW, 768, 768, 512, 768, 768, 768, 1, 1, EPI_ATOMIC, plan_only=True) != 7
    assert _gemm(dY, X, dW, 768, 768, 512, 768, 768, 768, 1, 1, EPI_ATOMIC, cfg=7, plan_only=True) < 0


@pytest.mark.parametrize("half", ["bf16", "fp16"])
@pytest.mark.parametrize("splits,accumulate", [(1, 0), (1, 1), (3, 1)])
def test_grouped_weight_gradients_against_matmul(half, splits, accumulate):
    """vault_wgrad_grouped: three kinds of weight gradient (different n_out / n_in, two layers each, layers at a stride) cut
    into segments that start and end in the middle of a layer, in two launches that together cover every tile exactly once:
    dW = dY^T X against a float matmul of the same 16-bit operands; tiles outside a launch's segments stay untouched."""
    from vault_amd import ops
    dt = ops.HALF_DTYPE[half]
    tokens, G = 448, 2                       # 7 K tiles: the 3-way split is ragged (3 + 3 + 1)
    kinds = [(512, 768), (768, 256), (256, 256)]        # (n_out, n_in): 6, 3, 1 tiles per layer
    dys = [(_rand(G, tokens, no, seed=10 + k) * 0.5).to(dt) for k, (no, ni) in enumerate(kinds)]
    xs = [(_rand(G, tokens, ni, seed=20 + k) * 0.5).to(dt) for k, (no, ni) in enumerate(kinds)]
    # dW of a kind's layers at a stride larger than the matrix (the flat gradient buffer's layout)
    pads = [no * ni + 1024 for no, ni in kinds]
    base = 0.25 if accumulate else 0.0
    dws = [torch.full((G, pads[k]), base, device="cuda") for k in range(3)]
    tiles = [(no // 256) * (ni // 256) * G for no, ni in kinds]          # 12, 6, 2 items
    # launch 1: kind 0 items 0..8, kind 1 items 0..1; launch 2: kind 0 items 9..11, kind 1 items 2..5, kind 2 all
    plan = [[(0, 0, 9), (1, 0, 2)], [(0, 9, 3), (1, 2, 4), (2, 0, 2)]]
    with ops.operand_format(half):
        for segs in plan:
            args = [dict(dy=dys[k][0], x=xs[k][0], dw=dws[k][0], n_out=kinds[k][0], n_in=kinds[k][1], batch=G, first=f, count=c,
                         batch_dy=dys[k].stride(0), batch_x=xs[k].stride(0), batch_dw=pads[k]) for k, f, c in segs]
            ops.wgrad_grouped(args, tokens, splits=splits, accumulate=accumulate)
    torch.cuda.synchronize()
    assert sum(c for segs in plan for _, _, c in segs) == sum(tiles)
    for k, (no, ni) in enumerate(kinds):
        for l in range(G):
            ref = dys[k][l].float().t() @ xs[k][l].float() + base
            got = dws[k][l, :no * ni].view(no, ni)
            err = float((got - ref).abs().max())
            assert err <= 2e-3 * float(ref.abs().max()), (k, l, err)
            assert float((dws[k][l, no * ni:] - base).abs().max()) == 0.0        # the padding behind the matrix is untouched
    # a segment that reaches beyond its kind's tiles is refused
    with pytest.raises(RuntimeError):
        with ops.operand_format(half):
            ops.wgrad_grouped([dict(dy=dys[2][0], x=xs[2][0], dw=dws[2][0], n_out=256, n_in=256, batch=G, first=1, count=2,
                                    batch_dy=dys[2].stride(0), batch_x=xs[2].stride(0), batch_dw=pads[2])], tokens)


@pytest.mark.parametrize("K", [64, 192, 768, 3072])
@pytest.mark.parametrize("rows", [512, 256 * 43])
def test_ring_kernel_128_wide_tiles(K, rows):
    """cfg 8 = the ring kernel with 256 x 128 block tiles, the two forms it exists in: residual forward (0,0) and the (0,1)
    data gradient; one round (12 tiles) and a persistent walk over 43 x 6 = 258 tiles with a partial last row block."""
    M, N = rows, 768
    m_valid = M - 35
    A = _rand(M, K, seed=61).bfloat16()
    Wnk = _rand(N, K, scale=0.05, seed=62).bfloat16()
    ref = A.float() @ Wnk.float().t()
    bias = _rand(N, seed=63)
    res = _rand(M, N, seed=64)
    for rep in range(2):
        o32 = torch.zeros(M, N, device="cuda")
        _gemm(A, Wnk, o32, M, N, K, K, K, N, 0, 0, EPI_RES, cfg=8, bias=bias, res=res, m_valid=m_valid)
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        _gemm(A, Wnk.t().contiguous(), out, M, N, K, K, N, N, 0, 1, EPI_BF16, cfg=8, m_valid=m_valid)
        torch.cuda.synchronize()
        r = ref + bias + res
        assert (o32[:m_valid] - r[:m_valid]).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-4
        assert o32[m_valid:].abs().max().item() == 0.0
        assert (out[:m_valid].float() - ref[:m_valid]).abs().max().item() <= ref.abs().max().item() * 2 ** -7
        assert out[m_valid:].abs().max().item() == 0.0
    # not the other forms
    with pytest.raises(RuntimeError):
        _gemm(A, Wnk, out, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=8, bias=bias)


def _to_hm(x, R):
    M, N = x.shape
    out = torch.zeros(N // 64, R, 64, dtype=x.dtype, device=x.device)
    out[:, :M] = x.view(M, N // 64, 64).permute(1, 0, 2)
    return out


@pytest.mark.parametrize("cfg", [5, 6])
@pytest.mark.parametrize("M,m_valid", [(512, 477), (256 * 43, 256 * 43 - 300)])
def test_8wave_kernel_head_major_output(cfg, M, m_valid):
    """``out_hm``: the QKV forward's 16-bit output written as [N / 64][R][64] (what the attention kernels read with qkv_hm)
    equals the row-major output bit for bit; rows >= m_valid and >= M of every plane stay untouched."""
    N, K, R = 2304, 768, M + 256
    A = _rand(M, K, seed=71).bfloat16()
    W = _rand(N, K, scale=0.05, seed=72).bfloat16()
    bias = _rand(N, seed=73)
    rm = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    _gemm(A, W, rm, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, m_valid=m_valid)
    hm = torch.full((N // 64, R, 64), 3.0, dtype=torch.bfloat16, device="cuda")
    _gemm(A, W, hm, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, m_valid=m_valid, out_hm=R)
    torch.cuda.synchronize()
    assert torch.equal(hm[:, :m_valid].permute(1, 0, 2).reshape(m_valid, N), rm[:m_valid])
    assert float((hm[:, m_valid:] - 3.0).abs().max()) == 0.0
    with pytest.raises(RuntimeError):       # not on the other kernels / epilogues
        _gemm(A, W, hm, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=3, bias=bias, out_hm=R)


@pytest.mark.parametrize("cfg", [3, 4, 8])
@pytest.mark.parametrize("M", [512, 256 * 43])
def test_ring_kernel_head_major_a_operand(cfg, M):
    """``a_hm``: the QKV data gradient reading dqkv as [K / 64][R][64]: same result as from the row-major operand, bit for
    bit (the same K order), in the (0,1) form of every tile width."""
    K, N, R = 2304, 768, M + 512
    dY = _rand(M, K, seed=81).bfloat16()
    W = _rand(K, N, scale=0.05, seed=82).bfloat16()          # [K][N]: b_mode 1
    rm = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    _gemm(dY, W, rm, M, N, K, K, N, N, 0, 1, EPI_BF16, cfg=cfg)
    o = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    _gemm(_to_hm(dY, R), W, o, M, N, K, K, N, N, 0, 1, EPI_BF16, cfg=cfg, a_hm=R)
    torch.cuda.synchronize()
    assert torch.equal(o, rm)


@pytest.mark.parametrize("splits", [1, 3])
def test_grouped_weight_gradients_head_major_dy(splits):
    """The QKV kind of a grouped weight-gradient launch with its dY (dqkv) head-major, beside a row-major kind."""
    from vault_amd import ops
    tokens, G, R = 448, 2, 512
    kinds = [(768, 256), (512, 256)]                       # (n_out, n_in); kind 0's dY is head-major
    dys = [(_rand(G, tokens, no, seed=90 + k) * 0.5).bfloat16() for k, (no, ni) in enumerate(kinds)]
    xs = [(_rand(G, tokens, ni, seed=95 + k) * 0.5).bfloat16() for k, (no, ni) in enumerate(kinds)]
    dy0_hm = torch.stack([_to_hm(dys[0][l], R) for l in range(G)])          # [G][12][R][64]
    dws = [torch.zeros(G, no * ni, device="cuda") for no, ni in kinds]
    args = [dict(dy=dy0_hm[0], x=xs[0][0], dw=dws[0][0], n_out=768, n_in=256, batch=G, first=0, count=6, batch_dy=dy0_hm.stride(0),
                 batch_x=xs[0].stride(0), batch_dw=768 * 256, dy_hm=R),
            dict(dy=dys[1][0], x=xs[1][0], dw=dws[1][0], n_out=512, n_in=256, batch=G, first=0, count=4, batch_dy=dys[1].stride(0),
                 batch_x=xs[1].stride(0), batch_dw=512 * 256)]
    ops.wgrad_grouped(args, tokens, splits=splits, accumulate=1)
    torch.cuda.synchronize()
    for k, (no, ni) in enumerate(kinds):
        for l in range(G):
            ref = dys[k][l].float().t() @ xs[k][l].float()
            err = float((dws[k][l].view(no, ni) - ref).abs().max())
            assert err <= 2e-3 * float(ref.abs().max()), (k, l, err)


# ---- split-K with the reduction inside the launch (ring kernel, 192-wide tiles: gemm256.hip SK; vault_gemm_args.splitk_ws)
@pytest.mark.parametrize("persist", [0, 1])
@pytest.mark.parametrize("epi", ["dgrad", "res"])
@pytest.mark.parametrize("shape", [(1280, 3072, 2), (6144, 3072, 2), (6144, 2304, 3), (2560, 3072, 4), (12032, 3072, 2), (768, 1536, 3)])
def test_ring_kernel_split_k_reduces_inside_the_launch(shape, epi, persist):
    """N = 768 Linears with long contractions at small batches (FFN-in / QKV data gradients: (0,1) EPI_BF16 + column sums; FFN-out
    forward: (0,0) EPI_F32_RES): every K split stores its accumulators into the caller's workspace, the split that draws the last
    ticket adds the slabs in split order and runs the epilogue.  Checked: every output word against the f32 product of the same
    operands and against the un-split kernel (accumulation order only), rows >= m_valid untouched, three runs bit-identical
    (the result must not depend on which split arrives last), the tile counters zero again after every launch, with fewer items
    than blocks (one item per block), more (376 and 564 items: persistent blocks take a second, staged item behind a hand-off) and
    under the dynamic tile scheduler (persist = 1)."""
    M, K, splits = shape
    N = 768
    m_valid = M - 41
    ws = torch.zeros(16384 + (M // 256) * 4 * splits * 256 * 192 * 4, dtype=torch.uint8, device="cuda")
    runs = []
    if epi == "dgrad":
        dY = _rand(M, K, seed=401).bfloat16()
        W = _rand(K, N, scale=0.05, seed=402).bfloat16()
        ref = dY.float() @ W.float()
        for rep in range(3):
            out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
            cs = torch.zeros(N, device="cuda")
            _gemm(dY, W, out, M, N, K, K, N, N, 0, 1, EPI_BF16, cfg=4, splits=splits, m_valid=m_valid, colsum=cs, persist=persist,
                  splitk_ws=ws)
            runs.append((out, cs))
        base = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        _gemm(dY, W, base, M, N, K, K, N, N, 0, 1, EPI_BF16, cfg=4, m_valid=m_valid)
        torch.cuda.synchronize()
        out, cs = runs[0]
        scale = ref.abs().max().item()
        assert (out[:m_valid].float() - ref[:m_valid]).abs().max().item() <= scale * 2 ** -7
        # against the un-split kernel: the f32 sums differ by their order only - at most one bf16 step, and rarely
        d = (out[:m_valid].float() - base[:m_valid].float()).abs()
        assert d.max().item() <= scale * 2 ** -7 and (d > 0).float().mean().item() < 0.02
        csr = ref[:m_valid].sum(0)          # (the kernel sums the f32 values in front of the 16-bit rounding)
        assert (cs - csr).abs().max().item() <= 2e-3 * csr.abs().max().item() + 1e-2
        assert bool((out[m_valid:] == 7.0).all())
    else:
        A = _rand(M, K, seed=411).bfloat16()
        W = _rand(N, K, scale=0.05, seed=412).bfloat16()
        bias, res = _rand(N, seed=413), _rand(M, N, seed=414)
        ref = A.float() @ W.float().t() + bias + res
        for rep in range(3):
            out = torch.full((M, N), 7.0, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_RES, cfg=4, splits=splits, m_valid=m_valid, bias=bias, res=res, persist=persist,
                  splitk_ws=ws)
            runs.append((out,))
        base = torch.zeros(M, N, device="cuda")
        _gemm(A, W, base, M, N, K, K, K, N, 0, 0, EPI_RES, cfg=4, m_valid=m_valid, bias=bias, res=res)
        torch.cuda.synchronize()
        out = runs[0][0]
        scale = (A.float() @ W.float().t()).abs().max().item()
        assert (out[:m_valid] - ref[:m_valid]).abs().max().item() <= 2e-4 * scale
        assert (out[:m_valid] - base[:m_valid]).abs().max().item() <= 2e-5 * scale
        assert bool((out[m_valid:] == 7.0).all())
    for r in runs[1:]:      # (the outputs; the column sums are float atomics over the blocks: order-dependent with any kernel)
        assert torch.equal(runs[0][0], r[0])
    assert int(ws[:16384].view(torch.int32).abs().max().item()) == 0


def test_split_k_is_planned_only_with_a_workspace_and_refused_elsewhere():
    """The automatic choice splits a long contraction of few row tiles (cfg 4 + splits) only when the caller lends a workspace
    (it is a no-op for the plan otherwise); a split count on an epilogue / kernel without the in-launch reduction is refused
    instead of writing partial sums."""
    M, N, K = 6144, 768, 3072
    dY = _rand(M, K, seed=421).bfloat16()
    W = _rand(K, N, scale=0.05, seed=422).bfloat16()
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ws = torch.zeros(16384 + 96 * 2 * 256 * 192 * 4, dtype=torch.uint8, device="cuda")
    assert _gemm(dY, W, out, M, N, K, K, N, N, 0, 1, EPI_BF16, plan_only=True, splitk_ws=ws) == 4
    assert _gemm(dY, W, out, M, N, K, K, N, N, 0, 1, EPI_BF16, plan_only=True) in (0, 7)
    _gemm(dY, W, out, M, N, K, K, N, N, 0, 1, EPI_BF16, splitk_ws=ws)            # automatic: two splits
    torch.cuda.synchronize()
    ref = dY.float() @ W.float()
    assert (out.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -7
    for kw in (dict(cfg=4, splits=2), dict(cfg=3, splits=2, splitk_ws=ws), dict(cfg=0, splits=2, splitk_ws=ws),
               dict(cfg=4, splits=2, splitk_ws=ws[:16384 + 1000])):
        with pytest.raises(RuntimeError):
            _gemm(dY, W, out, M, N, K, K, N, N, 0, 1, EPI_BF16, **kw)
    W2 = _rand(3072, 768, scale=0.05, seed=423).bfloat16()
    A2 = _rand(M, 768, seed=424).bfloat16()
    o2 = torch.zeros(M, 3072, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):        # (GELU epilogue: no split-K form)
        _gemm(A2, W2, o2, M, 3072, 768, 768, 768, 3072, 0, 0, EPI_GELU, cfg=4, splits=2, splitk_ws=ws)


@pytest.mark.parametrize("epi", ["dgrad", "res"])
def test_ring_kernel_tail_rows_on_128_wide_tiles(epi):
    """N = 768 at 93 row panels (B = 128): 372 tiles of 256 x 192 are 1.45 rounds; the automatic choice runs 64 panels (one full
    round) on them and the other 29 on 256 x 128 tiles (cfg 8) - bit-identical to the un-cut cfg 4 launch."""
    M, N, K = 93 * 256, 768, 768 if epi == "res" else 2304
    m_valid = M - 120
    A = _rand(M, K, seed=601).bfloat16()
    outs = []
    for cfg in (-1, 4):
        if epi == "dgrad":
            W = _rand(K, N, scale=0.05, seed=602).bfloat16()
            out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device="cuda")
            _gemm(A, W, out, M, N, K, K, N, N, 0, 1, EPI_BF16, cfg=cfg, m_valid=m_valid)
        else:
            W = _rand(N, K, scale=0.05, seed=602).bfloat16()
            bias, res = _rand(N, seed=603), _rand(M, N, seed=604)
            out = torch.full((M, N), 7.0, device="cuda")
            _gemm(A, W, out, M, N, K, K, K, N, 0, 0, EPI_RES, cfg=cfg, m_valid=m_valid, bias=bias, res=res)
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and bool((outs[0][m_valid:] == 7.0).all())
    ref = (A.float() @ W.float()) if epi == "dgrad" else (A.float() @ W.float().t() + bias + res)
    tol = ref.abs().max().item() * (2 ** -7 if epi == "dgrad" else 2e-4)
    assert (outs[0][:m_valid].float() - ref[:m_valid]).abs().max().item() <= tol
