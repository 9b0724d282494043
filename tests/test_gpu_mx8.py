"""MXFP8 forward GEMM and its quantiser (BASELINE config "fp8 MFMA forward, bf16 backward") through the C ABI.

Parity is defined on the format itself: the quantiser against the OCP MX rule restated with torch's CPU
float8_e4m3fn cast (bit-exact), the GEMM against an fp32 matmul of the DEQUANTISED operands (what the matrix pipe
computes, up to fp32 summation order) and, with integer data, exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EPI_BF16, EPI_GELU, EPI_RES = 0, 1, 3


def _quant_ref(x_bf16: torch.Tensor):
    """OCP MX restated on the CPU: per 32 k, scale exponent floor(log2 amax) - 8; e4m3 RNE, saturating."""
    x = x_bf16.float().cpu()
    rows, K = x.shape
    b = x.view(rows, K // 32, 32)
    amax = b.abs().amax(-1)
    ex = torch.where(amax > 0, torch.floor(torch.log2(amax.double())).to(torch.int32) + 127 - 8, torch.zeros_like(amax, dtype=torch.int32))
    e8 = ex.clamp(0, 254)
    inv = torch.pow(torch.tensor(2.0, dtype=torch.float64), (127 - e8).double()).float()
    y = (b * inv[..., None]).clamp(-448.0, 448.0)
    q = y.to(torch.float8_e4m3fn).view(torch.uint8).view(rows, K)
    return q, e8.to(torch.uint8)


def _dequant(q_u8: torch.Tensor, s_u8: torch.Tensor):
    rows, K = q_u8.shape
    v = q_u8.cpu().view(torch.float8_e4m3fn).float().view(rows, K // 32, 32)
    sc = torch.pow(torch.tensor(2.0, dtype=torch.float64), s_u8.cpu().double() - 127.0).float()
    return (v * sc[..., None]).view(rows, K)


def _quant_gpu(x_bf16):
    from vault_amd import ops
    rows, K = x_bf16.shape
    q = torch.empty(rows, K, dtype=torch.uint8, device="cuda")
    s = torch.empty(rows, K // 32, dtype=torch.uint8, device="cuda")
    ops.quant_mxfp8(x_bf16, rows, K, K, q, s)
    torch.cuda.synchronize()
    return q, s


@pytest.mark.parametrize("rows,K,scale", [(64, 128, 1.0), (300, 768, 0.02), (256, 3072, 30.0)])
def test_quantiser_bit_exact_against_the_ocp_rule(rows, K, scale):
    g = torch.Generator().manual_seed(rows + K)
    x = (torch.randn(rows, K, generator=g) * scale * torch.exp(2.0 * torch.randn(rows, 1, generator=g))).bfloat16()
    x[0, :32] = 0                        # an all-zero block
    x[1, 5] = 1e30                       # a huge outlier
    x[2, 40:44] = torch.tensor([448.0, -448.0, 500.0, 2.0 ** -20]).bfloat16()
    q, s = _quant_gpu(x.cuda())
    qr, sr = _quant_ref(x)
    assert torch.equal(s.cpu(), sr)
    assert torch.equal(q.cpu(), qr)


def _run_gemm(q_a, s_a, q_w, s_w, M, N, K, epi, **kw):
    from vault_amd import ops
    f32 = epi == EPI_RES
    out = torch.zeros(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    ops.gemm_mxfp8(q_a, s_a, q_w, s_w, out, M, N, K, N, epi, **kw)
    torch.cuda.synchronize()
    return out


def test_gemm_exact_on_integer_data_with_power_of_two_block_scales():
    """Checks the operand map of v_mfma_scale_f32_16x16x128_f8f6f4 as the kernel uses it: every product is an
    integer times a power of two, so any misplaced element, k block or scale byte changes the result."""
    M, N, K = 512, 256, 384
    g = torch.Generator().manual_seed(5)
    ia = torch.randint(-3, 4, (M, K), generator=g).float()
    iw = torch.randint(-3, 4, (N, K), generator=g).float()
    q_a = ia.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    q_w = iw.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    s_a = torch.randint(125, 130, (M, K // 32), generator=g).to(torch.uint8).cuda()
    s_w = torch.randint(126, 129, (N, K // 32), generator=g).to(torch.uint8).cuda()
    ref = _dequant(q_a, s_a).double() @ _dequant(q_w, s_w).double().t()
    res = torch.zeros(M, N, device="cuda")
    out = _run_gemm(q_a, s_a, q_w, s_w, M, N, K, EPI_RES, res=res)
    assert torch.equal(out.cpu().double(), ref)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (768, 2304, 768), (512, 768, 3072)])
def test_gemm_against_fp32_matmul_of_the_dequantised_operands(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    q_a, s_a = _quant_gpu(x)
    q_w, s_w = _quant_gpu(w)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    m_valid = M - 19
    ref = (_dequant(q_a, s_a).double() @ _dequant(q_w, s_w).double().t()).float() + bias.cpu()
    out = _run_gemm(q_a, s_a, q_w, s_w, M, N, K, EPI_RES, bias=bias, res=res, m_valid=m_valid)
    err = (out.cpu()[:m_valid] - (ref + res.cpu())[:m_valid]).abs().max().item()
    # (the matrix pipe aligns the 128 products of a step before adding them: ~3e-5 of the largest output, more
    #  than an fp32 sum of the same products would lose)
    assert err <= 2e-4 * ref.abs().max().item() + 1e-5, err
    assert out[m_valid:].abs().max().item() == 0.0
    # and how far the format itself is from the bf16 operands (information, loose bound: e4m3 has 3 mantissa bits)
    exact = x.float().cpu().double() @ w.float().cpu().double().t()
    rel = ((ref - bias.cpu()).double() - exact).norm() / exact.norm()
    assert rel < 0.06, rel


def test_gemm_gelu_and_bf16_epilogues():
    M, N, K = 512, 768, 768
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    q_a, s_a = _quant_gpu(x)
    q_w, s_w = _quant_gpu(w)
    bias = torch.randn(N, generator=g).cuda()
    pre = (_dequant(q_a, s_a).double() @ _dequant(q_w, s_w).double().t()).float() + bias.cpu()
    out = _run_gemm(q_a, s_a, q_w, s_w, M, N, K, EPI_BF16, bias=bias)
    assert (out.float().cpu() - pre).abs().max().item() <= pre.abs().max().item() * 2 ** -7
    out2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    y = _run_gemm(q_a, s_a, q_w, s_w, M, N, K, EPI_GELU, bias=bias, out2=out2)
    want = torch.nn.functional.gelu(pre)
    assert (y.float().cpu() - want).abs().max().item() <= want.abs().max().item() * 2 ** -7 + 2e-3
    xs = pre.clone().requires_grad_(True)
    torch.nn.functional.gelu(xs).sum().backward()
    assert (out2.float().cpu() - xs.grad).abs().max().item() <= 1e-2


@pytest.mark.parametrize("M,N,K,epi,cfg", [
    (8192, 2304, 768, EPI_BF16, 5),     # 288 items on 256 CUs: the load cursor crosses into a second item
    (2048, 2304, 768, EPI_BF16, 6),     # 192-wide tiles
    (1024, 3072, 768, EPI_GELU, 5),
    (1024, 3072, 768, EPI_GELU, 6),
    (16640, 768, 3072, EPI_RES, 5),     # 24 K tiles, 195 items
    (1024, 768, 768, EPI_RES, 6),
    (512, 768, 384, EPI_BF16, 5),       # the shortest contraction the form takes: first, one middle, last K tile
])
def test_8wave_form_equals_the_simple_kernel_bit_for_bit(M, N, K, epi, cfg):
    """The 8-wave kernel's MXFP8 form (gemm8w.hip, MX) issues the same scaled MFMAs in the same K order as the simple kernel of
    gemm_mx8.hip, so the accumulators agree to the bit; rows past m_valid are not stored."""
    g = torch.Generator().manual_seed(M + N + K + cfg)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    q_a, s_a = _quant_gpu(x)
    q_w, s_w = _quant_gpu(w)
    bias = torch.randn(N, generator=g).cuda()
    m_valid = M - 37
    kw = dict(bias=bias, m_valid=m_valid)
    if epi == EPI_RES:
        kw["res"] = torch.randn(M, N, generator=g).cuda()
    outs = []
    for c in (0, cfg):
        o2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda") if epi == EPI_GELU else None
        o = _run_gemm(q_a, s_a, q_w, s_w, M, N, K, epi, cfg=c, **kw, **({"out2": o2} if o2 is not None else {}))
        outs.append((o, o2))
    (o0, u0), (o1, u1) = outs
    assert torch.equal(o0[:m_valid], o1[:m_valid])
    assert o1[m_valid:].abs().max().item() == 0.0
    if u0 is not None:
        assert torch.equal(u0[:m_valid], u1[:m_valid])


def test_8wave_gelu_form_with_8bit_gelu_prime_repeats_and_matches_the_simple_kernel():
    """The instantiation in which hipcc had restored the bias pointer with v_readlane_b32 directly in front of the asm bias loads
    (no wait states: garbage bias in whole wave tiles, different ones every run - DESIGN 4.2a).  The build now scans the ISA for the
    pattern; this is the behavioural side: six runs, each equal to the simple kernel's 16-bit output, with and without a bias."""
    from vault_amd import ops
    M, N, K = 1280, 3072, 768
    g = torch.Generator().manual_seed(22)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    q_a, s_a = _quant_gpu(x)
    q_w, s_w = _quant_gpu(w)
    for bias in (torch.randn(N, generator=g).cuda(), None):
        ref = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        ref2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm_mxfp8(q_a, s_a, q_w, s_w, ref, M, N, K, N, EPI_GELU, bias=bias, out2=ref2, cfg=0)
        images = []
        for _ in range(6):
            out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
            u8 = torch.zeros(M * N, dtype=torch.uint8, device="cuda")
            ops.gemm_mxfp8(q_a, s_a, q_w, s_w, out, M, N, K, N, EPI_GELU, bias=bias, out2=u8, cfg=5, aux_u8=True)
            torch.cuda.synchronize()
            assert torch.equal(out, ref)
            images.append(u8)
        assert all(torch.equal(images[0], u) for u in images[1:])


def test_8wave_form_exact_on_integer_data():
    """The operand map, the scale bytes (two per register, selected by op_sel_hi) and the permuted weight rows of the 8-wave
    form: integer data with power-of-two block scales that differ from row to row and block to block."""
    M, N, K = 1024, 768, 512
    g = torch.Generator().manual_seed(6)
    ia = torch.randint(-3, 4, (M, K), generator=g).float()
    iw = torch.randint(-3, 4, (N, K), generator=g).float()
    q_a = ia.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    q_w = iw.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    s_a = torch.randint(124, 131, (M, K // 32), generator=g).to(torch.uint8).cuda()
    s_w = torch.randint(125, 130, (N, K // 32), generator=g).to(torch.uint8).cuda()
    ref = _dequant(q_a, s_a).double() @ _dequant(q_w, s_w).double().t()
    for cfg in (5, 6):
        out = _run_gemm(q_a, s_a, q_w, s_w, M, N, K, EPI_RES, res=torch.zeros(M, N, device="cuda"), cfg=cfg)
        assert torch.equal(out.cpu().double(), ref), cfg
        out = _run_gemm(q_a, s_a, q_w, s_w, M, N, K, EPI_BF16, cfg=cfg)      # (|values| < 2^8 * 2^k: exact in bf16? no - compare rounded)
        assert torch.equal(out.cpu(), ref.float().bfloat16()), cfg


@pytest.mark.parametrize("u8", [False, True])
def test_gelu_epilogue_emits_the_mxfp8_image_of_its_output(u8):
    """vault_gemm_args.out_q / out_scale (FFN-in feeding FFN-out): byte for byte what vault_quant_mxfp8 makes of the 16-bit
    output, beside either form of gelu'; the 16-bit output itself is unchanged by the request."""
    from vault_amd import ops
    M, N, K = 1280, 3072, 768
    g = torch.Generator().manual_seed(21 + u8)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    q_a, s_a = _quant_gpu(x)
    q_w, s_w = _quant_gpu(w)
    bias = torch.randn(N, generator=g).cuda()
    m_valid = M - 5
    def run(emit):
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        out2 = torch.zeros(M * N, dtype=torch.uint8, device="cuda") if u8 else torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        oq = torch.zeros(M, N, dtype=torch.uint8, device="cuda"); osc = torch.zeros(M, N // 32, dtype=torch.uint8, device="cuda")
        kw = dict(out_q=oq, out_scale=osc) if emit else {}
        ops.gemm_mxfp8(q_a, s_a, q_w, s_w, out, M, N, K, N, EPI_GELU, bias=bias, out2=out2, m_valid=m_valid, cfg=5, aux_u8=u8, **kw)
        torch.cuda.synchronize()
        return out, out2, oq, osc
    o0, u0, _, _ = run(False)
    o1, u1, oq, osc = run(True)
    assert torch.equal(o0, o1) and torch.equal(u0, u1)
    q2, s2 = _quant_gpu(o1)
    assert torch.equal(osc[:m_valid], s2[:m_valid])
    assert torch.equal(oq[:m_valid], q2[:m_valid])
    assert oq[m_valid:].abs().max().item() == 0 and osc[m_valid:].abs().max().item() == 0
    with pytest.raises(RuntimeError):      # 192-wide tiles: a block of 32 columns is not one lane quad
        ops.gemm_mxfp8(q_a, s_a, q_w, s_w, o1, M, N, K, N, EPI_GELU, bias=bias, out2=u1, cfg=6, aux_u8=u8, out_q=oq, out_scale=osc)
    with pytest.raises(RuntimeError):      # both pointers or neither
        ops.gemm_mxfp8(q_a, s_a, q_w, s_w, o1, M, N, K, N, EPI_GELU, bias=bias, out2=u1, cfg=5, aux_u8=u8, out_q=oq)


def test_rejects_what_the_kernel_cannot_do():
    from vault_amd import lib as L, ops
    q = torch.zeros(256, 128, dtype=torch.uint8, device="cuda")
    s = torch.zeros(256, 4, dtype=torch.uint8, device="cuda")
    out = torch.zeros(256, 256, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        ops.gemm_mxfp8(q, s, q, s, out, 256, 256, 96, 256, EPI_BF16)       # K % 128
    with pytest.raises(RuntimeError):
        ops.gemm_mxfp8(q, s, q, s, out, 256, 256, 128, 256, 5)             # split-K atomics: backward stays bf16
    with pytest.raises(RuntimeError):
        ops.quant_mxfp8(out, 256, 100, 256, q, s)                          # K % 32
    with pytest.raises(RuntimeError):
        ops.gemm_mxfp8(q, s, q, s, out, 256, 256, 128, 256, EPI_BF16, cfg=5)   # 8-wave form: K >= 384
    with pytest.raises(RuntimeError):
        ops.gemm_mxfp8(q, s, q, s, out, 256, 256, 128, 256, EPI_BF16, cfg=0, out_hm=256)   # head-major: 8-wave form only


# ---- the fp8-forward configuration of the engine ---------------------------------------------------------------

def _tiny_engine(fp8, kind="roberta", seed=0):
    from vault_amd.engine import VaultEngine
    from vault_amd.spec import VaultSpec
    spec = VaultSpec.tiny(3, kind)
    spec.vilt.hidden_dropout_prob = 0.0
    spec.vilt.attention_probs_dropout_prob = 0.0
    if spec.lm is not None:
        spec.lm.hidden_dropout_prob = 0.0
        spec.lm.attention_probs_dropout_prob = 0.0
    return spec, VaultEngine(spec, "cuda:0", seed=seed, classifier_dropout=0.0, fp8_forward=fp8, half="bf16")


def test_engine_fp8_forward_tracks_the_bf16_path_and_backward_stays_bf16():
    """Forward Linears on MXFP8 operands: outputs stay close to the bf16 engine (e4m3 has 3 mantissa bits: per-GEMM
    error ~3 %, averaged down over the contraction), the loss matches to a few 1e-2, and the gradients - computed by
    the unchanged bf16 backward from the fp8-forward activations - point the same way."""
    from vault_amd.spec import synthetic_batch
    spec, e16 = _tiny_engine(False)
    _, e8 = _tiny_engine(True)
    bn = synthetic_batch(spec, 3, seed=11, n_classes=3)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items()}
    outs = []
    for eng in (e16, e8):
        out = eng.forward(db, train=True, labels=db["labels"], need_hidden=True)
        eng.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in out.items() if isinstance(v, torch.Tensor)})
    assert e8._w8, "no Linear took the MXFP8 path"
    ref, got = outs
    assert (got["logits"] - ref["logits"]).abs().max().item() < 0.05 * ref["logits"].abs().max().item() + 2e-2
    assert abs(float(got["loss"]) - float(ref["loss"])) < 3e-2
    hr, hg = ref["last_hidden_state"].float(), got["last_hidden_state"].float()
    assert float((hg - hr).norm() / hr.norm()) < 0.05
    n = e16.params.n_train
    g16, g8 = e16.params.g[:n].double(), e8.params.g[:n].double()
    cos = float((g16 * g8).sum() / (g16.norm() * g8.norm()))
    assert cos > 0.97, cos
    assert 0.8 < float(g8.norm() / g16.norm()) < 1.25


@pytest.mark.parametrize("ffn_out", [False, True])
def test_engine_fp8_forward_full_size_against_reference_golden(ffn_out):
    """ViLT-B32 + bertweet-base shapes, B = 2, eval: against the fp32 reference golden with the tolerance the format
    allows (logits of this random-weight model are O(0.1): absolute bound)."""
    import os
    from vault_amd.engine import VaultEngine
    from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, synthetic_batch
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "full_bertweet_b2.npz"))
    spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=3)
    eng = VaultEngine(spec, "cuda:0", seed=0, classifier_dropout=0.0, fp8_forward=True, with_grads=False, half="bf16")
    eng.FFN_OUT_FP8 = ffn_out      # (measured slower in the step and off by default: engine._fp8_refresh_weights)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items()}
    out = eng.forward(db, train=False, need_hidden=True)
    torch.cuda.synchronize()
    assert len(eng._w8) == (3 if ffn_out else 2) * 24          # QKV, FFN-in (and FFN-out) of the 12 + 12 layers
    assert np.abs(out["logits"].cpu().numpy() - g["logits"]).max() < 6e-2
    T = bn["input_ids"].shape[1]
    h = out["last_hidden_state"][:, : T + 1].cpu().numpy()
    rel = np.linalg.norm(h - g["hidden_text_cls"]) / np.linalg.norm(g["hidden_text_cls"])
    # every GEMM output carries ~3 % of format noise (two e4m3 operands, random-sign sums do not average a relative
    # error down); over 12 + 12 layers of this random-weight model it accumulates to ~10 % of the hidden state
    assert rel < 0.15, rel


def test_train_step_with_fp8_forward_learns_a_fixed_batch():
    from vault_amd.spec import synthetic_batch
    from vault_amd.train import TrainStep
    spec, eng = _tiny_engine(True, seed=3)
    bn = synthetic_batch(spec, 8, seed=5, n_classes=3)
    step = TrainStep(eng, learning_rate=2e-4, warmup_ratio=0.0, total_steps=1000, constant_lr=True)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    losses = [float(step(db, labels)) for _ in range(40)]
    assert max(losses[-5:]) < 0.6 * losses[0], losses[::6]


def test_layernorm_emits_the_mxfp8_image_of_its_bf16_output():
    """vault_layernorm_fwd(y_q, y_scale): byte for byte what vault_quant_mxfp8 gives for y_bf16."""
    from vault_amd import ops
    rows, H = 777, 768
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(rows, H, generator=g) * torch.exp(torch.randn(rows, 1, generator=g))).cuda()
    gamma = (1.0 + 0.1 * torch.randn(H, generator=g)).cuda(); beta = (0.1 * torch.randn(H, generator=g)).cuda()
    yb = torch.zeros(rows, H, dtype=torch.bfloat16, device="cuda")
    q = torch.zeros(rows, H, dtype=torch.uint8, device="cuda"); s = torch.zeros(rows, H // 32, dtype=torch.uint8, device="cuda")
    ops.layernorm_fwd(x, gamma, beta, 1e-12, rows, H, y_bf16=yb, y_q=q, y_scale=s)
    q2, s2 = _quant_gpu(yb)
    assert torch.equal(s, s2) and torch.equal(q, q2)


def test_model_class_switch():
    """`model.fp8_forward = True` on the reference-shaped classes selects the MXFP8 forward."""
    from vault_amd.models.vault import VaultForTMSC
    from vault_amd.spec import VaultSpec, synthetic_batch
    spec = VaultSpec.tiny(3, "roberta")
    bn = synthetic_batch(spec, 3, seed=11, n_classes=3)
    model = VaultForTMSC(spec.vilt, n_classes=3, vilt_dropout_prob=0.0, bert_config=spec.lm)
    model.half_format = "bf16"            # (the MXFP8 forward quantises bf16 operands; the default format is fp16)
    model = model.to("cuda").eval()
    kw = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    with torch.no_grad():
        a = model(**kw).clone()
        model.fp8_forward = True
        b = model(**kw).clone()
        model.fp8_forward = False
        c = model(**kw).clone()
    assert model._engine._w8 and torch.equal(a, c)
    assert 0 < (a - b).abs().max().item() < 0.05 * a.abs().max().item() + 2e-2


def test_fp8_forward_keeps_head_major_qkv_and_8bit_gelu_prime():
    """fp8-forward mode at a batch where the engine plans the head-major qkv layout and the 8-bit gelu' (full width, 2 + 2 layers,
    B = 208): the plans are taken (vault_gemm_mxfp8_plan says the 8-wave form runs) and change addresses / the storage of gelu'
    only - logits bit-identical to the same engine with both plans off, gradients equal up to the 8-bit rounding of gelu'
    (step 0.005) and summation order."""
    from vault_amd.engine import VaultEngine
    from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, build_state, synthetic_batch
    spec = VaultSpec(vilt=ViltSpec(num_hidden_layers=2), lm=LMSpec.bertweet_base(), n_classes=3)
    spec.lm.num_hidden_layers = 2
    state = build_state(spec, 3)
    bn = synthetic_batch(spec, 208, seed=708, n_classes=3)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items()}
    res = []
    for plans in (True, False):
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.1, half="bf16", fp8_forward=True)
        eng.HEAD_MAJOR, eng.GELU8, eng.HEAD_MAJOR_MIN_ROWS = plans, plans, 0
        eng.drop_seed = 77
        out = eng.forward(db, train=True, labels=db["labels"], need_hidden=False)
        eng.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        assert (eng.last["qkv_hm"], eng.last["lm_qkv_hm"]) == ((eng.last["Mp"], eng.last["Mlp"]) if plans else (0, 0))
        assert (eng.last["gelu8_active"] is not None) == plans
        assert len(eng._w8) == 2 * 4
        res.append((out["logits"].clone(), float(out["loss"]), eng.params.g[:eng.params.n_train].clone()))
        del eng
    a, b = res
    assert torch.equal(a[0], b[0])
    assert abs(a[1] - b[1]) < 1e-6
    rel = float((a[2] - b[2]).norm() / b[2].norm())
    print(f"fp8 forward, plans on vs off: gradient rel diff {rel:.2e}")
    assert rel < 5e-3
