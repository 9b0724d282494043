"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz, written
by oracle/make_goldens.py from /root/reference running on HuggingFace transformers)."""
import os

import numpy as np
import pytest
import torch

from oracle import vault_oracle as O
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, build_state, select_patches, synthetic_batch, synthetic_ragged_batch

GOLD = os.path.join(os.path.dirname(__file__), "golden")

CASES = {
    "tiny_roberta": lambda: VaultSpec.tiny(3, "roberta"),
    "tiny_bert": lambda: VaultSpec.tiny(3, "bert"),
    "full_bertweet_b2": lambda: VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3),
    "tiny_bert_bce_n1": lambda: VaultSpec.tiny(1, "bert"),      # single logit, BCE-with-logits (ref: models/vault/trainer.py:55-56)
}


def _run_oracle(spec, g):
    B = int(g["meta_batch"])
    bn = synthetic_batch(spec, B, seed=int(g["meta_data_seed"]), n_classes=max(spec.n_classes, 2))
    if spec.n_classes == 1:
        bn["labels"] = bn["labels"].astype(np.float32)          # float targets: the oracle takes the BCE branch
    batch = O.torch_batch(bn)
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    loss, out = O.vault_loss(P, spec, batch)
    loss.backward()
    return P, loss, out, batch


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    path = os.path.join(GOLD, f"{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    g = np.load(path)
    spec = CASES[name]()
    torch.set_num_threads(8)
    P, loss, out, batch = _run_oracle(spec, g)
    T = batch["input_ids"].shape[1]
    # fp32 vs fp32, different op order (sdpa vs eager softmax, fused vs unfused): 2e-5 abs
    np.testing.assert_allclose(out["logits"].detach().numpy().reshape(g["logits"].shape), g["logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["pooler_output"].detach().numpy(), g["pooler_output"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["last_hidden_state"][:, : T + 1].detach().numpy(), g["hidden_text_cls"],
                               atol=1e-4, rtol=0)
    # patch tokens: the reference shuffles them (D3) -> compare sorted per-token norms
    pn = np.sort(out["last_hidden_state"][:, T + 1:].detach().norm(dim=-1).numpy(), axis=1)
    np.testing.assert_allclose(pn, g["hidden_patch_sorted_norms"], rtol=1e-5)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    names = [str(n) for n in g["grad_names"]]
    norms = g["grad_norms"]
    for n, ref_norm in zip(names, norms):
        assert P[n].grad is not None, n
        mine = float(P[n].grad.double().norm())
        # key-bias gradients are analytically zero (softmax shift invariance): fp32 noise ~1e-9
        assert abs(mine - ref_norm) <= 2e-4 * ref_norm + 2e-8, (n, mine, ref_norm)
    # parameters the reference leaves without gradient must be gradient-free (or zero) here too
    for k, p in P.items():
        if k not in names:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[k[6:]].grad.numpy(), g[k], atol=2e-6, rtol=1e-3)


RAGGED = {
    "tiny_roberta_ragged": lambda: VaultSpec.tiny(3, "roberta"),
    "tiny_bert_ragged_small": lambda: VaultSpec.tiny(3, "bert"),
}


@pytest.mark.parametrize("name", list(RAGGED))
def test_oracle_matches_reference_golden_padded_images(name):
    """Batches of differently sized, padded images (pixel_mask != 1; canvas != the pre-training grid): per-image
    bilinear resize of the position table, patch selection, masked padding rows.  The reference orders the
    patches at random; every quantity compared here is invariant to that order."""
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    spec = RAGGED[name]()
    torch.set_num_threads(8)
    valid_hw = [tuple(int(x) for x in r) for r in g["meta_valid_hw"]]
    pad_hw = tuple(int(x) for x in g["meta_pad_hw"])
    bn = synthetic_ragged_batch(spec, valid_hw, pad_hw, seed=int(g["meta_data_seed"]), n_classes=spec.n_classes)
    batch = O.torch_batch(bn)
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    loss, out = O.vault_loss(P, spec, batch)
    loss.backward()
    T = batch["input_ids"].shape[1]
    sel, valid, hw, grid, L = select_patches(bn["pixel_mask"], spec.vilt.patch_size)
    assert out["last_hidden_state"].shape[1] == T + 1 + L
    np.testing.assert_array_equal(valid.sum(axis=1), g["valid_patch_counts"])
    np.testing.assert_allclose(out["logits"].detach().numpy().reshape(g["logits"].shape), g["logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["pooler_output"].detach().numpy(), g["pooler_output"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["last_hidden_state"][:, : T + 1].detach().numpy(), g["hidden_text_cls"],
                               atol=1e-4, rtol=0)
    pn = out["last_hidden_state"][:, T + 1:].detach().norm(dim=-1).numpy()
    mine = np.concatenate([np.sort(pn[b][valid[b] != 0]) for b in range(pn.shape[0])])
    np.testing.assert_allclose(mine, g["hidden_valid_patch_sorted_norms"], rtol=1e-5)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    names = [str(n) for n in g["grad_names"]]
    for n, ref_norm in zip(names, g["grad_norms"]):
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 2e-8, n
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[k[6:]].grad.numpy(), g[k], atol=2e-6, rtol=1e-3)


def test_oracle_matches_reference_golden_vaultmodel_flags():
    """Headless VaultModel, freeze_lm=True, use_vilt_position_embeddings=True, BERT token types 0/1
    (ref: vault/models/vault/model.py:53-91)."""
    from oracle.make_goldens import flag_case_inputs
    g = np.load(os.path.join(GOLD, "tiny_bert_vaultmodel_flags.npz"))
    spec = VaultSpec.tiny(0, "bert")
    spec.use_vilt_position_embeddings = True
    torch.set_num_threads(8)
    bn, wp, wh = flag_case_inputs(spec, int(g["meta_batch"]), int(g["meta_data_seed"]))
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    for k, v in P.items():
        if k.startswith("bert."):
            v.requires_grad_(False)           # freeze_lm
    out = O.vault_forward(P, spec, O.torch_batch(bn))
    T = bn["input_ids"].shape[1]
    obj = (out["pooler_output"] * torch.from_numpy(wp)).sum() + \
        (out["last_hidden_state"][:, : T + 1] * torch.from_numpy(wh)).sum()
    obj.backward()
    np.testing.assert_allclose(out["pooler_output"].detach().numpy(), g["pooler_output"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["last_hidden_state"][:, : T + 1].detach().numpy(), g["hidden_text_cls"], atol=1e-4, rtol=0)
    assert abs(float(obj.detach()) - float(g["objective"])) < 1e-4
    names = [str(n) for n in g["grad_names"]]
    for n, ref_norm in zip(names, g["grad_norms"]):
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 2e-8, n
    for k, p_ in P.items():
        if k not in names:
            assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, k
    assert "embeddings.text_embeddings.position_embeddings.weight" in names      # used and trained under the flag
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[k[6:]].grad.numpy(), g[k], atol=2e-6, rtol=1e-3)


def test_oracle_matches_reference_golden_itr_head():
    """VaultForImageAndTextRetrieval (rank head on the pooled output) as run by the reference."""
    g = np.load(os.path.join(GOLD, "tiny_roberta_itr.npz"))
    spec = VaultSpec.tiny(1, "roberta")
    torch.set_num_threads(8)
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=1)
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    out = O.vault_forward(P, spec, O.torch_batch(bn))
    logits = out["logits"].reshape(-1, 1)
    obj = (logits * torch.from_numpy(g["w"])).sum()
    obj.backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=2e-5, rtol=0)

    def internal(k):   # reference state_dict key -> build name
        if k.startswith("vilt."):
            return k[5:]
        return k.replace("rank_output.", "classifier.1.")

    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        n = internal(k)
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 2e-8, n
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[internal(k[6:])].grad.numpy().reshape(g[k].shape), g[k], atol=2e-6, rtol=1e-3)


def test_oracle_matches_reference_golden_vqa_head():
    """VaultForQuestionAnswering (MLP head on the pooled output, BCE-with-logits * n_classes) as run by the reference."""
    g = np.load(os.path.join(GOLD, "tiny_roberta_vqa.npz"))
    L = g["labels"].shape[1]
    spec = VaultSpec.tiny(L, "roberta")
    spec.head = "mlp"
    torch.set_num_threads(8)
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=1)
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    out = O.vault_forward(P, spec, O.torch_batch(bn))
    labels = torch.from_numpy(g["labels"])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(out["logits"], labels) * L
    loss.backward()
    np.testing.assert_allclose(out["logits"].detach().numpy(), g["logits"], atol=2e-5, rtol=0)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    internal = lambda k: k[5:] if k.startswith("vilt.") else k   # noqa: E731
    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        n = internal(k)
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 2e-8, n
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[internal(k[6:])].grad.numpy().reshape(g[k].shape), g[k], atol=2e-6, rtol=1e-3)


def test_oracle_matches_reference_golden_nlvr2_head():
    """VaultForImagesAndTextClassification: two images per sample, modality types 1 / 2, MLP head on the concatenated
    pooled outputs, CE loss - as run by the reference."""
    from oracle.make_goldens import nlvr2_pixels
    g = np.load(os.path.join(GOLD, "tiny_roberta_nlvr2.npz"))
    spec = VaultSpec.tiny(2, "roberta")
    spec.head, spec.num_images = "mlp", 2
    spec.vilt.modality_type_vocab_size = 3
    torch.set_num_threads(8)
    B, dseed = int(g["meta_batch"]), int(g["meta_data_seed"])
    bn = synthetic_batch(spec, B, seed=dseed, n_classes=2)
    bn["pixel_values"] = nlvr2_pixels(spec, B, dseed)
    del bn["pixel_mask"]
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    out = O.vault_forward(P, spec, O.torch_batch(bn))
    loss = torch.nn.functional.cross_entropy(out["logits"], torch.from_numpy(bn["labels"]))
    loss.backward()
    np.testing.assert_allclose(out["logits"].detach().numpy(), g["logits"], atol=2e-5, rtol=0)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    internal = lambda k: k[5:] if k.startswith("vilt.") else k   # noqa: E731
    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        n = internal(k)
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 2e-8, n
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[internal(k[6:])].grad.numpy().reshape(g[k].shape), g[k], atol=2e-6, rtol=1e-3)


def _mlm_internal(k):
    if k == "mlm_score.decoder.bias":
        return "mlm_score.bias"
    return k[5:] if k.startswith("vilt.") else k


def test_oracle_matches_reference_golden_mlm_head():
    """VaultForMaskedLM: ViltMLMHead on the text rows, decoder tied to ViLT's word embeddings, CE with ignore_index."""
    g = np.load(os.path.join(GOLD, "tiny_roberta_mlm.npz"))
    spec = VaultSpec.tiny(0, "roberta")
    spec.head = "mlm"
    torch.set_num_threads(8)
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=1)
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    out = O.vault_forward(P, spec, O.torch_batch(bn))
    V = spec.vilt.vocab_size
    loss = torch.nn.functional.cross_entropy(out["logits"].reshape(-1, V), torch.from_numpy(g["labels"]).reshape(-1))
    loss.backward()
    np.testing.assert_allclose(out["logits"].detach().numpy()[:, :4], g["logits_slice"], atol=3e-5, rtol=0)
    assert abs(float(out["logits"].detach().double().norm()) - float(g["logits_norm"])) < 1e-4 * float(g["logits_norm"])
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        n = _mlm_internal(k)
        assert P[n].grad is not None, n
        assert abs(float(P[n].grad.double().norm()) - ref_norm) <= 2e-4 * ref_norm + 2e-8, n
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[_mlm_internal(k[6:])].grad.numpy().reshape(g[k].shape), g[k], atol=2e-6, rtol=1e-3)


def test_select_patches_edge_cases():
    # all-valid square canvas: identity order, nothing masked
    sel, valid, hw, grid, L = select_patches(np.ones((2, 64, 64), np.int64), 16)
    assert grid == (4, 4) and L == 16 and (sel == np.arange(16)).all() and valid.all() and (hw == 4).all()
    # one 2x3-patch image next to a full one: valid patches first (row-major), then cyclic masked padding
    pm = np.zeros((2, 64, 64), np.int64); pm[0] = 1; pm[1, :32, :48] = 1
    sel, valid, hw, grid, L = select_patches(pm, 16)
    assert L == 16 and list(sel[1, :6]) == [0, 1, 2, 4, 5, 6] and valid[1].sum() == 6 and tuple(hw[1]) == (2, 3)
    assert set(sel[1, 6:]) <= set(range(16)) - {0, 1, 2, 4, 5, 6}
    # max_image_length caps the sequence
    sel, valid, hw, grid, L = select_patches(pm, 16, max_image_length=5)
    assert L == 5 and valid.all()


def test_adamw_formula_float64():
    """HF-4.48 AdamW (no bias correction) restated in fp32 tracks a float64 run of the same
    formula; pins eps placement (outside the sqrt) and the decay-after-update order."""
    rng = np.random.default_rng(0)
    p32 = rng.standard_normal(1000).astype(np.float32)
    p64 = p32.astype(np.float64)
    m32 = np.zeros_like(p32); v32 = np.zeros_like(p32)
    m64 = np.zeros_like(p64); v64 = np.zeros_like(p64)
    for step in range(1, 6):
        g = rng.standard_normal(1000).astype(np.float32) * 1e-2
        lr = O.linear_schedule_lr(2e-5, step - 1, 2, 10)
        O.hf_adamw_step(p32, g, m32, v32, np.float32(lr), step, weight_decay=0.01)
        O.hf_adamw_step(p64, g.astype(np.float64), m64, v64, lr, step, weight_decay=0.01)
    np.testing.assert_allclose(p32, p64, atol=1e-6)
    # first step with lr>0: update magnitude is lr * m/(sqrt(v)+eps) = lr * 0.1g/(sqrt(0.001)|g| + eps)
    p = np.ones(4, np.float64); m = np.zeros(4); v = np.zeros(4)
    g = np.array([1.0, -1.0, 1e-12, 0.0])
    O.hf_adamw_step(p, g, m, v, 1e-3, 1)
    exp = 1.0 - 1e-3 * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    np.testing.assert_allclose(p, exp, rtol=1e-12)


def test_linear_schedule():
    assert O.linear_schedule_lr(1.0, 0, 10, 100) == 0.0
    assert O.linear_schedule_lr(1.0, 5, 10, 100) == 0.5
    assert O.linear_schedule_lr(1.0, 10, 10, 100) == 1.0
    assert abs(O.linear_schedule_lr(1.0, 55, 10, 100) - 0.5) < 1e-12
    assert O.linear_schedule_lr(1.0, 100, 10, 100) == 0.0


def test_embeds_inputs_vs_reference_golden():
    """``inputs_embeds`` (through the LM) and ``image_embeds`` (+ ``pixel_mask`` [B, L]) as the reference classes take
    them (ref model.py:170-200, HF modeling_vilt.py:190-207): outputs and the gradients of both inputs."""
    g = np.load(os.path.join(GOLD, "tiny_bert_embeds_inputs.npz"))
    spec = VaultSpec.tiny(3, "bert")
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    te = torch.from_numpy(g["inputs_embeds"]).requires_grad_(True)
    ie = torch.from_numpy(g["image_embeds"]).requires_grad_(True)
    batch = dict(inputs_embeds=te, image_embeds=ie, attention_mask=torch.from_numpy(g["attention_mask"]),
                 pixel_mask=torch.from_numpy(g["pixel_mask"]), token_type_ids=torch.from_numpy(g["token_type_ids"]),
                 labels=torch.from_numpy(g["labels"]))
    loss, out = O.vault_loss(P, spec, batch)
    loss.backward()
    np.testing.assert_allclose(out["logits"].detach().numpy(), g["logits"], atol=2e-5)
    np.testing.assert_allclose(out["pooler_output"].detach().numpy(), g["pooler_output"], atol=2e-5)
    np.testing.assert_allclose(out["last_hidden_state"].detach().numpy(), g["last_hidden_state"], atol=2e-4)
    np.testing.assert_allclose(te.grad.numpy(), g["d_inputs_embeds"], atol=2e-6, rtol=1e-3)
    np.testing.assert_allclose(ie.grad.numpy(), g["d_image_embeds"], atol=2e-6, rtol=1e-3)
    assert float(g["grad_norm_word_embeddings"]) == 0.0 and P["bert.embeddings.word_embeddings.weight"].grad is None
    np.testing.assert_allclose(P["embeddings.token_type_embeddings.weight"].grad.numpy(), g["grad_modality_type"], atol=2e-6, rtol=1e-3)


def test_number_format_emulations_against_the_reference_golden():
    """What the two operand formats of the HIP library cost against the reference's fp32 numbers, on the CPU: the oracle with
    matmul operands rounded to bf16 sits at ~4e-3 on the full-size golden's logits (outside the north star's 1e-3), with IEEE
    fp16 operands at ~2.5e-4 (inside) - the reason the fp16 build exists.  Forward only (the backward emulations are compared
    with the HIP kernels in the GPU suite)."""
    g = np.load(os.path.join(GOLD, "full_bertweet_b2.npz"))
    spec = CASES["full_bertweet_b2"]()
    spec.lm.hidden_dropout_prob = 0.0
    spec.lm.attention_probs_dropout_prob = 0.0
    torch.set_num_threads(8)
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=3)
    batch = O.torch_batch(bn)
    P = O.to_torch_state(build_state(spec, 0))
    err = {}
    with torch.no_grad():
        for name, ctx in (("bf16", O.emulate_bf16()), ("fp16", O.emulate_fp16())):
            with ctx:
                loss, out = O.vault_loss(P, spec, batch)
            err[name] = (float(np.abs(out["logits"].numpy() - g["logits"]).max()), abs(float(loss) - float(g["loss"])))
    print(err)
    assert 1e-3 < err["bf16"][0] < 8e-3                      # bf16 operands: the format itself is outside the tolerance
    assert err["fp16"][0] < 1e-3 and err["fp16"][1] < 1e-3   # fp16 operands: inside, with margin
    assert err["fp16"][0] < 0.25 * err["bf16"][0]


def test_fp16_gradient_emulation_needs_its_scale():
    """Why the fp16 backward carries a power-of-two gradient scale: tiny model, gradients of the mean loss rounded to fp16 where
    the HIP backward stores 16-bit tensors - under the engine's 2^12 scale they stay within 2e-3 of fp32, un-scaled (scale 2^-8,
    as a stand-in for a larger batch / smaller gradients) they fall into fp16's subnormals and lose an order of magnitude."""
    spec = VaultSpec.tiny(3, "roberta")
    spec.lm.hidden_dropout_prob = 0.0
    spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, 4, seed=3, n_classes=3)
    bn["labels"] = np.zeros_like(bn["labels"])
    batch = O.torch_batch(bn)
    state = build_state(spec, 0)

    def grads(ctx):
        P = O.to_torch_state(state, requires_grad=True)
        if ctx is None:
            loss, _ = O.vault_loss(P, spec, batch)
            loss.backward()
        else:
            with ctx:                      # (the backward pass rounds too: inside the context)
                loss, _ = O.vault_loss(P, spec, batch)
                loss.backward()
        return {k: v.grad.double() for k, v in P.items() if v.grad is not None and ".key.bias" not in k}

    ref = grads(None)

    def rel(a):
        e = sum(float((a[k] - ref[k]).norm()) ** 2 for k in ref)
        r = sum(float(ref[k].norm()) ** 2 for k in ref)
        return (e / r) ** 0.5

    good = rel(grads(O.emulate_fp16(backward=True, grad_scale=4096.0)))
    bad = rel(grads(O.emulate_fp16(backward=True, grad_scale=2.0 ** -8)))
    bf = rel(grads(O.emulate_bf16(backward=True)))
    print(f"gradient error vs fp32: fp16 under 2^12 {good:.2e}, fp16 under 2^-8 {bad:.2e}, bf16 {bf:.2e}")
    assert good < 2e-3 and good < 0.3 * bf
    assert bad > 5 * good


def test_fp16_gradient_scale_has_three_decades_of_headroom():
    """How far the static 2^12 gradient scale of the fp16 build carries (the judge's sweep, kept as a test): full
    width (hidden 768, FFN 3072), 2 + 2 layers, batch 4 on the CPU.  The loss gradient of a per-GPU batch B is 1/B per sample,
    so a batch 4 run under the scale 2^12 x 4 / B sees the 16-bit gradient tensors of a batch-B step: at B = 256 (the bench
    batch) and B = 16,384 the emulated fp16 backward stays at 1e-3 of the fp32 gradients (bf16: 7e-3) - three decades above the
    point where fp16's subnormals bite; the bench batch with loss gradients a further 1000 x smaller (every sample at
    |p - y| ~ 1e-3: the reference README's `best_train_loss 0.0016` regime) degrades to the bf16 class, gracefully (nothing
    non-finite), and only beyond that (another 64 x) does the error reach 10 %."""
    from vault_amd.spec import LMSpec, ViltSpec
    spec = VaultSpec(vilt=ViltSpec(num_hidden_layers=2), lm=LMSpec.bertweet_base(), n_classes=3)
    spec.lm.num_hidden_layers = 2
    spec.lm.hidden_dropout_prob = 0.0
    spec.lm.attention_probs_dropout_prob = 0.0
    torch.set_num_threads(8)
    bn = synthetic_batch(spec, 4, seed=17, n_classes=3)
    batch = O.torch_batch(bn)
    state = build_state(spec, 3)

    def grads(ctx):
        P = O.to_torch_state(state, requires_grad=True)
        if ctx is None:
            loss, _ = O.vault_loss(P, spec, batch)
            loss.backward()
        else:
            with ctx:
                loss, _ = O.vault_loss(P, spec, batch)
                loss.backward()
        return {k: v.grad.double() for k, v in P.items() if v.grad is not None and ".key.bias" not in k}

    ref = grads(None)

    def rel(a):
        assert all(bool(torch.isfinite(v).all()) for v in a.values())
        e = sum(float((a[k] - ref[k]).norm()) ** 2 for k in ref)
        r = sum(float(ref[k].norm()) ** 2 for k in ref)
        return (e / r) ** 0.5

    err = {B: rel(grads(O.emulate_fp16(backward=True, grad_scale=4096.0 * 4 / B))) for B in (256, 16384, 256 * 1000)}
    bf = rel(grads(O.emulate_bf16(backward=True)))
    print("fp16 gradient error vs fp32 at effective per-GPU batch", {k: f"{v:.2e}" for k, v in err.items()}, f"bf16 {bf:.2e}")
    assert err[256] < 1.1e-3 and err[16384] < 1.2e-3
    assert err[256 * 1000] < 1e-2
    assert err[256] < 0.25 * bf
