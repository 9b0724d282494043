"""GPU parity of the whole stacked BERT -> ViLT path (HIP engine through the C ABI) against
  (a) the CPU oracle in fp32 (the reference's arithmetic),
  (b) the CPU oracle with bf16-rounded matmul operands (the HIP path's number format), and
  (c) the committed golden vectors written by the reference itself (tests/golden/*.npz).

Tolerances (bf16 MFMA operands, fp32 accumulation / residual stream / LayerNorm / softmax):
  logits: 3e-3 abs at 2+2 layers, 8e-3 abs at 12+12 layers (|logits| ~ 0.2); loss: 2e-3 abs;
  hidden states: 1% of their max magnitude; gradients: global relative L2 error <= 6e-2 and cosine
  >= 0.995 (the bf16-operand oracle itself sits at 3e-3 .. 4.5e-2 from the fp32 oracle on these cases).
"""
import os

import numpy as np
import pytest
import torch

from oracle import vault_oracle as O
from vault_amd.engine import VaultEngine
from vault_amd.spec import (LMSpec, VaultSpec, ViltSpec, build_state, select_patches, synthetic_batch,
                            synthetic_ragged_batch)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _nodrop(spec):
    if spec.lm is not None:
        spec.lm.hidden_dropout_prob = 0.0
        spec.lm.attention_probs_dropout_prob = 0.0
    return spec


def _dev(bn):
    return {k: torch.from_numpy(v).cuda() for k, v in bn.items()}


def _grad_errors(eng, P):
    """HIP gradients against an oracle's: global relative L2 error and the per-parameter relative errors (largest first);
    the key biases are left out (their gradient is analytically zero: softmax is invariant to a per-query shift)."""
    tot_e = tot_r = 0.0
    per = []
    for n in eng.params.trainable:
        gr = P[n].grad
        if gr is None or ".key.bias" in n:
            continue
        mine = eng.params.gr(n).cpu().double().reshape(gr.shape)
        e, r = float((mine - gr.double()).norm()), float(gr.double().norm())
        tot_e += e * e; tot_r += r * r
        per.append((e / (r + 1e-30), n))
    return (tot_e / tot_r) ** 0.5, sorted(per, reverse=True)


def _emulated_backward(spec, state, bn, gelu8=False, inject=None, half="bf16"):
    """Oracle loss + gradients in the HIP path's own number format, forward AND backward (oracle.emulate_bf16(backward=True),
    or emulate_fp16 for the fp16 operand build); ``inject`` = (weight-name substring, factor): the mutation check's scaled
    data gradient."""
    P = O.to_torch_state(state, requires_grad=True)
    O._INJECT_DGRAD = inject
    try:
        with (O.emulate_bf16 if half == "bf16" else O.emulate_fp16)(backward=True, gelu8=gelu8):
            loss, ref = O.vault_loss(P, spec, O.torch_batch(bn))
            loss.backward()
    finally:
        O._INJECT_DGRAD = None
    return P, float(loss.detach())


# per-parameter relative error bounds of the same-format gradient comparison, by class (measured + margin; the tests print
# the measured values)
SAME_FORMAT_CLASS_BOUNDS = {"layernorm": 6e-3, "attention q/k": 1.5e-2, "other": 1e-2}
SAME_FORMAT_DEEP_BOUNDS = {"layernorm": 1.3e-2, "attention q/k": 3e-2, "other": 1.8e-2}     # 12 + 12 layers (see the tests)
SAME_FORMAT_B48_BOUNDS = {"layernorm": 2.8e-3, "attention q/k": 9.5e-3, "other": 1.5e-2}     # 12 + 12 layers at B = 48: measured + 50 %


def _param_class(n):
    """Gradient classes by conditioning: LayerNorm scales / shifts and the FFN / output matrices are sums of same-signed-ish
    large terms; the attention query / key gradients of a random-init model are differences of nearly equal numbers (the
    softmax is almost uniform: dS is tiny against P dP), relatively the noisiest."""
    if "ayer" in n and ("norm" in n.lower()):
        return "layernorm"
    if ".query." in n or ".key." in n:
        return "attention q/k"
    return "other"


def _assert_same_format_gradients(spec, state, bn, tag, glob_bound, class_bounds, gelu8=None,
                                  mutate="encoder.layer.1.attention.attention.qkv", half="bf16", mutate_bound=None):
    """The backward pinned to its own number format.  A fresh HIP forward + backward of `bn` with every label set to class 0
    (per-sample gradients then add up instead of cancelling: a relative bound means something) against the oracle emulating
    the HIP number format forward AND backward (oracle.emulate_bf16(backward=True)): global relative L2 <= glob_bound and per
    parameter <= its class bound - and those bounds notice a 1 % error in ONE data-gradient GEMM: the same comparison against
    an oracle whose `mutate` dgrad is scaled by 1.01 breaks the LayerNorm-class bound (that dgrad feeds the LayerNorm below it)."""
    bn = dict(bn)
    bn["labels"] = np.zeros_like(bn["labels"])
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half=half)
    db = _dev(bn)
    eng.forward(db, train=True, labels=db["labels"], need_hidden=False)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    if gelu8 is None:
        gelu8 = eng.last.get("gelu8_active") is not None
    Pe, _ = _emulated_backward(spec, state, bn, gelu8, half=half)
    glob, per = _grad_errors(eng, Pe)
    worst = {}
    for e, n in per:
        worst.setdefault(_param_class(n), (e, n))
    print(f"{tag}: gradients vs the {half}-emulating oracle (forward + backward): global rel L2 {glob:.2e}; worst per class: "
          + "; ".join(f"{c}: {n} {e:.2e}" for c, (e, n) in worst.items()))
    assert glob < glob_bound, glob
    for c, (e, n) in worst.items():
        assert e < class_bounds[c], (c, n, e)
    if mutate is None:
        del eng
        return
    Pm, _ = _emulated_backward(spec, state, bn, gelu8, inject=(mutate, 1.01), half=half)
    globm, perm = _grad_errors(eng, Pm)
    wm = max((e, n) for e, n in perm if _param_class(n) == "layernorm")
    print(f"{tag}: with a 1 % error injected into the {mutate} data gradient: global {globm:.2e}, worst LayerNorm parameter {wm[1]} {wm[0]:.2e}")
    assert wm[0] > class_bounds["layernorm"], wm
    if mutate_bound is not None:      # (deep models: the injected error must also break the GLOBAL bound)
        assert globm > mutate_bound, (globm, mutate_bound)
    del eng


@pytest.mark.parametrize("kind,seed", [("roberta", 11), ("bert", 12)])
def test_tiny_forward_backward_vs_oracle(kind, seed):
    spec = _nodrop(VaultSpec.tiny(3, kind))
    B = 3
    bn = synthetic_batch(spec, B, seed=seed, n_classes=3)
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    db = _dev(bn)
    out = eng.forward(db, train=True, labels=db["labels"], need_hidden=True)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    tb = O.torch_batch(bn)
    P = O.to_torch_state(state, requires_grad=True)
    loss, ref = O.vault_loss(P, spec, tb)
    loss.backward()
    with O.emulate_bf16():
        refb = O.vault_forward(O.to_torch_state(state), spec, tb)
    lg = out["logits"].cpu()
    assert (lg - ref["logits"].detach()).abs().max() < 3e-3
    assert (lg - refb["logits"]).abs().max() < 1e-3          # same number format: tighter
    assert abs(float(out["loss"]) - float(loss.detach())) < 2e-3
    hid = out["last_hidden_state"].cpu()
    rh = ref["last_hidden_state"].detach()
    assert (hid - rh).abs().max() < 1e-2 * rh.abs().max()
    assert (out["pooler_output"].cpu() - ref["pooler_output"].detach()).abs().max() < 5e-3
    # golden written by the reference itself
    g = np.load(os.path.join(GOLD, f"tiny_{kind}.npz"))
    assert np.abs(lg.numpy() - g["logits"]).max() < 3e-3
    T = bn["input_ids"].shape[1]
    assert np.abs(hid[:, : T + 1].numpy() - g["hidden_text_cls"]).max() < 1e-2 * np.abs(g["hidden_text_cls"]).max()
    # gradients
    num = den = dot = n1 = 0.0
    for n in eng.params.trainable:
        mine = eng.params.gr(n).cpu().double()
        r = P[n].grad.double()
        num += float((mine - r).pow(2).sum()); den += float(r.pow(2).sum())
        dot += float((mine * r).sum()); n1 += float(mine.pow(2).sum())
    assert (num / den) ** 0.5 < 6e-2
    assert dot / (n1 ** 0.5 * den ** 0.5) > 0.995
    # ... and against the oracle in the HIP backward's own number format (bounds: measured + margin, printed)
    _assert_same_format_gradients(spec, state, bn, f"tiny {kind}", 5e-3, SAME_FORMAT_CLASS_BOUNDS)
    # parameters without gradient in the reference have none here either
    assert not eng.params.has_grad("embeddings.text_embeddings.word_embeddings.weight")
    assert not eng.params.has_grad("embeddings.text_embeddings.position_embeddings.weight")


@pytest.mark.parametrize("name,kind", [("tiny_roberta_ragged", "roberta"), ("tiny_bert_ragged_small", "bert")])
def test_padded_image_batches_vs_oracle_and_reference_golden(name, kind):
    """Batches of differently sized images padded to a non-square canvas (pixel_mask != 1): selected patches only,
    per-image resize of the position table, masked padding rows; the fused sequence of the first case is 233
    tokens long (the 320-key attention instantiation)."""
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    spec = _nodrop(VaultSpec.tiny(3, kind))
    valid_hw = [tuple(int(x) for x in r) for r in g["meta_valid_hw"]]
    pad_hw = tuple(int(x) for x in g["meta_pad_hw"])
    bn = synthetic_ragged_batch(spec, valid_hw, pad_hw, seed=int(g["meta_data_seed"]), n_classes=3)
    B = len(valid_hw)
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    db = _dev(bn)
    out = eng.forward(db, train=True, labels=db["labels"], need_hidden=True)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    tb = O.torch_batch(bn)
    P = O.to_torch_state(state, requires_grad=True)
    loss, ref = O.vault_loss(P, spec, tb)
    loss.backward()
    lg = out["logits"].cpu()
    assert (lg - ref["logits"].detach()).abs().max() < 3e-3
    assert np.abs(lg.numpy() - g["logits"]).max() < 3e-3                      # the reference itself
    assert abs(float(out["loss"]) - float(g["loss"])) < 2e-3
    assert (out["pooler_output"].cpu() - ref["pooler_output"].detach()).abs().max() < 5e-3
    T = bn["input_ids"].shape[1]
    sel, valid, hw, grid, L = select_patches(bn["pixel_mask"], spec.vilt.patch_size)
    hid = out["last_hidden_state"].cpu()
    rh = ref["last_hidden_state"].detach()
    assert hid.shape[1] >= T + 1 + L                                          # the engine may append masked padding rows
    scale = float(rh.abs().max())
    assert np.abs(hid[:, : T + 1].numpy() - g["hidden_text_cls"]).max() < 1e-2 * scale
    for b in range(B):                                                        # real patches, same (row-major) order
        nvb = int(valid[b].sum())
        assert (hid[b, T + 1: T + 1 + nvb] - rh[b, T + 1: T + 1 + nvb]).abs().max() < 1e-2 * scale
    num = den = dot = n1 = 0.0
    for n in eng.params.trainable:
        mine = eng.params.gr(n).cpu().double()
        r = P[n].grad.double()
        num += float((mine - r).pow(2).sum()); den += float(r.pow(2).sum())
        dot += float((mine * r).sum()); n1 += float(mine.pow(2).sum())
    assert (num / den) ** 0.5 < 6e-2
    assert dot / (n1 ** 0.5 * den ** 0.5) > 0.995
    # the position table's gradient flows through the transposed interpolation
    gp = eng.params.gr("embeddings.position_embeddings").cpu().double().reshape(-1)
    rp = P["embeddings.position_embeddings"].grad.double().reshape(-1)
    assert float((gp - rp).norm() / rp.norm()) < 6e-2
    np.testing.assert_allclose(float(gp.norm()), float(g["grad_norms"][list(g["grad_names"]).index(
        "embeddings.position_embeddings")]), rtol=6e-2)


def test_vaultmodel_flags_freeze_lm_and_vilt_position_embeddings_vs_reference_golden():
    """Headless VaultModel with freeze_lm=True and use_vilt_position_embeddings=True, BERT token types 0/1: output
    gradients injected at pooler_output / last_hidden_state (the autograd-bridge entry), compared with the
    reference-generated golden and the oracle."""
    from oracle.make_goldens import flag_case_inputs
    g = np.load(os.path.join(GOLD, "tiny_bert_vaultmodel_flags.npz"))
    spec = VaultSpec.tiny(0, "bert")
    spec.use_vilt_position_embeddings = True
    spec.lm.hidden_dropout_prob = 0.0            # (the golden is an eval-mode run: a frozen LM still drops out in train mode)
    spec.lm.attention_probs_dropout_prob = 0.0
    B = int(g["meta_batch"])
    bn, wp, wh = flag_case_inputs(spec, B, int(g["meta_data_seed"]))
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, freeze_lm=True, half="bf16")
    db = _dev(bn)
    out = eng.forward(db, train=True, need_hidden=True)
    T = bn["input_ids"].shape[1]
    S = out["last_hidden_state"].shape[1]
    dh = torch.zeros(B, S, spec.vilt.hidden_size, device="cuda")
    dh[:, : T + 1] = torch.from_numpy(wh).cuda()
    eng.zero_grad()
    eng.backward(dpooled=torch.from_numpy(wp).cuda(), dhidden=dh)
    torch.cuda.synchronize()
    scale = float(np.abs(g["hidden_text_cls"]).max())
    assert np.abs(out["pooler_output"].cpu().numpy() - g["pooler_output"]).max() < 5e-3
    assert np.abs(out["last_hidden_state"][:, : T + 1].cpu().numpy() - g["hidden_text_cls"]).max() < 1e-2 * scale
    names = [str(n) for n in g["grad_names"]]
    # frozen LM: no gradient storage for it at all; ViLT's text position table is trained under the flag
    assert not any(eng.params.has_grad(n) for n in eng.params.offsets if n.startswith("bert."))
    assert eng.params.has_grad("embeddings.text_embeddings.position_embeddings.weight")
    num = den = 0.0
    for n, ref_norm in zip(names, g["grad_norms"]):
        mine = float(eng.params.gr(n).double().norm())
        num += (mine - ref_norm) ** 2; den += ref_norm ** 2
        # (key-bias gradients are analytically zero: bf16 rounding noise of ~1e-3 absolute is all there is)
        assert abs(mine - ref_norm) <= 8e-2 * ref_norm + 2e-3, (n, mine, ref_norm)
    assert (num / den) ** 0.5 < 3e-2
    for k in g.files:
        if k.startswith("grad::"):
            mine = eng.params.gr(k[6:]).cpu().numpy().reshape(g[k].shape)
            assert np.linalg.norm(mine - g[k]) <= 6e-2 * np.linalg.norm(g[k]) + 1e-5, k


def test_itr_head_model_class_vs_reference_golden():
    """VaultForImageAndTextRetrieval through the nn.Module API (autograd bridge) against the reference's own run."""
    from vault_amd.models.vault import VaultForImageAndTextRetrieval
    g = np.load(os.path.join(GOLD, "tiny_roberta_itr.npz"))
    spec = _nodrop(VaultSpec.tiny(1, "roberta"))
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=1)
    m = VaultForImageAndTextRetrieval(spec.vilt, bert_config=spec.lm).to("cuda").train()
    kw = {k: torch.from_numpy(bn[k]).cuda() for k in ("input_ids", "attention_mask", "pixel_values", "pixel_mask")}
    out = m(**kw)
    assert tuple(out.logits.shape) == (int(g["meta_batch"]), 1) and out.loss is None
    (out.logits * torch.from_numpy(g["w"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert np.abs(out.logits.detach().cpu().numpy() - g["logits"]).max() < 3e-3
    sd = dict(m.named_parameters())
    num = den = 0.0
    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        assert sd[k].grad is not None, k
        mine = float(sd[k].grad.double().norm())
        num += (mine - ref_norm) ** 2; den += ref_norm ** 2
    assert (num / den) ** 0.5 < 3e-2
    for k in g.files:
        if k.startswith("grad::"):
            mine = sd[k[6:]].grad.cpu().numpy().reshape(g[k].shape)
            # B = 3 and a mixed-sign objective: the pooler bias gradient is a sum of three terms that largely cancel,
            # each carrying one bf16 rounding of the tanh-backward operand (2^-9 of ~1e-3): 15 % of the small remainder
            assert np.linalg.norm(mine - g[k]) <= 0.15 * np.linalg.norm(g[k]) + 1e-6, k


def test_vqa_head_model_class_vs_reference_golden():
    """VaultForQuestionAnswering (MLP head through GEMM / LayerNorm / GELU kernels, output projection padded to 256
    columns) through the nn.Module API, loss and gradients against the reference's own run."""
    from vault_amd.models.vault import VaultForQuestionAnswering
    g = np.load(os.path.join(GOLD, "tiny_roberta_vqa.npz"))
    L = g["labels"].shape[1]
    spec = _nodrop(VaultSpec.tiny(L, "roberta"))
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=1)
    m = VaultForQuestionAnswering(spec.vilt, bert_config=spec.lm, n_classes=L).to("cuda").train()
    assert {"classifier.0.weight", "classifier.1.bias", "classifier.3.weight", "vilt.pooler.dense.weight"} <= set(m.state_dict())
    kw = {k: torch.from_numpy(bn[k]).cuda() for k in ("input_ids", "attention_mask", "pixel_values", "pixel_mask")}
    out = m(**kw, labels=torch.from_numpy(g["labels"]).cuda())
    out.loss.backward()
    torch.cuda.synchronize()
    assert tuple(out.logits.shape) == (int(g["meta_batch"]), L)
    assert np.abs(out.logits.detach().cpu().numpy() - g["logits"]).max() < 5e-3
    assert abs(float(out.loss) - float(g["loss"])) < 5e-3
    sd = dict(m.named_parameters())
    num = den = 0.0
    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        assert sd[k].grad is not None, k
        mine = float(sd[k].grad.double().norm())
        num += (mine - ref_norm) ** 2; den += ref_norm ** 2
    assert (num / den) ** 0.5 < 3e-2
    for k in g.files:
        if k.startswith("grad::"):
            mine = sd[k[6:]].grad.cpu().numpy().reshape(g[k].shape)
            assert np.linalg.norm(mine - g[k]) <= 6e-2 * np.linalg.norm(g[k]) + 1e-6, k


def test_nlvr2_head_model_class_vs_reference_golden():
    """VaultForImagesAndTextClassification: two encoder passes (own activation workspaces, modality types 1 / 2), the
    MLP head on the concatenated pooled outputs, CE loss - loss and gradients against the reference's own run."""
    from oracle.make_goldens import nlvr2_pixels
    from vault_amd.models.vault import VaultForImagesAndTextClassification
    g = np.load(os.path.join(GOLD, "tiny_roberta_nlvr2.npz"))
    spec = _nodrop(VaultSpec.tiny(2, "roberta"))
    B, dseed = int(g["meta_batch"]), int(g["meta_data_seed"])
    bn = synthetic_batch(spec, B, seed=dseed, n_classes=2)
    m = VaultForImagesAndTextClassification(spec.vilt, bert_config=spec.lm).to("cuda").train()
    assert tuple(m.state_dict()["vilt.embeddings.token_type_embeddings.weight"].shape) == (3, spec.vilt.hidden_size)
    out = m(input_ids=torch.from_numpy(bn["input_ids"]).cuda(), attention_mask=torch.from_numpy(bn["attention_mask"]).cuda(),
            pixel_values=torch.from_numpy(nlvr2_pixels(spec, B, dseed)).cuda(), labels=torch.from_numpy(bn["labels"]).cuda())
    out.loss.backward()
    torch.cuda.synchronize()
    assert np.abs(out.logits.detach().cpu().numpy() - g["logits"]).max() < 5e-3
    assert abs(float(out.loss.detach()) - float(g["loss"])) < 3e-3
    sd = dict(m.named_parameters())
    num = den = 0.0
    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        assert sd[k].grad is not None, k
        mine = float(sd[k].grad.double().norm())
        num += (mine - ref_norm) ** 2; den += ref_norm ** 2
    assert (num / den) ** 0.5 < 3e-2
    for k in g.files:
        if k.startswith("grad::"):
            mine = sd[k[6:]].grad.cpu().numpy().reshape(g[k].shape)
            assert np.linalg.norm(mine - g[k]) <= 8e-2 * np.linalg.norm(g[k]) + 1e-6, k
    with pytest.raises(ValueError):
        m(input_ids=torch.from_numpy(bn["input_ids"]).cuda(), pixel_values=torch.zeros(B, 3, 3, 192, 192, device="cuda"))


def test_mlm_head_model_class_vs_reference_golden():
    """VaultForMaskedLM: ViltMLMHead on the text rows through the HIP engine (decoder = ViLT's word embeddings read as a
    256-padded GEMM operand; their gradient comes from the head alone), CE with ignore_index, against the reference."""
    from vault_amd.models.vault import VaultForMaskedLM
    g = np.load(os.path.join(GOLD, "tiny_roberta_mlm.npz"))
    spec = _nodrop(VaultSpec.tiny(0, "roberta"))
    B = int(g["meta_batch"])
    bn = synthetic_batch(spec, B, seed=int(g["meta_data_seed"]), n_classes=1)
    m = VaultForMaskedLM(spec.vilt, bert_config=spec.lm).to("cuda").train()
    assert {"mlm_score.decoder.bias", "mlm_score.transform.dense.weight", "vilt.embeddings.cls_token"} <= set(m.state_dict())
    kw = {k: torch.from_numpy(bn[k]).cuda() for k in ("input_ids", "attention_mask", "pixel_values", "pixel_mask")}
    out = m(**kw, labels=torch.from_numpy(g["labels"]).cuda())
    out.loss.backward()
    torch.cuda.synchronize()
    T, V = bn["input_ids"].shape[1], spec.vilt.vocab_size
    assert tuple(out.logits.shape) == (B, T, V)
    lg = out.logits.detach().cpu().numpy()
    assert np.abs(lg[:, :4] - g["logits_slice"]).max() < 1e-2 * np.abs(g["logits_slice"]).max()
    assert abs(float(out.loss.detach()) - float(g["loss"])) < 5e-3
    sd = dict(m.named_parameters())
    internal = {"mlm_score.decoder.bias": "mlm_score.decoder.bias"}
    num = den = 0.0
    for k, ref_norm in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        assert sd[k].grad is not None, k
        mine = float(sd[k].grad.double().norm())
        num += (mine - ref_norm) ** 2; den += ref_norm ** 2
    assert (num / den) ** 0.5 < 3e-2
    for k in g.files:
        if k.startswith("grad::"):
            mine = sd[k[6:]].grad.cpu().numpy().reshape(g[k].shape)
            assert np.linalg.norm(mine - g[k]) <= 8e-2 * np.linalg.norm(g[k]) + 1e-6, k


def test_gradients_accumulate_across_backward_passes():
    """Two forward/backward passes without zeroing in between leave the SUM of the two gradients in every
    parameter (gradient accumulation; also what multi-image heads rely on)."""
    spec = _nodrop(VaultSpec.tiny(3, "bert"))
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    b1, b2 = _dev(synthetic_batch(spec, 3, seed=21, n_classes=3)), _dev(synthetic_batch(spec, 3, seed=22, n_classes=3))
    grads = []
    for b in (b1, b2):
        eng.forward(b, train=True, labels=b["labels"], need_hidden=False)
        eng.zero_grad()
        eng.backward()
        grads.append(eng.params.g[: eng.params.n_train].clone())
    eng.zero_grad()
    for b in (b1, b2):
        eng.forward(b, train=True, labels=b["labels"], need_hidden=False)
        eng.backward()
    torch.cuda.synchronize()
    both = eng.params.g[: eng.params.n_train]
    want = grads[0] + grads[1]
    assert float((both - want).norm() / want.norm()) < 1e-5       # float-atomic summation order only


@pytest.mark.gpu
@pytest.mark.parametrize("group", [0, 1])
def test_batched_weight_gradients_equal_the_per_layer_ones(group):
    """Deferred, batched weight gradients (vault_gemm batch: all layers of a stack per launch, or - `group` = 1 - one
    launch per layer at every group boundary, the data-parallel form) leave the gradients of the per-layer launches."""
    spec = _nodrop(VaultSpec.tiny(3, "roberta"))
    state = build_state(spec, 0)
    b = _dev(synthetic_batch(spec, 5, seed=31, n_classes=3))
    grads, tags = [], []
    for batched in (False, True):
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
        eng.LM_WGRAD_BATCHED, eng.LM_WGRAD_GROUP = batched, group
        eng.dp_world = 2      # (the LM stack's groups exist for steps with more than one rank; one rank takes whole stacks)
        eng.forward(b, train=True, labels=b["labels"], need_hidden=False)
        eng.zero_grad()
        seen = []
        eng.backward(after_layer=seen.append)
        torch.cuda.synchronize()
        assert ("lm_act_all" in eng.last) == batched and ("act_all" in eng.last) == batched
        grads.append(eng.params.g[: eng.params.n_train].clone())
        tags.append(seen)
    # every stage is reported exactly once, in descending address order - except that with deferred weight gradients and a
    # stage listener (data-parallel steps) the embedding backward runs ahead of the last group's launches: "lm_embed" comes
    # before that group's layers (its gradient range is all-reduced under their weight-gradient GEMMs)
    assert sorted(tags[0]) == sorted(tags[1]) and len(set(tags[1])) == len(tags[1])
    nl = spec.lm.num_hidden_layers
    last = min(nl, group) if group > 0 else nl                      # layers of the last LM group
    want = [t for t in tags[0] if t != "lm_embed"]
    want.insert(len(want) - last, "lm_embed")
    assert tags[1] == want, (tags[1], want)
    assert float((grads[0] - grads[1]).norm() / grads[0].norm()) < 1e-5   # float-atomic summation order only


def test_full_size_against_reference_golden():
    """12+12 layers, hidden 768, B=2 (one padded caption): compare with numbers produced by the
    reference (HuggingFace ViltModel + RobertaModel under ref VaultForTMSC) in the build container."""
    g = np.load(os.path.join(GOLD, "full_bertweet_b2.npz"))
    spec = _nodrop(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3))
    B = int(g["meta_batch"])
    bn = synthetic_batch(spec, B, seed=int(g["meta_data_seed"]), n_classes=3)
    eng = VaultEngine(spec, "cuda:0", seed=0, classifier_dropout=0.0, half="bf16")
    db = _dev(bn)
    out = eng.forward(db, train=True, labels=db["labels"], need_hidden=True)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    # bf16 fast mode, measured 4.3e-3 / 1.2-1.7e-3 / 1.1e-2 (the level moves with the kernels' summation
    # order; the north star's 1e-3 is met by the precise mode, test_precise_mode_meets_the_1e3_logits_bar):
    # bounds = largest measured + 25 %
    dl_, dloss_ = np.abs(out["logits"].cpu().numpy() - g["logits"]).max(), abs(float(out["loss"]) - float(g["loss"]))
    print(f"full_bertweet_b2: |dlogits| {dl_:.2e} |dloss| {dloss_:.2e}")
    assert dl_ < 5.5e-3
    assert dloss_ < 2.1e-3
    assert np.abs(out["pooler_output"].cpu().numpy() - g["pooler_output"]).max() < 1.4e-2
    T = bn["input_ids"].shape[1]
    h = out["last_hidden_state"][:, : T + 1].cpu().numpy()
    assert np.abs(h - g["hidden_text_cls"]).max() < 1.5e-2 * np.abs(g["hidden_text_cls"]).max()
    pn = np.sort(np.linalg.norm(out["last_hidden_state"][:, T + 1:].cpu().numpy(), axis=-1), axis=1)
    np.testing.assert_allclose(pn, g["hidden_patch_sorted_norms"], rtol=5e-3)
    # gradient norms per parameter (the analytically-zero key biases excluded)
    names = [str(n) for n in g["grad_names"]]
    bad = []
    for n, rn in zip(names, g["grad_norms"]):
        if ".key.bias" in n:
            continue
        mine = float(eng.params.gr(n).double().norm())
        if abs(mine - rn) > 0.08 * rn + 1e-7:
            bad.append((n, mine, rn))
    assert not bad, bad[:5]
    for k in g.files:
        if k.startswith("grad::") and ".key.bias" not in k:
            mine = eng.params.gr(k[6:]).cpu().numpy().reshape(g[k].shape)
            rel = np.linalg.norm(mine - g[k]) / (np.linalg.norm(g[k]) + 1e-12)
            assert rel < 8e-2, (k, rel)
    # the 8 % / 8e-2 bounds above are the fp32 comparison (bf16 operands against the reference's fp32 arithmetic); the
    # backward itself is pinned against the oracle run in the same number format
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    del eng
    # (24 layers deep the two runs no longer round the same values: one flipped bf16 rounding perturbs everything behind it
    #  by 2^-9 relative, which flips roundings wholesale a layer later - the same-format comparison decays towards the
    #  fp32 one with depth (measured 7.1e-3 here against 1.0e-2); the tight bounds and the mutation check are on the
    #  shallow full-width models of test_full_width_shallow_same_format_gradients)
    _assert_same_format_gradients(spec, build_state(spec, 0), bn, "full size B=2", 1e-2, SAME_FORMAT_DEEP_BOUNDS, mutate=None)


def test_single_image_and_caption_through_vaultmodel_full_size():
    """BASELINE configs[0]: one image + one caption through the headless VaultModel (ViLT-B32 + bertweet-base shapes)
    in eval mode; sample 0 of the full-size reference golden, run alone (B = 1: every buffer padded to one tile)."""
    from vault_amd.models.vault import VaultModel
    g = np.load(os.path.join(GOLD, "full_bertweet_b2.npz"))
    spec = _nodrop(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3))
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=3)
    kw = {k: torch.from_numpy(v[:1]).cuda() for k, v in bn.items() if k != "labels"}
    enc = VaultModel(spec.vilt, bert_config=spec.lm, vilt_dropout_prob=0.0).to("cuda").eval()
    with torch.no_grad():
        o = enc(**kw)
    T = bn["input_ids"].shape[1]
    assert o.last_hidden_state.shape == (1, T + 1 + 144, 768) and o.pooler_output.shape == (1, 768)
    assert np.abs(o.pooler_output.cpu().numpy() - g["pooler_output"][:1]).max() < 2e-2
    h = o.last_hidden_state[:, : T + 1].cpu().numpy()
    assert np.abs(h - g["hidden_text_cls"][:1]).max() < 1.5e-2 * np.abs(g["hidden_text_cls"]).max()
    pn = np.sort(np.linalg.norm(o.last_hidden_state[:, T + 1:].cpu().numpy(), axis=-1), axis=1)
    np.testing.assert_allclose(pn, g["hidden_patch_sorted_norms"][:1], rtol=5e-3)


def test_output_hidden_states_against_the_oracle_taps():
    """``VaultModel(..., output_hidden_states=True)``: HF's tuple of num_hidden_layers + 1 tensors - the embedding output and
    every ViLT layer's output (the residual stream before the final LayerNorm) - against the oracle's taps of the same
    quantities, in eval mode (the engine then keeps one buffer per layer instead of ping-ponging) and in train mode;
    ``output_attentions`` still raises."""
    from vault_amd.models.vault import VaultModel
    spec = _nodrop(VaultSpec.tiny(0, "roberta"))
    bn = synthetic_batch(spec, 3, seed=8)
    state = build_state(spec, 0)
    m = VaultModel(spec.vilt, bert_config=spec.lm, vilt_dropout_prob=0.0)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    m = m.to("cuda")
    kw = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    taps = {}
    ref = O.vault_forward(O.to_torch_state(state), spec, O.torch_batch(bn), taps=taps)
    want = [taps["vilt_embed"]] + [taps[f"vilt_layer{i}"] for i in range(spec.vilt.num_hidden_layers)]
    for mode in ("eval", "train"):
        (m.eval() if mode == "eval" else m.train())
        import contextlib
        with (torch.no_grad() if mode == "eval" else contextlib.nullcontext()):
            o = m(**kw, output_hidden_states=True)
            plain = m(**kw)
        assert plain.hidden_states is None and len(o.hidden_states) == spec.vilt.num_hidden_layers + 1
        assert torch.equal(o.last_hidden_state, plain.last_hidden_state)
        for got, w in zip(o.hidden_states, want):
            assert got.shape == w.shape and not got.requires_grad
            assert (got.cpu() - w).abs().max() < 1e-2 * w.abs().max()
    with pytest.raises(NotImplementedError):
        m(**kw, output_attentions=True)


def test_eval_determinism_and_no_lm():
    spec = VaultSpec.tiny(3, "roberta")
    spec_nolm = VaultSpec(vilt=spec.vilt, lm=None, n_classes=0)
    bn = synthetic_batch(spec_nolm, 2, seed=3)
    state = build_state(spec_nolm, 0)
    eng = VaultEngine(spec_nolm, "cuda:0", state=state, with_grads=False, half="bf16")
    db = _dev(bn)
    a = eng.forward(db, train=False)
    h1 = a["last_hidden_state"].clone(); p1 = a["pooler_output"].clone()
    b = eng.forward(db, train=False)
    torch.cuda.synchronize()
    assert torch.equal(h1, b["last_hidden_state"]) and torch.equal(p1, b["pooler_output"])
    ref = O.vault_forward(O.to_torch_state(state), spec_nolm, O.torch_batch(bn))
    assert (h1.cpu() - ref["last_hidden_state"]).abs().max() < 1e-2 * ref["last_hidden_state"].abs().max()
    assert (p1.cpu() - ref["pooler_output"]).abs().max() < 5e-3


def test_unsupported_inputs_raise():
    spec = VaultSpec.tiny(3, "roberta")
    eng = VaultEngine(spec, "cuda:0", with_grads=False, half="bf16")
    bn = synthetic_batch(spec, 2, seed=3)
    db2 = _dev(bn)
    db2["pixel_values"] = db2["pixel_values"][:, :, :90, :96]          # not a multiple of the patch size
    with pytest.raises(ValueError):
        eng.forward(db2)
    db3 = _dev(bn)
    db3["pixel_values"] = torch.zeros(2, 3, 320, 320, device="cuda")   # 400 patches: beyond the 320-key attention
    db3["pixel_mask"] = torch.ones(2, 320, 320, dtype=torch.int64, device="cuda")
    with pytest.raises(ValueError):
        eng.forward(db3)
    db4 = _dev(bn)
    db4["pixel_values"] = db4["pixel_values"][:1]                      # batch mismatch (HF error)
    with pytest.raises(ValueError):
        eng.forward(db4)


def test_model_api_autograd_bridge():
    from vault_amd.models.vault import VaultForTMSC, VaultModel
    spec = _nodrop(VaultSpec.tiny(3, "roberta"))
    bn = synthetic_batch(spec, 3, seed=11, n_classes=3)
    with pytest.raises(RuntimeError):
        VaultForTMSC(spec.vilt, n_classes=3, bert_config=spec.lm)(**{k: torch.from_numpy(v) for k, v in bn.items()
                                                                     if k != "labels"})
    model = VaultForTMSC(spec.vilt, n_classes=3, vilt_dropout_prob=0.0, bert_config=spec.lm).to("cuda")
    kw = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    model.eval()
    with torch.no_grad():
        lg_eval = model(**kw)
    assert lg_eval.shape == (3, 3)
    model.train()
    logits = model(**kw)
    loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(bn["labels"]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    state = build_state(spec, 0)
    P = O.to_torch_state(state, requires_grad=True)
    rl, ref = O.vault_loss(P, spec, O.torch_batch(bn))
    rl.backward()
    assert (logits.detach().cpu() - ref["logits"].detach()).abs().max() < 3e-3
    assert abs(float(loss) - float(rl.detach())) < 2e-3
    sd = dict(model.named_parameters())
    gw = sd["pooler.dense.weight"].grad
    assert gw is not None
    r = P["pooler.dense.weight"].grad
    assert float((gw.cpu() - r).norm() / r.norm()) < 5e-2
    assert sd["embeddings.text_embeddings.word_embeddings.weight"].grad is None
    # zero_grad(set_to_none) + second backward gives fresh (not doubled) gradients
    model.zero_grad(set_to_none=True)
    logits = model(**kw)
    torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(bn["labels"]).cuda()).backward()
    gw2 = dict(model.named_parameters())["pooler.dense.weight"].grad
    assert float((gw2.cpu() - r).norm() / r.norm()) < 5e-2
    # HF input validation errors
    with pytest.raises(ValueError):
        model(input_ids=kw["input_ids"], attention_mask=kw["attention_mask"])
    with pytest.raises(ValueError):
        model(input_ids=kw["input_ids"][:2], pixel_values=kw["pixel_values"])
    # VaultModel returns the HF output object
    enc = VaultModel(spec.vilt, bert_config=spec.lm).to("cuda").eval()
    with torch.no_grad():
        o = enc(**kw)
    assert o.last_hidden_state.shape == (3, 185, 256) and o.pooler_output.shape == (3, 256)
    assert "pooler_output" in o.keys()
    assert (o.pooler_output.cpu() - ref["pooler_output"].detach()).abs().max() < 5e-3
    # state_dict round trip keeps outputs
    sd2 = {k: v.clone() for k, v in model.state_dict().items()}
    m2 = VaultForTMSC(spec.vilt, n_classes=3, vilt_dropout_prob=0.0, bert_config=spec.lm, _seed=1).to("cuda").eval()
    m2.load_state_dict(sd2)
    with torch.no_grad():
        assert (m2(**kw) - lg_eval).abs().max() < 1e-6


def test_precise_mode_meets_the_1e3_logits_bar():
    """Split-bf16 ("bf16x3") inference: every Linear runs as A_hi W_hi + A_lo W_hi + A_hi W_lo on the bf16
    MFMA kernels.  This is the mode that meets the north-star tolerance (logits within 1e-3 of the fp32
    reference) at full depth; the plain bf16 mode sits at ~4e-3 there (see test above)."""
    g = np.load(os.path.join(GOLD, "full_bertweet_b2.npz"))
    spec = _nodrop(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3))
    B = int(g["meta_batch"])
    bn = synthetic_batch(spec, B, seed=int(g["meta_data_seed"]), n_classes=3)
    eng = VaultEngine(spec, "cuda:0", seed=0, classifier_dropout=0.0, with_grads=False, half="bf16")
    db = _dev(bn)
    out = eng.forward(db, train=False, labels=db["labels"], need_hidden=True, precise=True)
    torch.cuda.synchronize()
    dl = np.abs(out["logits"].cpu().numpy() - g["logits"]).max()
    assert dl < 1e-3, dl
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-3
    assert np.abs(out["pooler_output"].cpu().numpy() - g["pooler_output"]).max() < 2.5e-3
    T = bn["input_ids"].shape[1]
    h = out["last_hidden_state"][:, : T + 1].cpu().numpy()
    assert np.abs(h - g["hidden_text_cls"]).max() < 4e-3 * np.abs(g["hidden_text_cls"]).max()
    # and the fast mode on the same engine still works afterwards (separate buffers)
    out2 = eng.forward(db, train=False, need_hidden=False)
    assert np.abs(out2["logits"].cpu().numpy() - g["logits"]).max() < 8e-3


def test_precise_forward_training_step_meets_the_1e3_bar_with_bf16_level_gradients():
    """A TRAINING step whose logits / loss are inside the north star's 1e-3: split-bf16 forward GEMMs (``precise=True`` with
    ``train=True``), bf16 backward on the plain bf16 operands kept beside the split ones.  Full size against the reference
    golden: logits / loss < 1e-3 in train mode, gradients at the fast mode's bf16 level; then ``TrainStep(precise_forward=True)``
    for three steps against the fp32 oracle's trajectory (tape replay included: the split weight shadow is re-derived from the
    weights the optimizer wrote in every step)."""
    from vault_amd.train import TrainStep
    # three optimisation steps on the tiny model, against the fp32 oracle stepping with the HF-AdamW formula
    spec = _nodrop(VaultSpec.tiny(3, "roberta"))
    bn = synthetic_batch(spec, 4, seed=31, n_classes=3)
    state = build_state(spec, 0)
    tb = O.torch_batch(bn)
    P = O.to_torch_state(state, requires_grad=True)
    m = {k: torch.zeros_like(v) for k, v in P.items()}; v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    ref = []
    for t in range(1, 4):
        for p_ in P.values():
            p_.grad = None
        loss, _ = O.vault_loss(P, spec, tb)
        loss.backward()
        ref.append(float(loss.detach()))
        with torch.no_grad():
            for k, p_ in P.items():
                if p_.grad is not None:
                    O.hf_adamw_step(p_, p_.grad, m[k], v2[k], 5e-5, t)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    for use_tape in (False, True):
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
        step = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10, constant_lr=True, use_tape=use_tape,
                         precise_forward=True)
        losses = [float(step(db, labels)) for _ in range(3)]
        assert abs(losses[0] - ref[0]) < 2e-4, (losses, ref)            # the first loss is a pure forward quantity: fp32 class
        assert max(abs(a - b) for a, b in zip(losses, ref)) < 2e-3, (losses, ref)
        del eng, step
    # ---- full size, reference golden
    g = np.load(os.path.join(GOLD, "full_bertweet_b2.npz"))
    spec = _nodrop(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3))
    B = int(g["meta_batch"])
    bn = synthetic_batch(spec, B, seed=int(g["meta_data_seed"]), n_classes=3)
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    db = _dev(bn)
    out = eng.forward(db, train=True, labels=db["labels"], need_hidden=False, precise=True)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    dl = np.abs(out["logits"].cpu().numpy() - g["logits"]).max()
    dloss = abs(float(out["loss"]) - float(g["loss"]))
    print(f"precise-forward training step: |dlogits| {dl:.2e} |dloss| {dloss:.2e}")
    assert dl < 1e-3 and dloss < 1e-3
    names = [str(n) for n in g["grad_names"]]
    bad = []
    for n, rn in zip(names, g["grad_norms"]):
        if ".key.bias" in n:
            continue
        mine = float(eng.params.gr(n).double().norm())
        if abs(mine - rn) > 0.08 * rn + 1e-7:
            bad.append((n, mine, rn))
    assert not bad, bad[:5]
    for k in g.files:
        if k.startswith("grad::") and ".key.bias" not in k:
            mine = eng.params.gr(k[6:]).cpu().numpy().reshape(g[k].shape)
            rel = np.linalg.norm(mine - g[k]) / (np.linalg.norm(g[k]) + 1e-12)
            assert rel < 8e-2, (k, rel)


def test_precise_step_after_a_fast_step_on_the_same_workspace():
    """The gelu' format is a property of the FORWARD that wrote it: a fast-mode train step leaves the 8-bit
    tile image plan in the workspace, a precise-forward step on the same (B, T) workspace writes plain 16-bit gelu' - its
    backward must read what its own forward stored.  Full width, B = 48 (8,880 token rows: the per-kernel path with the 8-bit
    gelu'), 2 + 2 layers: gradients of (fast step, then precise step) on one engine equal those of a fresh engine's precise step."""
    spec = _nodrop(VaultSpec(vilt=ViltSpec(num_hidden_layers=2), lm=LMSpec.bertweet_base(), n_classes=3))
    spec.lm.num_hidden_layers = 2
    state = build_state(spec, 3)
    bn = synthetic_batch(spec, 48, seed=548, n_classes=3)
    db = _dev(bn)

    def precise_step(eng):
        out = eng.forward(db, train=True, labels=db["labels"], need_hidden=False, precise=True)
        eng.zero_grad()
        eng.backward()
        torch.cuda.synchronize()
        return out["logits"].clone(), eng.params.g[:eng.params.n_train].clone()

    a = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    a.forward(db, train=True, labels=db["labels"], need_hidden=False)
    a.zero_grad()
    a.backward()
    assert a.last.get("gelu8_active") in (5, 6)          # the fast step used the 8-bit image ...
    la, ga = precise_step(a)
    assert a.last.get("gelu8_active") is None            # ... the precise step did not
    b = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    lb, gb = precise_step(b)
    assert torch.equal(la, lb)
    rel = float((ga - gb).norm() / gb.norm())
    print(f"precise step after a fast step vs a fresh precise step: gradient rel diff {rel:.2e}")
    assert rel < 1e-5                                    # (float-atomic summation order only)


@pytest.mark.parametrize("kind,seed", [("roberta", 11), ("bert", 12)])
def test_precise_mode_tiny(kind, seed):
    spec = _nodrop(VaultSpec.tiny(3, kind))
    bn = synthetic_batch(spec, 3, seed=seed, n_classes=3)
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, with_grads=False, half="bf16")
    out = eng.forward(_dev(bn), train=False, need_hidden=True, precise=True)
    ref = O.vault_forward(O.to_torch_state(state), spec, O.torch_batch(bn))
    torch.cuda.synchronize()
    assert (out["logits"].cpu() - ref["logits"]).abs().max() < 3e-4
    rh = ref["last_hidden_state"]
    assert (out["last_hidden_state"].cpu() - rh).abs().max() < 2e-3 * rh.abs().max()


def test_frozen_bert_base_full_size_against_reference_golden():
    """BASELINE config 4 at its real shapes: ViLT-B32 + bert-base-uncased (vocab 30522, 512 positions, 2 token types,
    eps 1e-12) with the LM frozen like ref from_pretrained(freeze_lm=True) (model.py:124-126,189), B = 2, against numbers
    produced by the reference itself (oracle/make_goldens.py: full_bert_base_frozen_b2)."""
    g = np.load(os.path.join(GOLD, "full_bert_base_frozen_b2.npz"))
    spec = _nodrop(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bert_base_uncased(), n_classes=3))
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=3)
    eng = VaultEngine(spec, "cuda:0", seed=0, classifier_dropout=0.0, freeze_lm=True, half="bf16")
    assert not any(n.startswith("bert.") for n in eng.params.trainable)
    db = _dev(bn)
    out = eng.forward(db, train=True, labels=db["labels"], need_hidden=True)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    # measured 3.9e-3 / 1.0e-3 / 9e-3 (bf16 operands over 24 layers): bounds = measured + 25 %
    dl_, dloss_ = np.abs(out["logits"].cpu().numpy() - g["logits"]).max(), abs(float(out["loss"]) - float(g["loss"]))
    print(f"full_bert_base_frozen_b2: |dlogits| {dl_:.2e} |dloss| {dloss_:.2e}")
    assert dl_ < 5.5e-3
    assert dloss_ < 2.1e-3
    assert np.abs(out["pooler_output"].cpu().numpy() - g["pooler_output"]).max() < 1.5e-2
    names = [str(n) for n in g["grad_names"]]
    assert names and not any(n.startswith("bert.") for n in names)
    bad = []
    for n, rn in zip(names, g["grad_norms"]):
        if ".key.bias" in n:
            continue
        mine = float(eng.params.gr(n).double().norm())
        if abs(mine - rn) > 0.08 * rn + 1e-7:
            bad.append((n, mine, rn))
    assert not bad, bad[:5]
    for k in g.files:
        if k.startswith("grad::") and ".key.bias" not in k and not k.startswith("grad::bert."):
            mine = eng.params.gr(k[6:]).cpu().numpy().reshape(g[k].shape)
            rel = np.linalg.norm(mine - g[k]) / (np.linalg.norm(g[k]) + 1e-12)
            assert rel < 8e-2, (k, rel)


def test_full_size_batch_48_forward_backward_vs_oracle():
    """The code paths of the B = 256 bench inside a real step, against the fp32 CPU oracle computed here: B = 48 at full
    size gives 8,880 fused tokens = 35 row tiles x 9-12 column tiles (more GEMM tiles than the 256 persistent blocks:
    several tiles per block in the 8-wave and ring kernels), 576 (batch, head) items in the resident attention
    backward (> 256 workgroups), batched weight gradients in groups of 6 layers with the cost-model split count,
    LayerNorm / column sums over thousands of rows.  Compared: logits, loss, pooled output, every parameter's
    gradient norm, cosine and relative error of the large gradients."""
    spec = _nodrop(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3))
    B = 48
    bn = synthetic_batch(spec, B, seed=77, n_classes=3)
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    eng.HEAD_MAJOR_MIN_ROWS = 0                                        # (the bench shape's head-major qkv / dqkv at this row count too)
    db = _dev(bn)
    out = eng.forward(db, train=True, labels=db["labels"], need_hidden=False)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    assert "act_all" in eng.last and "lm_act_all" in eng.last          # the deferred, batched weight gradients ran
    assert eng.last.get("gelu8_cfg") in (5, 6)                         # ... with the 8-bit tile-native gelu' in the ViLT FFN
    assert eng.GRAD_STREAM_BF16                                        # ... and the bf16 residual-gradient stream
    assert eng.last.get("qkv_hm") == eng.last["Mp"]                    # ... and head-major qkv / dqkv in the ViLT stack
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    P = O.to_torch_state(state, requires_grad=True)
    loss, ref = O.vault_loss(P, spec, O.torch_batch(bn))
    loss.backward()
    dl = (out["logits"].cpu() - ref["logits"].detach()).abs().max().item()
    dloss = abs(float(out["loss"]) - float(loss.detach()))
    print(f"full size B=48: |dlogits| {dl:.2e} |dloss| {dloss:.2e}")
    assert dl < 7e-3, dl                                               # (max over 48 samples; B = 2 golden: 4.3e-3)
    assert dloss < 1.5e-3
    assert (out["pooler_output"].cpu() - ref["pooler_output"].detach()).abs().max().item() < 1.5e-2
    bad, tot_err, tot_ref = [], 0.0, 0.0
    for n in eng.params.trainable:
        gr = P[n].grad
        if gr is None or ".key.bias" in n:
            continue
        mine = eng.params.gr(n).cpu().double().reshape(gr.shape)
        rn = float(gr.double().norm())
        err = float((mine - gr.double()).norm())
        tot_err += err ** 2; tot_ref += rn ** 2
        if abs(float(mine.norm()) - rn) > 0.08 * rn + 1e-7:
            bad.append((n, float(mine.norm()), rn))
        if gr.numel() >= 768 * 768:
            cos = float((mine * gr.double()).sum() / (mine.norm() * gr.double().norm() + 1e-30))
            assert cos > 0.995 and err < 6e-2 * rn, (n, cos, err / rn)
    assert not bad, bad[:5]
    print(f"full size B=48: gradient global relative L2 error {(tot_err / tot_ref) ** 0.5:.3e}")
    assert (tot_err / tot_ref) ** 0.5 < 2.5e-2                         # global relative L2 (1.0 % at B = 2)
    # same number format (8-bit gelu' grid in the ViLT FFN, bf16 gradient stream, bf16 dY / saved operands): the tight bound
    del eng
    # (at B = 48 the per-sample rounding noise averages out: measured 1.5e-3 globally / 1.8e-3, 6.2e-3, 1.0e-2 per class - the
    #  bounds are measured + 50 %, and they notice a 1 % error injected into ONE data-gradient GEMM of layer 9: the upper
    #  weight-gradient group (layers 6-11), the 8-bit gelu' path and the 16-bit gradient stream are what carries it down)
    _assert_same_format_gradients(spec, state, bn, "full size B=48", 2.3e-3, SAME_FORMAT_B48_BOUNDS, gelu8=True,
                                  mutate="encoder.layer.9.attention.attention.qkv")


@pytest.mark.parametrize("B", [2, 48])
def test_full_width_shallow_same_format_gradients(B):
    """The backward pinned to its own number format at FULL WIDTH (hidden 768, FFN 3072, 185-token fused sequence, 12 heads)
    and 2 + 2 layers: B = 2 runs the stage-level layer calls with 128 x 128 tiles and bf16 gelu', B = 48 the kernels of the
    B = 256 bench (8-wave / ring GEMMs, 8-bit gelu', batched weight gradients, resident attention backward, bf16 gradient
    stream).  HIP gradients against the oracle emulating the HIP number format forward and backward: <= 5e-3 global relative L2,
    per-parameter class bounds, and the bounds notice a 1 % error in one data-gradient GEMM."""
    spec = _nodrop(VaultSpec(vilt=ViltSpec(num_hidden_layers=2), lm=LMSpec.bertweet_base(), n_classes=3))
    spec.lm.num_hidden_layers = 2
    state = build_state(spec, 3)
    bn = synthetic_batch(spec, B, seed=500 + B, n_classes=3)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    _assert_same_format_gradients(spec, state, bn, f"full width 2+2 layers B={B}", 5e-3, SAME_FORMAT_CLASS_BOUNDS)


@pytest.mark.parametrize("half", ["bf16", "fp16"])
def test_full_size_batch_256_equals_its_sub_batches(half):
    """BASELINE's headline shape itself (full size, per-GPU batch 256: 47,360 fused tokens = 185 row tiles, several rounds of
    every persistent GEMM, un-split batched weight gradients, 3,072 attention items, the 8-bit gelu', the bf16 gradient stream)
    through a size-independent property: a batch is its samples - the eval logits of every sample and the training loss equal
    those of the same samples run in 8 sub-batches of 32 (oracle-pinned code paths of a smaller shape, other kernels and tiles),
    and the gradient of the mean loss equals the mean of the sub-batch gradients.  On both operand formats: bf16 (what bench.py
    times as `value`) and fp16 (the API default: head-major qkv x whole-stack grouped weight gradients x the scaled 16-bit
    gradient stream at the batch the bench runs it at), the latter with the logits bound at 5e-4."""
    spec = _nodrop(VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3))
    B, SB = 256, 32
    bn = synthetic_batch(spec, B, seed=2024, n_classes=3)
    state = build_state(spec, 0)
    big = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half=half)
    db = _dev(bn)
    ev_big = big.forward(db, train=False)["logits"].clone()
    out = big.forward(db, train=True, labels=db["labels"], need_hidden=False)
    big.zero_grad(); big.backward()
    torch.cuda.synchronize()
    assert big.last.get("gelu8_cfg") in (5, 6) and "act_all" in big.last
    assert big.last.get("qkv_hm") == big.last["Mp"]                    # head-major qkv / dqkv in the ViLT stack (47,360 rows)
    loss_big = float(out["loss"]); tr_big = out["logits"].clone()
    g_big = big.params.g[: big.params.n_train].clone()
    names = list(big.params.trainable)
    views = {n: big.params.gr(n).clone() for n in names if big.params.gr(n).numel() >= 768 * 768}
    del big, out
    torch.cuda.empty_cache()
    small = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half=half)
    small.zero_grad()
    ev, tr, losses = [], [], []
    for k in range(0, B, SB):
        sb = {n: v[k:k + SB].contiguous() for n, v in db.items()}
        ev.append(small.forward(sb, train=False)["logits"].clone())
        o = small.forward(sb, train=True, labels=sb["labels"], need_hidden=False)
        small.backward(grad_scale=1.0 / B)                  # accumulates: sum over sub-batches of sum_i dloss_i / B
        tr.append(o["logits"].clone()); losses.append(float(o["loss"]))
    torch.cuda.synchronize()
    ev, tr = torch.cat(ev), torch.cat(tr)
    # same samples, same weights, different kernels / tile shapes / summation orders: agreement at the operand format's level
    # (round 6: the 32-sample sub-batches split the contraction of their N = 768 Linears in two - engine.SPLITK - and so add the
    #  same products in another order than the 256-sample batch: bf16 logits moved from 2.3e-3 to 2.6e-3 apart, fp16 stayed)
    lb, lossb, gb_, cosb = (3e-3, 2e-4, 1.5e-2, 0.9995) if half == "bf16" else (5e-4, 5e-5, 3e-3, 0.99995)
    assert float((ev_big - ev).abs().max()) < lb and float((tr_big - tr).abs().max()) < lb
    assert abs(loss_big - sum(losses) / len(losses)) < lossb
    g_small = small.params.g[: small.params.n_train]
    assert bool(torch.isfinite(g_big).all()) and bool(torch.isfinite(g_small).all())
    rel = float((g_big - g_small).norm() / g_small.norm())
    print(f"{half}: B=256 against 8 x 32: |dlogits| {float((tr_big - tr).abs().max()):.2e}, loss diff "
          f"{abs(loss_big - sum(losses) / len(losses)):.2e}, gradient relative L2 {rel:.2e}")
    assert rel < gb_
    worst = 1.0
    for n, gb in views.items():
        gs = small.params.gr(n)
        cos = float((gb * gs).sum() / (gb.norm() * gs.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > cosb, (n, cos)
    print(f"{half}: lowest cosine over the {len(views)} large gradients {worst:.6f}")


@pytest.mark.parametrize("half", ["bf16", "fp16"])
def test_head_major_layout_changes_addresses_not_results(half):
    """qkv / dqkv head-major ([3][heads][rows][64]: engine._plan_head_major) against the row-major layout on one model and batch:
    full width, 2 + 2 layers, B = 208 (38,480 fused / 8,320 text rows: both stacks on the per-kernel path), with LM dropout on -
    logits, loss and every activation-side quantity bit-identical (the same arithmetic on other addresses), parameter gradients
    equal up to the summation order of float atomics and of the bias column sums."""
    spec = VaultSpec(vilt=ViltSpec(num_hidden_layers=2), lm=LMSpec.bertweet_base(), n_classes=3)
    spec.lm.num_hidden_layers = 2
    state = build_state(spec, 3)
    bn = synthetic_batch(spec, 208, seed=708, n_classes=3)
    db = _dev(bn)
    from vault_amd import ops
    res = []
    # third run: the data-parallel step's way through the same kernels - dynamic tile scheduling in the GEMMs (persist bit 0)
    # and a stage listener, i.e. weight-gradient groups of LM_WGRAD_GROUP layers instead of the whole stack
    for hm, dp_like in ((True, False), (False, False), (True, True)):
        sched = ops.GEMM_SCHED
        try:
            ops.GEMM_SCHED = 3 if dp_like else 0
            eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.1, half=half)
            eng.HEAD_MAJOR, eng.HEAD_MAJOR_MIN_ROWS, eng.LM_WGRAD_GROUP = hm, 0, 1
            eng.drop_seed = 77
            out = eng.forward(db, train=True, labels=db["labels"], need_hidden=False)
            eng.zero_grad()
            eng.backward(after_layer=(lambda tag: None) if dp_like else None)
            torch.cuda.synchronize()
        finally:
            ops.GEMM_SCHED = sched
        assert (eng.last["qkv_hm"], eng.last["lm_qkv_hm"]) == ((eng.last["Mp"], eng.last["Mlp"]) if hm else (0, 0))
        res.append((out["logits"].clone(), float(out["loss"]), eng.params.g[:eng.params.n_train].clone(), eng.last["ctx1"].clone(),
                    eng.last["lm_ctx1"].clone()))
        del eng
    a, b, c = res
    for x in (b, c):
        assert torch.equal(a[0], x[0])                         # logits
        assert abs(a[1] - x[1]) < 1e-6                         # loss (a float-atomic sum over the samples: order only)
        assert torch.equal(a[3], x[3])                         # ViLT layer 1 attention output
        assert torch.equal(a[4], x[4])                         # LM layer 1 attention output (dropout on the probabilities)
        rel = float((a[2] - x[2]).norm() / x[2].norm())
        print(f"head-major vs {'row-major' if x is b else 'head-major under dynamic scheduling and per-layer groups'}: gradient rel diff {rel:.2e}")
        assert rel < 1e-5


def test_experiment_script_call_sequence(tmp_path):
    """The model-side calls of the reference's driver (ref: experiments/clsf_vault.py:196-220): from_pretrained with the
    script's keyword arguments -> resize_token_embeddings(len(tokenizer)) -> integrate_entities_into_model (resize, get,
    max-pool description rows into the new rows, set: vault/entity_linking.py:115-148, restated here with a stand-in
    tokenizer) -> .to(device) -> forward; checked against the oracle on the resulting state_dict."""
    import json
    from safetensors.torch import save_file
    from vault.models.vault import VaultForTMSC
    spec = _nodrop(VaultSpec.tiny(3, "roberta"))
    v, lm = spec.vilt, spec.lm
    state = build_state(spec, 0)
    for d, cfg, keys in (("vilt", {f: getattr(v, f) for f in ("vocab_size", "max_position_embeddings", "type_vocab_size",
                                                               "modality_type_vocab_size", "hidden_size", "num_hidden_layers",
                                                               "num_attention_heads", "intermediate_size", "layer_norm_eps",
                                                               "image_size", "patch_size", "num_channels")},
                          [k for k in state if not k.startswith(("bert.", "classifier."))]),
                         ("bert", dict(model_type="roberta", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                       **{f: getattr(lm, f) for f in ("vocab_size", "max_position_embeddings", "type_vocab_size",
                                                                      "hidden_size", "num_hidden_layers", "num_attention_heads",
                                                                      "intermediate_size", "layer_norm_eps", "pad_token_id")}),
                          [k for k in state if k.startswith("bert.")])):
        (tmp_path / d).mkdir()
        json.dump(cfg, open(tmp_path / d / "config.json", "w"))
        save_file({(k[5:] if d == "bert" else k): torch.from_numpy(state[k]) for k in keys}, str(tmp_path / d / "model.safetensors"))
    model = VaultForTMSC.from_pretrained(str(tmp_path / "vilt"), str(tmp_path / "bert"), freeze_lm=False, n_classes=3,
                                         vilt_dropout_prob=0.0, use_vilt_position_embeddings=False)

    class Tok:                       # a tokenizer two entity tokens larger than the checkpoint's vocabulary
        def __len__(self):
            return lm.vocab_size + 2

        def encode(self, text):
            return [0] + [5 + (ord(c) % 100) for c in text][:12] + [2]

    tok, descriptions = Tok(), ["a river in spain", "football club"]
    model.resize_token_embeddings(len(tok))
    # integrate_entities_into_model(model, descriptions, tok):
    model.resize_token_embeddings(len(tok))
    ecls = model.get_input_embeddings()
    emb = ecls.weight.clone()
    for i, desc in enumerate(reversed(descriptions)):
        emb[-(i + 1)] = emb[tok.encode(desc)].max(0)[0]
    ecls.weight = torch.nn.parameter.Parameter(emb)
    model.set_input_embeddings(ecls)
    model = model.to("cuda").eval()
    assert model._engine.spec.lm.vocab_size == len(tok)
    bn = synthetic_batch(model.spec, 3, seed=55, n_classes=3)
    bn["input_ids"][0, 3] = len(tok) - 1                              # the new entity tokens occur in the text
    bn["input_ids"][1, 2] = len(tok) - 2
    kw = {k: torch.from_numpy(x).cuda() for k, x in bn.items() if k != "labels"}
    with torch.no_grad():
        logits = model(**kw)
    sd = {k: t.detach().cpu().numpy() for k, t in model.state_dict().items()}
    assert np.array_equal(sd["bert.embeddings.word_embeddings.weight"], emb.detach().numpy())
    ref = O.vault_forward(O.to_torch_state(sd), model.spec, O.torch_batch(bn))
    assert (logits.cpu() - ref["logits"]).abs().max().item() < 3e-3


def test_external_torch_optimizer_through_the_module():
    """INTEGRATION.md's claim that the reference trainer's loop works unchanged: ``torch.optim.AdamW`` stepping
    ``model.parameters()`` (views of the fp32 master buffer) - the bf16 shadows every GEMM reads must follow.  Three
    steps against the oracle driven by the same optimizer class; without the refresh the loss would not move."""
    from vault_amd.models.vault import VaultForTMSC
    spec = _nodrop(VaultSpec.tiny(3, "roberta"))
    state = build_state(spec, 0)
    model = VaultForTMSC(spec.vilt, n_classes=3, vilt_dropout_prob=0.0, bert_config=spec.lm, _state=state).to("cuda").train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.0)
    bn = synthetic_batch(spec, 4, seed=61, n_classes=3)
    kw = {k: torch.from_numpy(x).cuda() for k, x in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(**kw), labels)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    P = O.to_torch_state(state, requires_grad=True)
    ropt = torch.optim.AdamW([p for p in P.values()], lr=1e-3, weight_decay=0.0)
    ref = []
    for _ in range(3):
        ropt.zero_grad()
        loss, _ = O.vault_loss(P, spec, O.torch_batch(bn))
        loss.backward()
        ropt.step()
        ref.append(float(loss.detach()))
    assert ref[2] < ref[0] - 0.05                                     # lr 1e-3: the loss moves visibly
    for a, b in zip(losses, ref):
        assert abs(a - b) < 0.05 * abs(ref[0] - ref[2]) + 5e-3, (losses, ref)


def test_inputs_embeds_and_image_embeds_vs_reference_golden():
    """``inputs_embeds`` (text embeddings handed to the LM in place of ids, ref model.py:170-200) and ``image_embeds`` +
    ``pixel_mask`` [B, L] (HF modeling_vilt.py:190-207; the path of ref TomViltForTMSC, tomvilt/model.py:281-287) through
    the module API against numbers produced by the reference classes: outputs, and the gradients that flow back to both
    inputs through the autograd bridge."""
    from vault_amd.models.vault import VaultForTMSC, VaultModel
    g = np.load(os.path.join(GOLD, "tiny_bert_embeds_inputs.npz"))
    spec = _nodrop(VaultSpec.tiny(3, "bert"))
    state = build_state(spec, 0)
    model = VaultForTMSC(spec.vilt, n_classes=3, vilt_dropout_prob=0.0, bert_config=spec.lm, _state=state).to("cuda").train()
    te = torch.from_numpy(g["inputs_embeds"]).cuda().requires_grad_(True)
    ie = torch.from_numpy(g["image_embeds"]).cuda().requires_grad_(True)
    kw = dict(inputs_embeds=te, image_embeds=ie, attention_mask=torch.from_numpy(g["attention_mask"]).cuda(),
              pixel_mask=torch.from_numpy(g["pixel_mask"]).cuda(), token_type_ids=torch.from_numpy(g["token_type_ids"]).cuda())
    logits = model(**kw)
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(g["labels"]).cuda())
    loss.backward()
    assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() < 3e-3
    assert abs(float(loss) - float(g["loss"])) < 2e-3
    for mine, ref in ((te.grad, g["d_inputs_embeds"]), (ie.grad, g["d_image_embeds"])):
        mine = mine.cpu().numpy()
        assert np.linalg.norm(mine - ref) < 5e-2 * np.linalg.norm(ref)
    assert float(model._engine.params.gr("bert.embeddings.word_embeddings.weight").abs().max()) == 0.0   # no lookup, no gradient
    gm = model._engine.params.gr("embeddings.token_type_embeddings.weight").cpu().numpy()
    assert np.linalg.norm(gm - g["grad_modality_type"]) < 5e-2 * np.linalg.norm(g["grad_modality_type"])
    # headless encoder, eval mode: hidden states of every valid row
    enc = VaultModel(spec.vilt, bert_config=spec.lm, _state={k: v for k, v in state.items() if not k.startswith("classifier.")}).to("cuda").eval()
    with torch.no_grad():
        o = enc(**{k: (v.detach() if torch.is_tensor(v) else v) for k, v in kw.items()})
    T = g["attention_mask"].shape[1]
    valid = np.concatenate([g["attention_mask"], g["pixel_mask"]], axis=1).astype(bool)
    h, hr = o.last_hidden_state.cpu().numpy(), g["last_hidden_state"]
    assert h.shape == hr.shape == (3, T + g["image_embeds"].shape[1], spec.vilt.hidden_size)
    assert np.abs(h[valid] - hr[valid]).max() < 1.5e-2 * np.abs(hr).max()
    assert np.abs(o.pooler_output.cpu().numpy() - g["pooler_output"]).max() < 1e-2
    # HF's errors for inconsistent inputs
    with pytest.raises(ValueError):
        enc(input_ids=torch.zeros(3, T, dtype=torch.long), inputs_embeds=te.detach(), image_embeds=ie.detach())
    with pytest.raises(ValueError):
        enc(inputs_embeds=te.detach())
