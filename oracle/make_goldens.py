"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Runs only where /root/reference exists.  It stub-imports
``/root/reference/vault/models/vault/model.py`` (SURVEY Appendix A recipe: the package
``__init__`` pulls torchvision/ekphrasis/emoji which are absent here), builds the
reference ``VaultForTMSC`` on HuggingFace ``ViltModel`` + ``RobertaModel``/``BertModel``
from configs, overwrites every parameter with ``vault_amd.spec.fill_param`` values,
applies the D1 fix (transformers 5.15 ignores ``position_embedding_type``; the reference
under its pinned transformers 4.48 skips ViLT's text position embeddings when an LM is
present), runs forward + backward on ``vault_amd.spec.synthetic_batch`` inputs and stores
inputs-independent expected outputs.  Nothing from the reference is copied: the files
hold numbers only (logits, pooled output, hidden-state slices, loss, gradient norms).

Usage:  python oracle/make_goldens.py            (writes tests/golden/*.npz)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vault_amd.spec import (LMSpec, VaultSpec, ViltSpec, build_state, param_entries, synthetic_batch,  # noqa: E402
                            synthetic_ragged_batch)

REF = "/root/reference"


def import_reference():
    for n in ["ekphrasis", "ekphrasis.classes", "ekphrasis.classes.tokenizer",
              "ekphrasis.classes.preprocessor", "emoji"]:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["ekphrasis.classes.tokenizer"].SocialTokenizer = object
    sys.modules["ekphrasis.classes.preprocessor"].TextPreProcessor = object
    sys.modules["emoji"].demojize = lambda x, **k: x

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    pkg = types.ModuleType("vault")
    pkg.__path__ = [f"{REF}/vault"]
    sys.modules["vault"] = pkg
    load("vault.utils", f"{REF}/vault/utils.py")
    return load("vault_model_ref", f"{REF}/vault/models/vault/model.py")


def hf_configs(spec: VaultSpec):
    from transformers import BertConfig, RobertaConfig, ViltConfig
    v = spec.vilt
    vc = ViltConfig(vocab_size=v.vocab_size, hidden_size=v.hidden_size, num_hidden_layers=v.num_hidden_layers,
                    num_attention_heads=v.num_attention_heads, intermediate_size=v.intermediate_size,
                    image_size=v.image_size, patch_size=v.patch_size, num_channels=v.num_channels,
                    max_position_embeddings=v.max_position_embeddings, type_vocab_size=v.type_vocab_size,
                    modality_type_vocab_size=v.modality_type_vocab_size, layer_norm_eps=v.layer_norm_eps)
    lm = spec.lm
    kw = dict(vocab_size=lm.vocab_size, max_position_embeddings=lm.max_position_embeddings,
              type_vocab_size=lm.type_vocab_size, hidden_size=lm.hidden_size,
              num_hidden_layers=lm.num_hidden_layers, num_attention_heads=lm.num_attention_heads,
              intermediate_size=lm.intermediate_size, layer_norm_eps=lm.layer_norm_eps,
              pad_token_id=lm.pad_token_id)
    if lm.kind == "roberta":
        lc = RobertaConfig(bos_token_id=0, eos_token_id=2, **kw)
    else:
        lc = BertConfig(**kw)
    return vc, lc


def build_reference_model(ref, spec: VaultSpec, seed: int = 0):
    vc, lc = hf_configs(spec)
    model = ref.VaultForTMSC(vc, n_classes=spec.n_classes, vilt_dropout_prob=0.1, bert_config=lc).eval()
    state = build_state(spec, seed)
    sd = model.state_dict()
    ours = {n for n, _, _ in param_entries(spec)}
    theirs = {k for k in sd.keys()}
    missing = theirs - ours
    extra = ours - theirs
    assert not extra, f"names not in reference state_dict: {sorted(extra)[:5]}"
    # the reference may carry non-parameter buffers (position_ids ...) - those stay
    for k in missing:
        assert "position_ids" in k or "token_type_ids" in k, f"unexpected reference key {k}"
    with torch.no_grad():
        for k, v in state.items():
            assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(v))
        if not spec.use_vilt_position_embeddings:
            # D1: transformers-4.48 semantics of ``position_embedding_type = "NOT_absolute"``
            model.embeddings.text_embeddings.position_embeddings.weight.zero_()
    return model, state


def run_reference(model, spec, batch_np):
    kw = dict(input_ids=torch.from_numpy(batch_np["input_ids"]),
              attention_mask=torch.from_numpy(batch_np["attention_mask"]),
              pixel_values=torch.from_numpy(batch_np["pixel_values"]),
              pixel_mask=torch.from_numpy(batch_np["pixel_mask"]))
    if "token_type_ids" in batch_np:
        kw["token_type_ids"] = torch.from_numpy(batch_np["token_type_ids"])
    torch.manual_seed(0)
    model.zero_grad(set_to_none=True)
    # the image part of the attention mask (which of the randomly ordered patch rows are real patches) is internal to
    # ViltEmbeddings.visual_embed: capture its return value
    captured = {}
    emb = model.embeddings
    orig_ve = emb.visual_embed

    def spy(*a, **k):
        r = orig_ve(*a, **k)
        captured["x_mask"] = r[1].detach().clone()
        return r

    emb.visual_embed = spy
    # VaultForTMSC.forward returns logits only; grab the encoder output through the mixin
    try:
        enc = super(type(model), model).forward(**kw)
    finally:
        emb.visual_embed = orig_ve
    logits = model.classifier(enc.pooler_output).squeeze(-1)
    if spec.n_classes == 1:
        # the single-logit fine-tune (Bloomberg): ref vault/models/vault/trainer.py:55-56 (BCEWithLogitsLoss on float labels)
        loss = torch.nn.BCEWithLogitsLoss()(logits, torch.from_numpy(batch_np["labels"]).float())
    else:
        loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(batch_np["labels"]))
    loss.backward()
    T = batch_np["input_ids"].shape[1]
    out = {
        "logits": logits.detach().numpy(),
        "pooler_output": enc.pooler_output.detach().numpy(),
        "hidden_text_cls": enc.last_hidden_state[:, : T + 1].detach().numpy(),
        "loss": np.float32(loss.item()),
    }
    # patch tokens come back in a random order (D3): store an order-free summary - of the REAL patches only when the
    # batch holds padded images (masked padding rows are arbitrary)
    pn = enc.last_hidden_state[:, T + 1:].detach().norm(dim=-1).numpy()
    xm = captured["x_mask"][:, 1:].numpy() != 0
    if xm.all():
        out["hidden_patch_sorted_norms"] = np.sort(pn, axis=1)
    else:
        out["valid_patch_counts"] = xm.sum(axis=1).astype(np.int64)
        out["hidden_valid_patch_sorted_norms"] = np.concatenate([np.sort(pn[b][xm[b]]) for b in range(pn.shape[0])])
    names, norms = [], []
    small = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        if k == "embeddings.text_embeddings.position_embeddings.weight" and not spec.use_vilt_position_embeddings:
            # D1: under the reference's pinned transformers 4.48 this table is skipped (no grad);
            # the 5.15 emulation adds a zeroed table, which still receives a gradient
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if k in ("classifier.1.weight", "classifier.1.bias", "layernorm.weight", "layernorm.bias",
                 "embeddings.cls_token", "embeddings.token_type_embeddings.weight",
                 "bert.embeddings.LayerNorm.weight", "encoder.layer.0.attention.attention.query.bias",
                 "bert.encoder.layer.0.attention.self.key.bias",
                 "bert.embeddings.position_embeddings.weight"):
            small["grad::" + k] = p.grad.detach().numpy().copy()
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, np.float64)
    out.update(small)
    return out


CASES = {
    # name: (spec factory, batch, data seed)
    "tiny_roberta": (lambda: VaultSpec.tiny(3, "roberta"), 3, 11),
    "tiny_bert": (lambda: VaultSpec.tiny(3, "bert"), 3, 12),
    "full_bertweet_b2": (lambda: VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3), 2, 13),
    "tiny_bert_bce_n1": (lambda: VaultSpec.tiny(1, "bert"), 5, 15),     # n_classes = 1: logits [B], BCE-with-logits, 0/1 labels
}
# BASELINE config 4: bert-base-uncased shapes (vocab 30522, 512 positions, 2 token types, eps 1e-12) with the LM frozen
# the way the reference's from_pretrained(freeze_lm=True) freezes it (ref: vault/models/vault/model.py:124-126:
# ``freeze_lm`` attribute + ``set_parameter_requires_grad(model.bert, False)``; forward then runs the LM under no_grad, :189)
FROZEN_CASES = {
    "full_bert_base_frozen_b2": (lambda: VaultSpec(vilt=ViltSpec(), lm=LMSpec.bert_base_uncased(), n_classes=3), 2, 14),
}
# padded batches of different image sizes: name -> (spec factory, valid (h, w) per sample, canvas, seed).
# tiny: patch 16, position table 12 x 12; canvas 192 x 256 = 12 x 16 patches: one full-canvas image (192 patches: the
# sequence is longer than the table), one small image (10 x 7 patches, 122 masked padding rows), one square image
RAGGED_CASES = {
    "tiny_roberta_ragged": (lambda: VaultSpec.tiny(3, "roberta"), [(192, 256), (160, 112), (192, 192)], (192, 256), 21),
    "tiny_bert_ragged_small": (lambda: VaultSpec.tiny(3, "bert"), [(96, 160), (128, 64), (80, 80), (128, 160)], (128, 160), 22),
}


def flag_case_inputs(spec, B, dseed):
    """Inputs + fixed output-gradient weights of the VaultModel flag case (shared with the tests)."""
    batch = synthetic_batch(spec, B, seed=dseed, n_classes=3)
    rng = np.random.Generator(np.random.PCG64(dseed + 1))
    T = batch["input_ids"].shape[1]
    batch["token_type_ids"] = (rng.random((B, T)) < 0.4).astype(np.int64) * batch["attention_mask"]
    wp = rng.standard_normal((B, spec.vilt.hidden_size)).astype(np.float32)
    wh = rng.standard_normal((B, T + 1, spec.vilt.hidden_size)).astype(np.float32) * 0.1
    return batch, wp, wh


def run_reference_vaultmodel_flags(ref, name, outdir):
    """Headless VaultModel with freeze_lm=True and use_vilt_position_embeddings=True (ref: model.py:53-91): the LM
    receives no gradient, ViLT's own text position table is used and trained, BERT token types 0/1 are live.
    Scalar objective: <pooler_output, Wp> + <last_hidden_state[:, :T+1], Wh> with fixed Wp, Wh."""
    spec = VaultSpec.tiny(0, "bert")
    spec.use_vilt_position_embeddings = True
    vc, lc = hf_configs(spec)
    model = ref.VaultModel(vc, bert_config=lc, freeze_lm=True, vilt_dropout_prob=0.0,
                           use_vilt_position_embeddings=True).eval()
    state = build_state(spec, 0)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in state.items():
            assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(v))
    B, dseed = 3, 31
    batch, wp, wh = flag_case_inputs(spec, B, dseed)
    kw = {k: torch.from_numpy(batch[k]) for k in ("input_ids", "attention_mask", "token_type_ids", "pixel_values",
                                                  "pixel_mask")}
    torch.manual_seed(0)
    enc = model(**kw)
    T = batch["input_ids"].shape[1]
    obj = (enc.pooler_output * torch.from_numpy(wp)).sum() + (enc.last_hidden_state[:, : T + 1] * torch.from_numpy(wh)).sum()
    obj.backward()
    out = {"pooler_output": enc.pooler_output.detach().numpy(),
           "hidden_text_cls": enc.last_hidden_state[:, : T + 1].detach().numpy(),
           "hidden_patch_sorted_norms": np.sort(enc.last_hidden_state[:, T + 1:].detach().norm(dim=-1).numpy(), axis=1),
           "objective": np.float32(obj.item()), "meta_batch": np.int64(B), "meta_data_seed": np.int64(dseed)}
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if k in ("embeddings.text_embeddings.position_embeddings.weight", "embeddings.cls_token",
                 "embeddings.text_embeddings.token_type_embeddings.weight", "pooler.dense.bias"):
            out["grad::" + k] = p.grad.detach().numpy().copy()
    assert not any(n.startswith("bert.") for n in names), "freeze_lm: the LM must not receive gradients"
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array(norms, np.float64)
    path = os.path.join(outdir, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "objective", out["objective"], "n grads", len(names), "->", path, f"{os.path.getsize(path)/1024:.0f} KiB")


def run_reference_itr(ref, name, outdir):
    """VaultForImageAndTextRetrieval (ref: model.py:375-405; HF ViltForImageAndTextRetrieval: Linear(H, 1) on the pooled
    output, encoder under the `vilt.` prefix).  Objective: <logits, w>."""
    spec = VaultSpec.tiny(1, "roberta")
    vc, lc = hf_configs(spec)
    model = ref.VaultForImageAndTextRetrieval(vc, bert_config=lc).eval()
    state = build_state(spec, 0)
    sd = model.state_dict()

    def ext(n):   # build name -> reference state_dict key
        if n.startswith("bert."):
            return n
        if n.startswith("classifier.1."):
            return "rank_output." + n[len("classifier.1."):]
        return "vilt." + n

    with torch.no_grad():
        for k, v in state.items():
            assert tuple(sd[ext(k)].shape) == tuple(v.shape), (k, sd[ext(k)].shape, v.shape)
            sd[ext(k)].copy_(torch.from_numpy(v))
        model.vilt.embeddings.text_embeddings.position_embeddings.weight.zero_()   # D1
    unexpected = [k for k in sd if k not in {ext(n) for n in state} and "position_ids" not in k and "token_type_ids" not in k]
    assert not unexpected, unexpected[:5]
    B, dseed = 3, 41
    batch = synthetic_batch(spec, B, seed=dseed, n_classes=1)
    rng = np.random.Generator(np.random.PCG64(dseed + 1))
    w = rng.standard_normal((B, 1)).astype(np.float32)
    kw = {k: torch.from_numpy(batch[k]) for k in ("input_ids", "attention_mask", "pixel_values", "pixel_mask")}
    torch.manual_seed(0)
    out = model(**kw)
    obj = (out.logits * torch.from_numpy(w)).sum()
    obj.backward()
    res = {"logits": out.logits.detach().numpy(), "objective": np.float32(obj.item()), "w": w,
           "meta_batch": np.int64(B), "meta_data_seed": np.int64(dseed)}
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is None or k == "vilt.embeddings.text_embeddings.position_embeddings.weight":
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if k in ("rank_output.weight", "rank_output.bias", "vilt.pooler.dense.bias", "vilt.embeddings.cls_token"):
            res["grad::" + k] = p.grad.detach().numpy().copy()
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array(norms, np.float64)
    path = os.path.join(outdir, f"{name}.npz")
    np.savez_compressed(path, **res)
    print(name, "logits", res["logits"].ravel(), "n grads", len(names), "->", path, f"{os.path.getsize(path)/1024:.0f} KiB")


def run_reference_vqa(ref, name, outdir):
    """VaultForQuestionAnswering (ref: model.py:472-509 on HF ViltForQuestionAnswering: Linear(H, 2H) - LayerNorm - GELU -
    Linear(2H, n_classes) on the pooled output, loss = BCE-with-logits * n_classes).  n_classes = 10 exercises
    ``renew_classifier``; the loss is the reference's own."""
    L = 10
    spec = VaultSpec.tiny(L, "roberta")
    spec.head = "mlp"
    vc, lc = hf_configs(spec)
    model = ref.VaultForQuestionAnswering(vc, bert_config=lc, n_classes=L).eval()
    state = build_state(spec, 0)
    sd = model.state_dict()

    def ext(n):
        return n if n.startswith(("bert.", "classifier.")) else "vilt." + n

    with torch.no_grad():
        for k, v in state.items():
            assert tuple(sd[ext(k)].shape) == tuple(v.shape), (k, sd[ext(k)].shape, v.shape)
            sd[ext(k)].copy_(torch.from_numpy(v))
        model.vilt.embeddings.text_embeddings.position_embeddings.weight.zero_()   # D1
    unexpected = [k for k in sd if k not in {ext(n) for n in state} and "position_ids" not in k and "token_type_ids" not in k]
    assert not unexpected, unexpected[:5]
    B, dseed = 3, 51
    batch = synthetic_batch(spec, B, seed=dseed, n_classes=1)
    rng = np.random.Generator(np.random.PCG64(dseed + 1))
    labels = (rng.random((B, L)) < 0.3).astype(np.float32) * rng.random((B, L)).astype(np.float32)   # soft VQA scores
    kw = {k: torch.from_numpy(batch[k]) for k in ("input_ids", "attention_mask", "pixel_values", "pixel_mask")}
    torch.manual_seed(0)
    out = model(**kw, labels=torch.from_numpy(labels))
    out.loss.backward()
    res = {"logits": out.logits.detach().numpy(), "loss": np.float32(out.loss.item()), "labels": labels,
           "meta_batch": np.int64(B), "meta_data_seed": np.int64(dseed)}
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is None or k == "vilt.embeddings.text_embeddings.position_embeddings.weight":
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if (k.startswith("classifier.") and k != "classifier.0.weight") or k in ("vilt.pooler.dense.bias",):
            res["grad::" + k] = p.grad.detach().numpy().copy()
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array(norms, np.float64)
    path = os.path.join(outdir, f"{name}.npz")
    np.savez_compressed(path, **res)
    print(name, "loss", res["loss"], "logits", res["logits"].ravel()[:4], "n grads", len(names), "->", path,
          f"{os.path.getsize(path)/1024:.0f} KiB")


def run_reference_nlvr2(ref, name, outdir):
    """VaultForImagesAndTextClassification (ref: model.py:408-465 on HF ViltForImagesAndTextClassification): two images
    per sample, one encoder pass each with modality type i + 1 (table resized to 3 rows), MLP head on the
    concatenated pooled outputs, CE loss."""
    spec = VaultSpec.tiny(2, "roberta")
    spec.head, spec.num_images = "mlp", 2
    spec.vilt.modality_type_vocab_size = 3
    vc, lc = hf_configs(spec)
    vc.num_images = 2
    vc.num_labels = 2
    vc.modality_type_vocab_size = 2      # the reference resizes the table itself (resize_token_type_embeddings)
    model = ref.VaultForImagesAndTextClassification(vc, bert_config=lc).eval()
    state = build_state(spec, 0)
    sd = model.state_dict()
    ext = lambda n: n if n.startswith(("bert.", "classifier.")) else "vilt." + n   # noqa: E731
    with torch.no_grad():
        for k, v in state.items():
            assert tuple(sd[ext(k)].shape) == tuple(v.shape), (k, sd[ext(k)].shape, v.shape)
            sd[ext(k)].copy_(torch.from_numpy(v))
        model.vilt.embeddings.text_embeddings.position_embeddings.weight.zero_()   # D1
    unexpected = [k for k in sd if k not in {ext(n) for n in state} and "position_ids" not in k and "token_type_ids" not in k]
    assert not unexpected, unexpected[:5]
    B, dseed = 3, 61
    batch = synthetic_batch(spec, B, seed=dseed, n_classes=2)
    rng = np.random.Generator(np.random.PCG64(dseed + 1))
    img = spec.vilt.image_size
    pix2 = np.clip(rng.standard_normal((B, 2, spec.vilt.num_channels, img, img), dtype=np.float32), -1.0, 1.0)
    kw = {k: torch.from_numpy(batch[k]) for k in ("input_ids", "attention_mask")}
    torch.manual_seed(0)
    out = model(**kw, pixel_values=torch.from_numpy(pix2), labels=torch.from_numpy(batch["labels"]))
    out.loss.backward()
    res = {"logits": out.logits.detach().numpy(), "loss": np.float32(out.loss.item()),
           "meta_batch": np.int64(B), "meta_data_seed": np.int64(dseed)}
    names, norms = [], []
    for k, p in model.named_parameters():
        if p.grad is None or k == "vilt.embeddings.text_embeddings.position_embeddings.weight":
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if (k.startswith("classifier.") and k != "classifier.0.weight") or k in ("vilt.embeddings.token_type_embeddings.weight",
                                                                               "vilt.pooler.dense.bias"):
            res["grad::" + k] = p.grad.detach().numpy().copy()
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array(norms, np.float64)
    path = os.path.join(outdir, f"{name}.npz")
    np.savez_compressed(path, **res)
    print(name, "loss", res["loss"], "logits", res["logits"].ravel(), "n grads", len(names), "->", path,
          f"{os.path.getsize(path)/1024:.0f} KiB")


def run_reference_mlm(ref, name, outdir):
    """VaultForMaskedLM (ref: model.py:467-469 on HF ViltForMaskedLM): ViltMLMHead on the text rows of the fused
    sequence, decoder tied to ViLT's word embeddings, CE over ViLT's vocabulary with ignore_index -100."""
    spec = VaultSpec.tiny(0, "roberta")
    spec.head = "mlm"
    vc, lc = hf_configs(spec)
    model = ref.VaultForMaskedLM(vc, bert_config=lc).eval()
    state = build_state(spec, 0)
    sd = model.state_dict()
    def ext(n):   # (transformers 5.15 keeps the vocabulary bias as mlm_score.decoder.bias; 4.48 also as mlm_score.bias)
        if n == "mlm_score.bias":
            return "mlm_score.decoder.bias"
        return n if n.startswith(("bert.", "mlm_score.")) else "vilt." + n

    with torch.no_grad():
        for k, v in state.items():
            assert tuple(sd[ext(k)].shape) == tuple(v.shape), (k, sd[ext(k)].shape, v.shape)
            sd[ext(k)].copy_(torch.from_numpy(v))
        model.vilt.embeddings.text_embeddings.position_embeddings.weight.zero_()   # D1
    # the decoder must be tied to the (just overwritten) word embeddings and to mlm_score.bias
    assert model.mlm_score.decoder.weight.data_ptr() == model.vilt.embeddings.text_embeddings.word_embeddings.weight.data_ptr()
    B, dseed = 3, 71
    batch = synthetic_batch(spec, B, seed=dseed, n_classes=1)
    rng = np.random.Generator(np.random.PCG64(dseed + 1))
    T = batch["input_ids"].shape[1]
    labels = rng.integers(0, spec.vilt.vocab_size, size=(B, T), dtype=np.int64)
    labels[rng.random((B, T)) > 0.3] = -100
    labels[batch["attention_mask"] == 0] = -100
    kw = {k: torch.from_numpy(batch[k]) for k in ("input_ids", "attention_mask", "pixel_values", "pixel_mask")}
    torch.manual_seed(0)
    out = model(**kw, labels=torch.from_numpy(labels))
    out.loss.backward()
    res = {"logits_slice": out.logits.detach().numpy()[:, :4], "logits_norm": np.float64(out.logits.detach().double().norm()),
           "loss": np.float32(out.loss.item()), "labels": labels, "meta_batch": np.int64(B), "meta_data_seed": np.int64(dseed)}
    names, norms = [], []
    seen = set()
    for k, p in model.named_parameters():
        if p.grad is None or k == "vilt.embeddings.text_embeddings.position_embeddings.weight" or id(p) in seen:
            continue
        seen.add(id(p))
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        if k.startswith("mlm_score.") and p.grad.numel() <= 4096:
            res["grad::" + k] = p.grad.detach().numpy().copy()
    res["grad_names"] = np.array(names)
    res["grad_norms"] = np.array(norms, np.float64)
    path = os.path.join(outdir, f"{name}.npz")
    np.savez_compressed(path, **res)
    print(name, "loss", res["loss"], "n grads", len(names), [n for n in names if "mlm" in n or "word_emb" in n], "->", path,
          f"{os.path.getsize(path)/1024:.0f} KiB")


def run_reference_embeds(ref, name, outdir):
    """``inputs_embeds`` + ``image_embeds`` through the reference classes (ref model.py:170-200 hands inputs_embeds to the
    LM; HF ViltEmbeddings.forward takes image_embeds as they are, modeling_vilt.py:190-207; the path TomViltForTMSC uses,
    ref: vault/models/tomvilt/model.py:281-287): logits, pooled output, hidden states and the gradients of both inputs."""
    spec = VaultSpec.tiny(3, "bert")
    model, _ = build_reference_model(ref, spec, seed=0)
    B, T, L, H = 3, 40, 37, spec.vilt.hidden_size
    rng = np.random.default_rng(31)
    te = torch.from_numpy((rng.standard_normal((B, T, H)) * 0.5).astype(np.float32)).requires_grad_(True)
    ie = torch.from_numpy((rng.standard_normal((B, L, H)) * 0.5).astype(np.float32)).requires_grad_(True)
    am = np.ones((B, T), np.int64); am[1, 29:] = 0
    pm = np.ones((B, L), np.int64); pm[2, 30:] = 0
    tt = np.zeros((B, T), np.int64); tt[:, 20:] = 1
    labels = np.array([0, 2, 1], np.int64)
    model.zero_grad(set_to_none=True)
    enc = super(type(model), model).forward(inputs_embeds=te, image_embeds=ie, attention_mask=torch.from_numpy(am),
                                            pixel_mask=torch.from_numpy(pm), token_type_ids=torch.from_numpy(tt))
    logits = model.classifier(enc.pooler_output).squeeze(-1)
    loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(labels))
    loss.backward()
    out = dict(inputs_embeds=te.detach().numpy(), image_embeds=ie.detach().numpy(), attention_mask=am, pixel_mask=pm,
               token_type_ids=tt, labels=labels, logits=logits.detach().numpy(),
               pooler_output=enc.pooler_output.detach().numpy(), last_hidden_state=enc.last_hidden_state.detach().numpy(),
               loss=np.float32(loss.item()), d_inputs_embeds=te.grad.numpy(), d_image_embeds=ie.grad.numpy(),
               grad_norm_word_embeddings=np.float64(0.0 if model.bert.embeddings.word_embeddings.weight.grad is None else
                                                    float(model.bert.embeddings.word_embeddings.weight.grad.norm())),
               grad_modality_type=model.embeddings.token_type_embeddings.weight.grad.numpy())
    path = os.path.join(outdir, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "loss", out["loss"], "logits", out["logits"].ravel()[:6], "->", path, f"{os.path.getsize(path)/1024:.0f} KiB")


def nlvr2_pixels(spec, B, dseed):
    """The two-image pixel tensor of the NLVR2 case (shared with the tests)."""
    rng = np.random.Generator(np.random.PCG64(dseed + 1))
    img = spec.vilt.image_size
    return np.clip(rng.standard_normal((B, 2, spec.vilt.num_channels, img, img), dtype=np.float32), -1.0, 1.0)


def main():
    ref = import_reference()
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    only = sys.argv[1:]
    for name, (mk, B, dseed) in CASES.items():
        if only and name not in only:
            continue
        spec = mk()
        model, _ = build_reference_model(ref, spec, seed=0)
        batch = synthetic_batch(spec, B, seed=dseed, n_classes=max(spec.n_classes, 2))   # (single logit: 0 / 1 targets)
        # make sure one caption is short so padding/masking is exercised
        out = run_reference(model, spec, batch)
        out["meta_batch"] = np.int64(B)
        out["meta_data_seed"] = np.int64(dseed)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, "loss", out["loss"], "logits", out["logits"].ravel()[:6], "->", path,
              f"{os.path.getsize(path)/1024:.0f} KiB")
    for name, (mk, B, dseed) in FROZEN_CASES.items():
        if only and name not in only:
            continue
        spec = mk()
        model, _ = build_reference_model(ref, spec, seed=0)
        model.freeze_lm = True
        sys.modules["vault.utils"].set_parameter_requires_grad(model.bert, False)
        batch = synthetic_batch(spec, B, seed=dseed, n_classes=spec.n_classes)
        out = run_reference(model, spec, batch)
        assert not any(str(n).startswith("bert.") for n in out["grad_names"]), "frozen LM received gradients"
        out["meta_batch"] = np.int64(B)
        out["meta_data_seed"] = np.int64(dseed)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, "loss", out["loss"], "logits", out["logits"].ravel()[:6], "->", path,
              f"{os.path.getsize(path)/1024:.0f} KiB")
    if not only or "tiny_bert_embeds_inputs" in only:
        run_reference_embeds(ref, "tiny_bert_embeds_inputs", outdir)
    if not only or "tiny_roberta_mlm" in only:
        run_reference_mlm(ref, "tiny_roberta_mlm", outdir)
    if not only or "tiny_roberta_nlvr2" in only:
        run_reference_nlvr2(ref, "tiny_roberta_nlvr2", outdir)
    if not only or "tiny_roberta_vqa" in only:
        run_reference_vqa(ref, "tiny_roberta_vqa", outdir)
    if not only or "tiny_roberta_itr" in only:
        run_reference_itr(ref, "tiny_roberta_itr", outdir)
    if not only or "tiny_bert_vaultmodel_flags" in only:
        run_reference_vaultmodel_flags(ref, "tiny_bert_vaultmodel_flags", outdir)
    for name, (mk, valid_hw, pad_hw, dseed) in RAGGED_CASES.items():
        if only and name not in only:
            continue
        spec = mk()
        model, _ = build_reference_model(ref, spec, seed=0)
        batch = synthetic_ragged_batch(spec, valid_hw, pad_hw, seed=dseed, n_classes=spec.n_classes)
        out = run_reference(model, spec, batch)
        out["meta_valid_hw"] = np.array(valid_hw, np.int64)
        out["meta_pad_hw"] = np.array(pad_hw, np.int64)
        out["meta_data_seed"] = np.int64(dseed)
        path = os.path.join(outdir, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(name, "loss", out["loss"], "logits", out["logits"].ravel()[:6], "valid", out.get("valid_patch_counts"),
              "->", path, f"{os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
