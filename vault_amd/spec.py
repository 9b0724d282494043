"""Architecture description, parameter inventory and the deterministic weight filler.

The parameter names are the HuggingFace ``state_dict`` keys the reference relies on
(ref: vault/models/vault/model.py:92-128 loads a ``ViltModel`` checkpoint into the
object itself and a BERT-family LM under ``.bert``; ``VaultForTMSC`` adds
``classifier.1`` at model.py:547-550).  Shapes follow
HF:models/vilt/modeling_vilt.py:68-300,303-492,651-663 and
HF:models/roberta/modeling_roberta.py:56-155 / HF:models/bert/modeling_bert.py:53-107.

This module is host-side product code (numpy only).  Both the HIP engine and the
oracle tests build their weights through :func:`fill_param`, so the two sides of a
parity test hold bit-identical fp32 weights without shipping 246 M numbers.
"""
from __future__ import annotations

import dataclasses
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np


@dataclasses.dataclass
class LMSpec:
    """BERT-family language model shape (ref: model.py:83-87 builds it via AutoModel)."""

    vocab_size: int = 64001           # vinai/bertweet-base
    max_position_embeddings: int = 130
    type_vocab_size: int = 1
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 1
    # "roberta": position ids = cumsum(ids != pad) * (ids != pad) + pad
    #            (HF:models/roberta/modeling_roberta.py:142-155)
    # "bert":    position ids = arange(T)  (HF:models/bert/modeling_bert.py:85-86)
    kind: str = "roberta"
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1

    @staticmethod
    def bertweet_base() -> "LMSpec":
        return LMSpec()

    @staticmethod
    def bert_base_uncased() -> "LMSpec":
        return LMSpec(vocab_size=30522, max_position_embeddings=512, type_vocab_size=2,
                      layer_norm_eps=1e-12, pad_token_id=0, kind="bert")


@dataclasses.dataclass
class ViltSpec:
    """ViLT-B/32 shape == ``transformers.ViltConfig()`` defaults."""

    vocab_size: int = 30522
    max_position_embeddings: int = 40
    type_vocab_size: int = 2
    modality_type_vocab_size: int = 2
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    layer_norm_eps: float = 1e-12
    image_size: int = 384
    patch_size: int = 32
    num_channels: int = 3
    max_image_length: int = -1     # ViltConfig.max_image_length: cap on the image part of the sequence (-1: none)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid


@dataclasses.dataclass
class VaultSpec:
    vilt: ViltSpec = dataclasses.field(default_factory=ViltSpec)
    lm: Optional[LMSpec] = dataclasses.field(default_factory=LMSpec)
    n_classes: int = 0                 # 0 => VaultModel (no classifier head)
    head: str = "linear"               # "linear": Dropout-Linear on the pooled output (TMSC, retrieval rank head);
                                       # "mlp": Linear(H,2H)-LayerNorm-GELU-Linear(2H,n_classes) (HF VQA head)
                                       # "mlm": HF ViltMLMHead on the text rows (dense-GELU-LayerNorm, decoder tied to
                                       #        ViLT's word embeddings, n_classes = ViLT vocab size)
    num_images: int = 1                # > 1: HF ViltForImagesAndTextClassification - one encoder pass per image, the MLP
                                       # head (Linear(nH,nH)-LayerNorm-GELU-Linear(nH,n_classes)) on the concatenated
                                       # pooled outputs

    @property
    def mlp_dims(self) -> Tuple[int, int]:
        """(input width, hidden width) of the MLP head."""
        H = self.vilt.hidden_size
        return (H, 2 * H) if self.num_images == 1 else (self.num_images * H, self.num_images * H)
    use_vilt_position_embeddings: bool = False
    add_pooling_layer: bool = True

    @staticmethod
    def tiny(n_classes: int = 3, lm_kind: str = "roberta") -> "VaultSpec":
        """Narrow/shallow configuration for quick CPU tests (SURVEY §8c golden set 1).

        Head dim stays 64 and the image stays a 12x12 grid of patches so the HIP
        kernels (which are specialised for d=64 and S=40+145) run the same code
        path as at full width; only hidden/FFN/depth/vocab shrink.
        """
        vilt = ViltSpec(vocab_size=128, hidden_size=256, num_hidden_layers=2,
                        num_attention_heads=4, intermediate_size=512,
                        image_size=192, patch_size=16)
        if lm_kind == "roberta":
            lm = LMSpec(vocab_size=160, max_position_embeddings=50, hidden_size=256,
                        num_hidden_layers=2, num_attention_heads=4, intermediate_size=512)
        else:
            lm = LMSpec(vocab_size=160, max_position_embeddings=64, type_vocab_size=2,
                        hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                        intermediate_size=512, layer_norm_eps=1e-12, pad_token_id=0,
                        kind="bert")
        return VaultSpec(vilt=vilt, lm=lm, n_classes=n_classes)


# --------------------------------------------------------------------------------------
# parameter inventory
# --------------------------------------------------------------------------------------
# init kinds: "normal" (std 0.02), "zeros", "ones".  HF initialises Linear/Embedding/Conv
# weights N(0, 0.02), biases 0, LayerNorm (1, 0); cls_token / position_embeddings of ViLT
# are zero-initialised Parameters.  The filler below uses N(0, 0.02) for *every*
# non-LayerNorm-weight tensor (biases, cls token and position tables included) so that a
# parity test exercises every add in the path; LayerNorm weights are 1 + N(0, 0.02).

ParamEntry = Tuple[str, Tuple[int, ...], str]


def _layer_entries(prefix: str, H: int, FF: int, style: str) -> List[ParamEntry]:
    e: List[ParamEntry] = []
    att = "attention.attention" if style == "vilt" else "attention.self"
    for n in ("query", "key", "value"):
        e.append((f"{prefix}.{att}.{n}.weight", (H, H), "normal"))
        e.append((f"{prefix}.{att}.{n}.bias", (H,), "normal"))
    e.append((f"{prefix}.attention.output.dense.weight", (H, H), "normal"))
    e.append((f"{prefix}.attention.output.dense.bias", (H,), "normal"))
    if style == "bert":
        e.append((f"{prefix}.attention.output.LayerNorm.weight", (H,), "ln_w"))
        e.append((f"{prefix}.attention.output.LayerNorm.bias", (H,), "normal"))
    e.append((f"{prefix}.intermediate.dense.weight", (FF, H), "normal"))
    e.append((f"{prefix}.intermediate.dense.bias", (FF,), "normal"))
    e.append((f"{prefix}.output.dense.weight", (H, FF), "normal"))
    e.append((f"{prefix}.output.dense.bias", (H,), "normal"))
    if style == "bert":
        e.append((f"{prefix}.output.LayerNorm.weight", (H,), "ln_w"))
        e.append((f"{prefix}.output.LayerNorm.bias", (H,), "normal"))
    else:
        e.append((f"{prefix}.layernorm_before.weight", (H,), "ln_w"))
        e.append((f"{prefix}.layernorm_before.bias", (H,), "normal"))
        e.append((f"{prefix}.layernorm_after.weight", (H,), "ln_w"))
        e.append((f"{prefix}.layernorm_after.bias", (H,), "normal"))
    return e


def param_entries(spec: VaultSpec) -> List[ParamEntry]:
    """All ``state_dict`` keys in HF order: ViLT part, then ``bert.*``, then the head."""
    v = spec.vilt
    H = v.hidden_size
    e: List[ParamEntry] = [
        ("embeddings.cls_token", (1, 1, H), "normal"),
        ("embeddings.position_embeddings", (1, v.num_patches + 1, H), "normal"),
        ("embeddings.text_embeddings.word_embeddings.weight", (v.vocab_size, H), "normal"),
        ("embeddings.text_embeddings.position_embeddings.weight", (v.max_position_embeddings, H), "normal"),
        ("embeddings.text_embeddings.token_type_embeddings.weight", (v.type_vocab_size, H), "normal"),
        ("embeddings.text_embeddings.LayerNorm.weight", (H,), "ln_w"),
        ("embeddings.text_embeddings.LayerNorm.bias", (H,), "normal"),
        ("embeddings.patch_embeddings.projection.weight", (H, v.num_channels, v.patch_size, v.patch_size), "normal"),
        ("embeddings.patch_embeddings.projection.bias", (H,), "normal"),
        ("embeddings.token_type_embeddings.weight", (v.modality_type_vocab_size, H), "normal"),
    ]
    for i in range(v.num_hidden_layers):
        e += _layer_entries(f"encoder.layer.{i}", H, v.intermediate_size, "vilt")
    e += [("layernorm.weight", (H,), "ln_w"), ("layernorm.bias", (H,), "normal")]
    if spec.add_pooling_layer:
        e += [("pooler.dense.weight", (H, H), "normal"), ("pooler.dense.bias", (H,), "normal")]
    if spec.lm is not None:
        lm = spec.lm
        HL = lm.hidden_size
        e += [
            ("bert.embeddings.word_embeddings.weight", (lm.vocab_size, HL), "normal"),
            ("bert.embeddings.position_embeddings.weight", (lm.max_position_embeddings, HL), "normal"),
            ("bert.embeddings.token_type_embeddings.weight", (lm.type_vocab_size, HL), "normal"),
            ("bert.embeddings.LayerNorm.weight", (HL,), "ln_w"),
            ("bert.embeddings.LayerNorm.bias", (HL,), "normal"),
        ]
        for i in range(lm.num_hidden_layers):
            e += _layer_entries(f"bert.encoder.layer.{i}", HL, lm.intermediate_size, "bert")
    if spec.head == "mlm":
        e += [("mlm_score.transform.dense.weight", (H, H), "normal"), ("mlm_score.transform.dense.bias", (H,), "normal"),
              ("mlm_score.transform.LayerNorm.weight", (H,), "ln_w"), ("mlm_score.transform.LayerNorm.bias", (H,), "normal"),
              ("mlm_score.bias", (v.vocab_size,), "normal")]
    elif spec.n_classes > 0 and spec.head == "mlp":
        hin, hmid = spec.mlp_dims
        e += [("classifier.0.weight", (hmid, hin), "normal"), ("classifier.0.bias", (hmid,), "normal"),
              ("classifier.1.weight", (hmid,), "ln_w"), ("classifier.1.bias", (hmid,), "normal"),
              ("classifier.3.weight", (spec.n_classes, hmid), "normal"), ("classifier.3.bias", (spec.n_classes,), "normal")]
    elif spec.n_classes > 0:
        e += [("classifier.1.weight", (spec.n_classes, H), "normal"),
              ("classifier.1.bias", (spec.n_classes,), "normal")]
    return e


def fill_param(name: str, shape: Tuple[int, ...], kind: str, seed: int = 0) -> np.ndarray:
    """Deterministic fp32 values for one tensor: a PCG64 stream keyed by (seed, crc32(name)).

    Independent of torch's RNG, of the order tensors are created in and of the machine,
    so the container that wrote ``tests/golden`` and the GPU box regenerate identical
    weights.
    """
    key = zlib.crc32(name.encode("utf-8"))
    rng = np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, key]))
    n = int(np.prod(shape))
    if kind == "zeros":
        return np.zeros(shape, np.float32)
    x = rng.standard_normal(n, dtype=np.float32) * np.float32(0.02)
    if kind == "ln_w":
        x = x + np.float32(1.0)
    elif kind == "ones":
        x = np.ones(n, np.float32)
    return x.reshape(shape)


def build_state(spec: VaultSpec, seed: int = 0) -> Dict[str, np.ndarray]:
    return {n: fill_param(n, s, k, seed) for n, s, k in param_entries(spec)}


# --------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------
def synthetic_batch(spec: VaultSpec, batch: int, seed: int = 1234, text_len: int = 40,
                    min_len: int = 8, n_classes: int = 3) -> Dict[str, np.ndarray]:
    """Synthetic image+caption pairs shaped like ``VaultTrainerForTMSC.input_batch_kwargs``
    (ref: vault/models/vault/trainer.py:19-36): int64 ids/mask, fp32 NCHW pixels, full
    pixel mask; per-sample caption length U{min_len..text_len}, BOS at 0, EOS last, pad after.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    lm = spec.lm
    vocab = lm.vocab_size if lm is not None else spec.vilt.vocab_size
    pad = lm.pad_token_id if lm is not None else 0
    img = spec.vilt.image_size
    lo = 3 if vocab > 8 else 0
    ids = rng.integers(lo, vocab, size=(batch, text_len), dtype=np.int64)
    lens = rng.integers(min_len, text_len + 1, size=(batch,))
    mask = (np.arange(text_len)[None, :] < lens[:, None]).astype(np.int64)
    ids[:, 0] = 0 if pad != 0 else 2
    for b in range(batch):
        ids[b, lens[b] - 1] = 2 if pad != 2 else 3
    ids = np.where(mask == 1, ids, pad).astype(np.int64)
    pix = rng.standard_normal((batch, spec.vilt.num_channels, img, img), dtype=np.float32)
    pix = np.clip(pix, -1.0, 1.0)
    out = {
        "input_ids": ids,
        "attention_mask": mask,
        "pixel_values": pix,
        "pixel_mask": np.ones((batch, img, img), np.int64),
        "labels": rng.integers(0, max(n_classes, 1), size=(batch,), dtype=np.int64),
    }
    if lm is not None and lm.type_vocab_size > 1:
        out["token_type_ids"] = np.zeros((batch, text_len), np.int64)
    return out


def synthetic_ragged_batch(spec: VaultSpec, valid_hw, pad_hw, seed: int = 1234, text_len: int = 40,
                           n_classes: int = 3) -> Dict[str, np.ndarray]:
    """Like :func:`synthetic_batch`, but with images of different sizes padded to a common
    ``pad_hw = (Hp, Wp)`` canvas the way the HF ViLT image processor pads a batch: sample b holds a
    ``valid_hw[b] = (h, w)`` pixel image in the top-left corner, zeros elsewhere, and ``pixel_mask`` marks the
    image (ref: vault/models/vault/dataset.py:323-347 calls that processor per item)."""
    B = len(valid_hw)
    out = synthetic_batch(spec, B, seed=seed, text_len=text_len, n_classes=n_classes)
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    Hp, Wp = pad_hw
    pix = np.zeros((B, spec.vilt.num_channels, Hp, Wp), np.float32)
    pm = np.zeros((B, Hp, Wp), np.int64)
    for b, (h, w) in enumerate(valid_hw):
        pix[b, :, :h, :w] = np.clip(rng.standard_normal((spec.vilt.num_channels, h, w), dtype=np.float32), -1.0, 1.0)
        pm[b, :h, :w] = 1
    out["pixel_values"], out["pixel_mask"] = pix, pm
    return out


def select_patches(pixel_mask: np.ndarray, patch_size: int, max_image_length: int = -1):
    """Host-side patch bookkeeping of ``ViltEmbeddings.visual_embed`` (HF:models/vilt/modeling_vilt.py:92-160)
    with a DETERMINISTIC choice where the reference draws at random (``torch.multinomial``): valid patches in
    row-major order, then - for images with fewer valid patches than the longest one of the batch - masked
    padding patches taken cyclically from the image's non-valid ones.  Every random outcome of the reference is
    a permutation of the valid part plus arbitrary masked padding, which valid tokens never attend to.

    pixel_mask [B, Hp, Wp] (0/1) ->
      sel   int32 [B, L]   patch slot (row * gw + col on the gh x gw patch grid of the padded canvas)
      valid int32 [B, L]   1 = real patch, 0 = padding (masked key)
      hw    int32 [B, 2]   valid patch rows / cols of each image (its position table is resized to h x w)
      (gh, gw), L
    """
    pm = np.asarray(pixel_mask)
    B, Hp, Wp = pm.shape
    if patch_size == 1:
        # already on the patch grid (the engine subsamples on the device: pixel_mask[:, ::ps, ::ps] is exactly what
        # nearest-neighbour interpolation to the grid reads)
        gh, gw = Hp, Wp
        xm = (pm != 0).astype(np.int64)
    else:
        gh, gw = Hp // patch_size, Wp // patch_size
        # nn.functional.interpolate(mode="nearest") to (gh, gw): source index floor(dst * Hp / gh)
        ri = (np.arange(gh) * Hp) // gh
        ci = (np.arange(gw) * Wp) // gw
        xm = (pm[:, ri][:, :, ci] != 0).astype(np.int64)          # [B, gh, gw]
    x_h = xm.sum(axis=1)[:, 0]
    x_w = xm.sum(axis=2)[:, 0]
    eff = x_h * x_w
    L = int(eff.max())
    if isinstance(max_image_length, int) and max_image_length > 0:
        L = min(L, int(max_image_length))
    sel = np.zeros((B, L), np.int32)
    valid = np.zeros((B, L), np.int32)
    flat = xm.reshape(B, -1)
    for b in range(B):
        v = np.flatnonzero(flat[b] != 0)
        nv = np.flatnonzero(flat[b] == 0)
        if len(v) >= L:
            sel[b] = v[:L]
            valid[b] = 1
        else:
            pad = L - len(v)
            if len(nv) == 0:
                raise ValueError("pixel_mask: an image with fewer valid patches than the batch maximum has no padding patch")
            sel[b, :len(v)] = v
            sel[b, len(v):] = nv[np.arange(pad) % len(nv)]
            valid[b, :len(v)] = 1
    hw = np.stack([x_h, x_w], axis=1).astype(np.int32)
    return sel, valid, hw, (gh, gw), L
