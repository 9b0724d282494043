"""Drop-in for ``vault.models.vault.processor`` (ref: vault/models/vault/processor.py:6-18): a ViLT
processor whose tokenizer is replaced by the language model's.  Host-side only; delegates to
HuggingFace ``ViltProcessor`` / ``AutoTokenizer`` (pre-processing is outside the hot path).
"""
from __future__ import annotations

from typing import Optional

try:
    from transformers import AutoTokenizer, ViltProcessor
except Exception as e:  # pragma: no cover
    ViltProcessor = object
    AutoTokenizer = None
    _IMPORT_ERROR = e
else:
    _IMPORT_ERROR = None

DEFAULT_VILT = "dandelin/vilt-b32-mlm"


class VaultProcessor(ViltProcessor):
    """``VaultProcessor.from_pretrained(vilt_directory, bert_directory=None)``: image processor (and
    tokenizer) of the ViLT checkpoint; if ``bert_directory`` is given its tokenizer is used instead.
    As in the reference, a ViLT directory that holds no processor falls back to the stock
    ``dandelin/vilt-b32-mlm`` processor."""

    @classmethod
    def from_pretrained(cls, vilt_directory: str, bert_directory: Optional[str] = None, **kwargs):
        if _IMPORT_ERROR is not None:
            raise RuntimeError(f"transformers is required for VaultProcessor: {_IMPORT_ERROR}")
        try:
            proc = super().from_pretrained(vilt_directory, **kwargs)
        except Exception:
            proc = super().from_pretrained(DEFAULT_VILT, **kwargs)
        if bert_directory is not None:
            proc.tokenizer = AutoTokenizer.from_pretrained(bert_directory)
        return proc
