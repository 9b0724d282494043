"""``vault_amd.models.vault`` mirrors the export list of the reference's ``vault.models.vault``
(ref: vault/models/vault/__init__.py:6-22) for the hot path built here."""
from .model import (VaultForImageAndTextRetrieval, VaultForImagesAndTextClassification, VaultForMaskedLM,
                    VaultForQuestionAnswering, VaultForTMSC, VaultMixin, VaultModel)
from .processor import VaultProcessor

__all__ = ["VaultModel", "VaultForTMSC", "VaultForImageAndTextRetrieval", "VaultForImagesAndTextClassification",
           "VaultForMaskedLM", "VaultForQuestionAnswering", "VaultMixin", "VaultProcessor"]
