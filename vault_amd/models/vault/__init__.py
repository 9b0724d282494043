"""``vault_amd.models.vault`` mirrors the export list of the reference's ``vault.models.vault``
(ref: vault/models/vault/__init__.py:6-22) for the hot path built here (the MLM head is not built: SURVEY 8 f-4)."""
from .model import (VaultForImageAndTextRetrieval, VaultForImagesAndTextClassification, VaultForQuestionAnswering,
                    VaultForTMSC, VaultMixin, VaultModel)
from .processor import VaultProcessor

__all__ = ["VaultModel", "VaultForTMSC", "VaultForImageAndTextRetrieval", "VaultForImagesAndTextClassification",
           "VaultForQuestionAnswering", "VaultMixin", "VaultProcessor"]
