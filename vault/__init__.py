"""Import alias: ``from vault.models.vault import VaultModel, VaultForTMSC, VaultProcessor`` resolves to the MI355X-native
implementation (``vault_amd.models.vault``), so code written against the reference package (ref: vault/models/vault/
__init__.py:6-22, README.md:34-58, experiments/clsf_vault.py:95-97,196-203) runs unchanged.  Only the hot-path
subpackage exists here; the reference's datasets / trainers / logging are out of scope."""
