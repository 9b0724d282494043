"""``vault.models.vault`` -> ``vault_amd.models.vault`` (same export list as ref: vault/models/vault/__init__.py:6-22)."""
from vault_amd.models.vault import *  # noqa: F401,F403
from vault_amd.models.vault import __all__  # noqa: F401
