"""ctypes loader for libvault_hip.so.  There is no fallback: a missing or stale library raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VAULT_HIP_LIB: development override (A/B of two builds of the SAME HIP library on one box); still no fallback
LIB_PATH = os.environ.get("VAULT_HIP_LIB") or os.path.join(_HERE, "libvault_hip.so")
ABI_VERSION = 7
_lib = None


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("aux", C.c_void_p), ("addtab", C.c_void_p),
        ("colsum", C.c_void_p), ("split3", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int),
        ("ldo", C.c_int), ("m_valid", C.c_int),
        ("a_mode", C.c_int), ("b_mode", C.c_int), ("epi", C.c_int), ("cfg", C.c_int),
        ("splits", C.c_int), ("accumulate", C.c_int),
        ("rpg", C.c_int), ("gstride", C.c_int), ("goff", C.c_int),
        ("drop_thresh", C.c_uint32), ("drop_seed", C.c_uint32), ("drop_stream", C.c_uint32),
        ("drop_scale", C.c_float), ("gn", C.c_int), ("persist", C.c_int),
        ("batch", C.c_int), ("batch_a", C.c_longlong), ("batch_b", C.c_longlong), ("batch_o", C.c_longlong),
        ("aux_u8", C.c_int),
    ]


def load() -> C.CDLL:
    """Load the HIP library; raise (never fall back) when it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64: it must be in the process before our library is, or the
    # kernels would launch through a second, uninitialised HIP runtime
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m vault_amd.build` "
            "(the VAuLT hot path has no CPU/PyTorch fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.vault_abi_version.restype = C.c_int
    v = lib.vault_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libvault_hip.so ABI {v} != expected {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}" + (" (EINVAL: bad shape/alignment)" if code == 22 else ""))
