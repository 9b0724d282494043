"""ctypes loader for libvault_hip.so.  There is no fallback: a missing or stale library raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VAULT_HIP_LIB: development override (A/B of two builds of the SAME HIP library on one box); still no fallback
LIB_PATH = os.environ.get("VAULT_HIP_LIB") or os.path.join(_HERE, "libvault_hip.so")
# the same sources compiled for the IEEE fp16 operand type (csrc/common.h `h16`, build.py VARIANTS): same exported ABI
LIB_PATH_F16 = os.environ.get("VAULT_HIP_LIB_F16") or os.path.join(_HERE, "libvault_hip_f16.so")
ABI_VERSION = 12
FORMATS = ("bf16", "fp16")
_libs = {}


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
        ("aux_u8", C.c_int), ("out_hm", C.c_int), ("a_hm", C.c_int),
        ("out_q", C.c_void_p), ("out_scale", C.c_void_p),
        ("splitk_ws", C.c_void_p), ("splitk_bytes", C.c_longlong),
    ]


def load(fmt: str = "bf16") -> C.CDLL:
    """Load the HIP library of the 16-bit operand format ``fmt``; raise (never fall back) when it is absent or has the
    wrong ABI."""
    lib = _libs.get(fmt)
    if lib is not None:
        return lib
    if fmt not in FORMATS:
        raise ValueError(f"operand format must be one of {FORMATS}")
    # torch ships its own libamdhip64: it must be in the process before our library is, or the
    # kernels would launch through a second, uninitialised HIP runtime
    import torch  # noqa: F401
    path = LIB_PATH if fmt == "bf16" else LIB_PATH_F16
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -m vault_amd.build` "
            "(the VAuLT hot path has no CPU/PyTorch fallback)")
    lib = C.CDLL(path)           # (RTLD_LOCAL: the two builds export the same names)
    lib.vault_abi_version.restype = C.c_int
    v = lib.vault_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"{os.path.basename(path)} ABI {v} != expected {ABI_VERSION}: rebuild")
    lib.vault_operand_format.restype = C.c_int
    if lib.vault_operand_format() != FORMATS.index(fmt):
        raise RuntimeError(f"{os.path.basename(path)} was not built for {fmt} operands: rebuild")
    _libs[fmt] = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}" + (" (EINVAL: bad shape/alignment)" if code == 22 else ""))
