"""Build vault_amd/libvault_hip.so and libvault_hip_f16.so (gfx950): the same sources compiled for the two 16-bit
operand types (csrc/common.h `h16`: bf16, and IEEE fp16 with -DVAULT_F16); same exported C ABI in both.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is kept
in-tree (git-ignored) and travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import re
import subprocess
import sys

from .isa_check import sgpr_vmem_hazards, spilling_kernels

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvault_hip.so")
OBJ = os.path.join(CSRC, "_obj")
# (library, object directory, extra compiler flags) per operand format
VARIANTS = {"bf16": (OUT, OBJ, []),
            "fp16": (os.path.join(HERE, "libvault_hip_f16.so"), os.path.join(CSRC, "_obj_f16"), ["-DVAULT_F16"])}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# kernel files whose spilling kernels are listed after a build (asm global accesses with hand-counted vmcnt waits, register budgets
# chosen per kernel: a spill in a hot instantiation is a performance bug, and puts compiler-issued scratch operations between them)
NO_SPILL = ("gemm256.hip", "gemm256_dyn.hip", "gemm8w.hip")
# ... and the instantiations in which a spill FAILS the build (mangled-name patterns): every 8-wave kernel, and the ring kernel's
# forms the train step spends its time in - weight gradients <1,1,5,4>, data gradients <0,1,0,3|2>, f32-residual forwards
# <0,0,3,3|2> - as the static translation unit compiles them (Lb0E; the dynamic-scheduler twins of gemm256_dyn.hip, used by
# data-parallel steps, carry the scheduler's state across the main loop and are reported only)
NO_SPILL_KERNELS = (r"gemm8w_kernel", r"gemm256_kernelILi1ELi1ELi5ELi4ELb0E", r"gemm256_kernelILi0ELi1ELi0ELi[23]ELb0E",
                    r"gemm256_kernelILi0ELi0ELi3ELi[23]ELb0E")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
         "-Wno-unused-result"]


def hot_spills(spills):
    """The entries of a spilling_kernels() list that name a kernel of NO_SPILL_KERNELS: a build failure."""
    return [s_ for s_ in spills if any(re.search(pat, s_) for pat in NO_SPILL_KERNELS)]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path: str, extra=()) -> str:
    h = hashlib.sha1()
    # (a .hip file that compiles another one - gemm256_dyn.hip - hashes what it includes)
    included = set(re.findall(r'#include\s+"([\w.]+\.hip)"', open(path).read()))
    for f in sorted(os.listdir(CSRC)) + ["../../include/vault_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and (f.endswith(".h") or os.path.abspath(p) == os.path.abspath(path) or f in included):
            h.update(open(p, "rb").read())
    h.update(open(os.path.join(HERE, "isa_check.py"), "rb").read())
    h.update(" ".join(FLAGS + list(extra)).encode())
    return h.hexdigest()


def _compile(job) -> str:
    src, objdir, extra = job
    path = os.path.join(CSRC, src)
    obj = os.path.join(objdir, src[:-4] + ".o")
    stamp = obj + ".sha1"
    dg = _digest(path, extra)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
        return obj
    # -save-temps=obj: the device ISA text falls out beside the object - checked for the hazard hipcc does not pad inside
    # inline asm (isa_check.py), then the intermediate files are removed
    cmd = [HIPCC, *FLAGS, *extra, "-save-temps=obj", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    stem = src[:-4]
    isa = os.path.join(objdir, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")
    isa_text = open(isa).read()
    found = sgpr_vmem_hazards(isa_text)
    # (spills are reported per file - see spills.txt after a build - and refused in the hot instantiations: hot_spills())
    spills = spilling_kernels(isa_text) if src in NO_SPILL else []
    with open(os.path.join(objdir, stem + ".spills.txt"), "w") as f:
        f.write("\n".join(spills) + ("\n" if spills else ""))
    hot = hot_spills(spills)
    for f in os.listdir(objdir):
        if f.startswith(stem + "-h") or f.startswith(stem + ".hip-"):
            os.remove(os.path.join(objdir, f))
    if found:
        os.remove(obj)
        raise RuntimeError(f"{src}: VALU-writes-SGPR -> VMEM hazard in front of an asm statement (isa_check.py):\n" + "\n".join(found))
    if hot:
        os.remove(obj)
        raise RuntimeError(f"{src}: register spills in a hot GEMM kernel (build.py NO_SPILL_KERNELS):\n" + "\n".join(hot))
    open(stamp, "w").write(dg)
    return obj


def build(verbose: bool = False, formats=("bf16", "fp16")) -> str:
    srcs = _sources()
    jobs = []
    for fmt in formats:
        _, objdir, extra = VARIANTS[fmt]
        os.makedirs(objdir, exist_ok=True)
        jobs += [(s, objdir, extra) for s in srcs]
    with cf.ThreadPoolExecutor(max_workers=min(7, len(jobs))) as ex:
        objs = list(ex.map(_compile, jobs))
    for k, fmt in enumerate(formats):
        out = VARIANTS[fmt][0]
        mine = objs[k * len(srcs):(k + 1) * len(srcs)]
        newest = max(os.path.getmtime(o) for o in mine)
        if not os.path.exists(out) or os.path.getmtime(out) < newest:
            # -Bsymbolic: calls between the library's own translation units bind inside the library (both builds export
            # the same names and may be loaded into one process)
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", out, *mine]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print("built", out, f"{os.path.getsize(out) / 1e6:.1f} MB")
    return OUT


if __name__ == "__main__":
    build(verbose=True)
    sys.exit(0)
