"""Build vault_amd/libvault_hip.so (gfx950) and, for tests, the C-ABI smoke objects.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is kept
in-tree (git-ignored) and travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvault_hip.so")
OBJ = os.path.join(CSRC, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
         "-Wno-unused-result"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path: str) -> str:
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)) + ["../../include/vault_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and (f.endswith(".h") or os.path.abspath(p) == os.path.abspath(path)):
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src[:-4] + ".o")
    stamp = obj + ".sha1"
    dg = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dg:
        return obj
    cmd = [HIPCC, *FLAGS, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    open(stamp, "w").write(dg)
    return obj


def build(verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", OUT, f"{os.path.getsize(OUT) / 1e6:.1f} MB")
    return OUT


if __name__ == "__main__":
    build(verbose=True)
    sys.exit(0)
