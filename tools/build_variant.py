"""Development: a second build of libvault_hip.so with extra -D flags on some sources (same-box A/B through VAULT_HIP_LIB):
   python tools/build_variant.py NAME "-DATTN_ABLATE=1" attention.hip [more.hip ...]  ->  vault_amd/libvault_hip.NAME.so"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vault_amd import build as B

name, flags, srcs = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
B.build()
objs = []
for f in B._sources():
    if f in srcs:
        o = os.path.join(B.OBJ, f[:-4] + f".{name}.o")
        subprocess.run([B.HIPCC, *B.FLAGS, *flags, "-c", os.path.join(B.CSRC, f), "-o", o], check=True)
    else:
        o = os.path.join(B.OBJ, f[:-4] + ".o")
    objs.append(o)
out = os.path.join(B.HERE, f"libvault_hip.{name}.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
print("built", out)
