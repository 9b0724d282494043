"""Build a variant of the HIP library for same-box A/B runs: the in-tree objects, with the named sources recompiled under extra
compiler flags.      python tools/build_variant.py NAME [--fmt fp16] attention.hip:-DATTN_ST_NT=1 [more.hip:-DX=2,-DY=3]
Writes build_ab/libvault_hip_NAME.so (git-ignored; travels to the GPU box); select it with VAULT_HIP_LIB=/root/repo/build_ab/...
(VAULT_HIP_LIB_F16 for --fmt fp16)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vault_amd.build as b   # noqa: E402


def main():
    args = sys.argv[1:]
    name = args.pop(0)
    fmt = "bf16"
    if args and args[0] == "--fmt":
        fmt = args[1]
        args = args[2:]
    b.build(formats=(fmt,))
    _, objdir, extra = b.VARIANTS[fmt]
    over = {a.split(":", 1)[0]: a.split(":", 1)[1].split(",") for a in args}
    vdir = os.path.join(ROOT, "build_ab", "_obj_" + name)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for s in b._sources():
        if s in over:
            o = os.path.join(vdir, s[:-4] + ".o")
            cmd = [b.HIPCC, *b.FLAGS, *extra, *over[s], "-c", os.path.join(b.CSRC, s), "-o", o]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(r.stdout + r.stderr)
            objs.append(o)
        else:
            objs.append(os.path.join(objdir, s[:-4] + ".o"))
    out = os.path.join(ROOT, "build_ab", f"libvault_hip_{name}.so")
    subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", out, *objs], check=True)
    print("built", out)


if __name__ == "__main__":
    main()
