"""Lint of the ring GEMM's ISA: a register written by an LDS read must not be touched before an lgkmcnt wait.

The fragment reads of gemm256.hip are issued through inline asm, so hipcc's wait-count insertion does not know that
their destination registers are written asynchronously (it has been seen to reuse the destination of a never
consumed read for a staging address while the LDS data was still on its way).  This script builds the control-flow
graph of every gemm256 kernel from the compiler's assembly, propagates the set of registers with an LDS read
pending (forward, union over predecessors) and reports every instruction that reads or writes such a register.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ivault_amd/csrc --cuda-device-only -S \
        vault_amd/csrc/gemm256.hip -o /tmp/g.s && python tools/lint_async_lds.py /tmp/g.s
"""
import re, sys

def regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return frozenset(out)

def lint_kernel(name, ins, labels, path):
    """ins: list of (line, op, rest); labels: name -> index"""
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (ln, op, rest) in enumerate(ins):
        tgt = rest.strip()
        if op == "s_endpgm": continue
        if op == "s_branch" or op.startswith("s_cbranch"):
            j = i + 1 + int(tgt) if re.fullmatch(r"-?\d+", tgt) else labels.get(tgt)   # numeric: dwords, all skipped ops here are one dword
            if j is not None and j < n: succ[i].append(j)
            if op == "s_branch": continue
        if i + 1 < n: succ[i].append(i + 1)
    state = [None] * n          # pending: dict reg -> line of the read
    state[0] = {}
    work, findings = [0], {}
    while work:
        i = work.pop()
        ln, op, rest = ins[i]
        pend = dict(state[i])
        if op == "s_waitcnt" and ("lgkmcnt" in rest or "vmcnt" not in rest):
            pend = {}           # any lgkm wait: hipcc's own counted waits belong to reads it knows about
        else:
            touched = regs(rest)
            if op.startswith("ds_read"):
                dst = regs(rest.split(",")[0])
                hit = (touched - dst) & pend.keys()
                if hit: findings[ln] = f"address of `{op} {rest}` has an LDS read pending (line {pend[min(hit)]})"
                for r in dst: pend[r] = ln
            else:
                hit = touched & pend.keys()
                if hit: findings[ln] = f"`{op} {rest}` touches v{sorted(hit)[:4]} with an LDS read pending (line {pend[min(hit)]})"
        for j in succ[i]:
            if state[j] is None: state[j] = dict(pend); work.append(j)
            else:
                new = [r for r in pend if r not in state[j]]
                if new:
                    for r in new: state[j][r] = pend[r]
                    work.append(j)
    for ln in sorted(findings): print(f"{path}:{ln}: {name[:70]}: {findings[ln]}")
    return len(findings)

def main(path):
    kernels, cur = [], None
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].strip()
        if not s: continue
        if s.endswith(":"):
            if "gemm256_kernel" in s and not s.startswith("."): cur = (s[:-1], [], {}); kernels.append(cur)
            elif cur is not None: cur[2][s[:-1]] = len(cur[1])
            continue
        if cur is None or s.startswith("."): continue
        op, _, rest = s.partition(" ")
        cur[1].append((ln, op, rest))
        if op == "s_endpgm": cur = None
    bad = sum(lint_kernel(k[0], k[1], k[2], path) for k in kernels)
    print(f"{len(kernels)} kernels, {bad} findings")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
