"""Build-time check of hipcc's device ISA text for a hazard hipcc does not pad inside inline asm.

gfx9 needs 5 wait states between a VALU instruction that writes an SGPR (v_readlane_b32, v_readfirstlane_b32, a compare) and a
vector-memory instruction that reads that SGPR.  hipcc inserts them for its own instructions; the GEMM kernels issue their
global loads / stores / atomics / LDS-DMA through asm statements with SGPR base addresses, and when register pressure makes hipcc
park such a base in a VGPR lane, the v_readlane_b32 that brings it back can sit directly in front of the asm statement - the
access then runs on a stale base (round 5: garbage bias values in the first two of four back-to-back loads of one kernel
variant).  `vault_amd.build` runs this over every kernel file it compiles and fails the build on a finding."""
from __future__ import annotations

import re
from typing import List

_VMEM = ("global_", "buffer_", "flat_", "scratch_")
_WAIT_STATES = 5
# VALU instructions that write an SGPR: lane reads (first operand), VOP3 compares (first operand; v_cmpx too), carry-out /
# scale forms (second operand: v_add_co_u32 v1, s[4:5], ..; v_mad_u64_u32 v[0:1], s[2:3], ..; v_div_scale_f32 v0, s[2:3], ..)
_SGPR_WRITERS = re.compile(r'v_(readlane|readfirstlane|cmpx?_|\w+_co_|addc_|subb_|subbrev_|mad_u64_u32|mad_i64_i32|div_scale_)')
_NO_FALLTHROUGH = ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap")


def _sregs(operand: str):
    """SGPR numbers named by one operand text ('s12', 's[4:5]'), else the empty set."""
    m = re.fullmatch(r's(\d+)', operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r's\[(\d+):(\d+)\]', operand)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def _operands(line: str):
    parts = line.split(None, 1)
    return [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []


def _written_sgprs(line: str):
    """SGPRs a VALU instruction writes: any SGPR among the first TWO operands of the writer classes above (a source SGPR in
    second position - v_cmp_lt_u32_e64 s[4:5], s6, v1 - is taken for written too: conservative)."""
    if not _SGPR_WRITERS.match(line):
        return set()
    out = set()
    for o in _operands(line)[:2]:
        out |= _sregs(o)
    return out


def sgpr_vmem_hazards(asm_text: str, kernel_substr: str = "") -> List[str]:
    """Findings 'kernel: writer -> vmem (n wait states)' in hipcc -S output (all kernels whose name contains the substring).

    Every SGPR operand of a vector-memory instruction counts (the s[a:b] base pair, a single-register soffset of buffer_*); the
    backward scan follows the fall-through path AND every branch into a label it passes (a reload that reaches the access through
    a branch target), and does not fall through an unconditional branch."""
    out = []
    for m in re.finditer(r'^(_Z\S+|[A-Za-z_]\w*):.*?\n(.*?)\.Lfunc_end', asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if kernel_substr not in name:
            continue
        lines = []
        for l in body.split('\n'):
            l = l.split(';')[0].strip()
            if not l:
                continue
            if re.fullmatch(r'[.\w$]+:', l):
                lines.append(l)                      # a label (kept: branch target)
            elif not l.startswith('.'):
                lines.append(l)
        targets = {}
        for i, l in enumerate(lines):
            if l.startswith(("s_branch", "s_cbranch")):
                ops = _operands(l)
                if ops:
                    targets.setdefault(ops[-1] + ":", []).append(i)
        for i, l in enumerate(lines):
            if not l.startswith(_VMEM):
                continue
            regs = set()
            for o in _operands(l):
                regs |= _sregs(o.split()[0] if o else o)      # ('s[2:3] offset:16': the register part)
            if not regs:
                continue
            seen = set()
            stack = [(i - 1, 0)]
            hit = None
            while stack and hit is None:
                j, ws = stack.pop()
                while j >= 0 and ws < _WAIT_STATES:
                    if (j, ws) in seen:
                        break
                    seen.add((j, ws))
                    p = lines[j]
                    if p.endswith(':'):
                        for b in targets.get(p, ()):          # paths that arrive by a branch (the branch itself: one wait state)
                            stack.append((b - 1, ws + 1))
                        if j > 0 and lines[j - 1].startswith(_NO_FALLTHROUGH):
                            break
                        j -= 1
                        continue
                    if p.startswith('v_') and (_written_sgprs(p) & regs):
                        hit = (p, ws)
                        break
                    nop = re.match(r's_nop (\d+)', p)
                    ws += (int(nop.group(1)) + 1) if nop else 1
                    j -= 1
            if hit is not None:
                out.append(f"{name}: {hit[0]}  ->  {l}   ({hit[1]} wait states)")
    return out


def spilling_kernels(asm_text: str) -> List[str]:
    """'kernel: n spilled VGPRs' for every kernel of a hipcc -S file whose metadata reports scratch spills.  The GEMM kernels
    issue every global access through asm with hand-counted `s_waitcnt vmcnt(N)`: a spill puts compiler-issued scratch loads /
    stores (counted by the same counter) between them - the waits then over-wait at best - and costs what the register budget was
    chosen to avoid; the build lists them per GEMM kernel file (build.py NO_SPILL -> csrc/_obj*/<file>.spills.txt)."""
    out = []
    for m in re.finditer(r'\.name:\s+(\S+)\s*\n(?:.*\n)*?\s*\.vgpr_spill_count:\s+(\d+)', asm_text):
        if int(m.group(2)) > 0:
            out.append(f"{m.group(1)}: {m.group(2)} spilled VGPRs")
    return out
