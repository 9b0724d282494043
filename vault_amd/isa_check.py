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


def sgpr_vmem_hazards(asm_text: str, kernel_substr: str = "") -> List[str]:
    """Findings 'kernel: writer -> vmem (n wait states)' in hipcc -S output (all kernels whose name contains the substring)."""
    out = []
    for m in re.finditer(r'^(_Z\S+|[A-Za-z_]\w*):.*?\n(.*?)\.Lfunc_end', asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if kernel_substr not in name:
            continue
        lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.'))]
        for i, l in enumerate(lines):
            if not l.startswith(_VMEM):
                continue
            regs = set()
            for a, b in re.findall(r's\[(\d+):(\d+)\]', l):
                regs |= set(range(int(a), int(b) + 1))
            if not regs:
                continue
            ws, j = 0, i - 1
            while j >= 0 and ws < _WAIT_STATES:
                p = lines[j]
                mm = re.match(r'v_\S+\s+s(\d+|\[(\d+):(\d+)\])', p)
                if mm:
                    d = set(range(int(mm.group(2)), int(mm.group(3)) + 1)) if mm.group(2) else {int(mm.group(1))}
                    if d & regs:
                        out.append(f"{name}: {p}  ->  {l}   ({ws} wait states)")
                        break
                nop = re.match(r's_nop (\d+)', p)
                ws += (int(nop.group(1)) + 1) if nop else 1
                j -= 1
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
