"""Development: the hottest basic block (most MFMAs) of a kernel in hipcc's -S output.  usage: isa_loop.py file.s NAME_SUBSTR [pattern ...]"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^(_Z\S*' + re.escape(name) + r'\S*):.*?\n(.*?)\.Lfunc_end', s, re.S | re.M)
body = m.group(2)
blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
best = None
for i in range(1, len(blocks), 2):
    n = blocks[i + 1].count('v_mfma')
    if best is None or n > best[0]:
        best = (n, blocks[i], blocks[i + 1])
b = best[2]
print(m.group(1)[:80], 'block', best[1], 'mfma', best[0],
      {p: len(re.findall(p, b)) for p in ('global_load_lds_dwordx4', r'global_load_lds_dword\s', 'v_readlane', 'v_writelane', 'scratch_',
                                           'v_mad_u64', 's_waitcnt vmcnt', 'buffer_', r'global_store', r'global_load_dword')})
bl = [l for l in b.split('\n') if 'ASM' not in l]
for pat in sys.argv[3:]:
    idx = [i for i, l in enumerate(bl) if re.search(pat, l)]
    for k in idx[:2]:
        st = max([j for j in range(k) if 'v_mfma' in bl[j]] or [0])
        en = min([j for j in range(k, len(bl)) if 'v_mfma' in bl[j]] or [len(bl) - 1])
        print('----', pat)
        print('\n'.join(bl[st:en + 1]))
