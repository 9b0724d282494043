"""The LM embedding backward's scatter (word / position / token-type tables) at the step's row counts (development tool)."""
import sys
import torch
sys.path.insert(0, ".")
from vault_amd import ops

H = 768
for rows in [int(a) for a in sys.argv[1:]] or [2560, 10240]:
    d = torch.randn(rows, H, device="cuda")
    ids = torch.randint(0, 64001, (rows,), device="cuda")
    pos = (torch.arange(rows, device="cuda") % 40 + 2).int()
    tt = torch.zeros(rows, dtype=torch.int64, device="cuda")
    word = torch.zeros(64001, H, device="cuda"); ptab = torch.zeros(130, H, device="cuda"); ttab = torch.zeros(1, H, device="cuda")
    mask = torch.ones(rows, device="cuda")

    def fn():
        ops.scatter_add(d, [(word, ids), (ptab, pos), (ttab, tt)], rows, H, rowmask=mask)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    print(f"rows={rows}: {s.elapsed_time(e) / 20 * 1e3:.1f} us")
