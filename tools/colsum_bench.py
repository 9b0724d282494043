"""Column-sum kernel (bias gradients) at the shapes of a step (development tool): python tools/colsum_bench.py"""
import sys, torch
sys.path.insert(0, ".")
from vault_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for M in (47360, 11840, 10240, 2560):
    x = torch.randn(M, 2304, device="cuda").bfloat16()
    out = torch.zeros(2304, device="cuda")
    a = t(lambda: ops.colsum(x, 2304, M, 2304, out)); b = t(lambda: ops.colsum(x, 2304, M, 768, out))
    out.zero_(); ops.colsum(x, 2304, M, 2304, out); torch.cuda.synchronize()
    err = float((out - x.float().sum(0)).abs().max())
    print(f"M={M:6d}: N=2304 {a:6.1f} us   N=768 (ld 2304) {b:6.1f} us   max err {err:.2e}")
