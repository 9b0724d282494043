"""Ring-kernel (0,1) data gradients with the A operand row-major against K-tile-major ([K / 64][rows][64]: a_hm) - development."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_BF16

M, H = 47360, 768
rb = lambda *s: torch.randn(*s, device="cuda").bfloat16()   # noqa: E731


def t(fn, n=30):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for K in (3072, 2304):
    A = [rb(M, K) for _ in range(2)]
    W = rb(K, H) * 0.05
    out = [torch.empty(M, H, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    for rep in range(2):
        a = t(lambda i: _gemm(A[i & 1], W, out[i & 1], M, H, K, K, H, H, 0, 1, EPI_BF16, cfg=-1))
        b = t(lambda i: _gemm(A[i & 1], W, out[i & 1], M, H, K, K, H, H, 0, 1, EPI_BF16, cfg=-1, a_hm=M))
        print(f"dgrad N = 768, K = {K}: A row-major {a:6.1f} us   K-tile-major {b:6.1f} us")
