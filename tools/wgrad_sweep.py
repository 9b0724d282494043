"""wgrad (dW[N][K] += dY[M][N]^T X[M][K]) config / split sweep (development tool): python tools/wgrad_sweep.py B SEQ"""
import sys, torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_ATOMIC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 40
M = ((B * SEQ + 255) // 256) * 256
def rb(*s): return torch.randn(*s, device="cuda").bfloat16()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for (N, K) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
    dY = rb(M, N); X = rb(M, K); dW = torch.zeros(N, K, device="cuda")
    row = []
    for cfg in (0, 1, 2, 3):
        best = None
        for splits in (1, 2, 3, 4, 5, 7, 8, 10, 12, 16, 20, 28, 40):
            try:
                us = t(lambda: _gemm(dY, X, dW, N, K, M, N, K, K, 1, 1, EPI_ATOMIC, cfg=cfg, splits=splits, accumulate=1))
            except Exception:
                continue
            if best is None or us < best[0]: best = (us, splits)
        row.append(f"cfg{cfg}: {best[0]:6.1f}us s{best[1]}" if best else f"cfg{cfg}: n/a")
    print(f"M={M} dW[{N}x{K}] ({2*M*N*K/1e9:.0f} GF): " + " | ".join(row))
