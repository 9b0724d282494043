"""K sweep of the 8-wave register-direct GEMM (cfg 5 / 6) against the ring kernel (cfg 3 / 4), plain bf16 epilogue (development tool).
usage: python tools/pp_sweep.py [cfgs]   (VAULT_HIP_LIB selects the library build)"""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_BF16

CFGS = tuple(int(c) for c in sys.argv[1].split(",")) if len(sys.argv) > 1 else (3, 4, 5, 6)
M = 47360


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for N in (768, 2304):
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    bias = torch.randn(N, device="cuda")
    for K in (128, 768, 1536, 3072):
        X = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        row = []
        for cfg in CFGS:
            for persist in (0,):
                us = t(lambda: _gemm(X, W, out, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, persist=persist))
                row.append(f"cfg{cfg}{'s%x' % (persist >> 4) if persist else ''}:{us:7.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF")
        if K == 768:   # what the stores cost: the same launch with (nearly) every row masked
            for cfg in CFGS:
                us = t(lambda: _gemm(X, W, out, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=cfg, bias=bias, m_valid=256))
                row.append(f"cfg{cfg}nostore:{us:7.1f}us")
        print(f"N={N:5d} K={K:5d} ", "  ".join(row), flush=True)
