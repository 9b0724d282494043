"""f32-residual Linears (attention-out, FFN-out) at batch B: the automatic kernel choice (ring kernel) against the 8-wave kernel's
register-direct residual epilogue with 256- / 192-wide tiles (cfg 5 / 6), rotating buffers (development tool)."""
import sys, torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_RES
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 185
M = ((B * SEQ + 255) // 256) * 256
H, FF = 768, 3072
NB = 3
def t(fn, n=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for name, N, K in (("attention-out", H, H), ("FFN-out", H, FF)):
    X = [(torch.randn(M, K, device="cuda")).bfloat16() for _ in range(NB)]
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); bias = torch.randn(N, device="cuda")
    res = [torch.randn(M, N, device="cuda") for _ in range(NB)]
    out = [torch.empty(M, N, device="cuda") for _ in range(NB)]
    row = []
    for cfg in (-1, 5, 6):
        try:
            us = t(lambda i: _gemm(X[i % NB], W, out[i % NB], M, N, K, K, K, N, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res[i % NB]))
            row.append(f"cfg {cfg:2d}: {us:6.1f} us")
        except Exception as ex:
            row.append(f"cfg {cfg:2d}: refused")
    print(f"{name:14s} M={M} N={N} K={K}: " + " | ".join(row))
