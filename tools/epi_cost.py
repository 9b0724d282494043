"""Epilogue cost breakdown on the FFN-in shape (development tool): same GEMM, different epilogues."""
import sys, torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_BF16, EPI_GELU, EPI_DGELU, EPI_RES
B = 256; M = ((B * 185 + 255) // 256) * 256; H, FF = 768, 3072
def rb(*s): return torch.randn(*s, device="cuda").bfloat16()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
X = rb(M, H); W1 = rb(FF, H) * 0.05; bias = torch.randn(FF, device="cuda")
o = torch.empty(M, FF, dtype=torch.bfloat16, device="cuda"); o2 = torch.empty_like(o); aux = rb(M, FF)
o32 = torch.empty(M, FF, device="cuda"); res = torch.randn(M, FF, device="cuda")
for cfg in (2, 3):
    print(f"cfg{cfg}: bf16 {t(lambda: _gemm(X, W1, o, M, FF, H, H, H, FF, 0, 0, EPI_BF16, cfg=cfg, bias=bias)):.1f}us"
          f" | gelu {t(lambda: _gemm(X, W1, o, M, FF, H, H, H, FF, 0, 0, EPI_GELU, cfg=cfg, bias=bias)):.1f}us"
          f" | gelu+out2 {t(lambda: _gemm(X, W1, o, M, FF, H, H, H, FF, 0, 0, EPI_GELU, cfg=cfg, bias=bias, out2=o2)):.1f}us"
          f" | dgelu {t(lambda: _gemm(X, W1, o, M, FF, H, H, H, FF, 0, 0, EPI_DGELU, cfg=cfg, aux=aux)):.1f}us"
          f" | res(f32) {t(lambda: _gemm(X, W1, o32, M, FF, H, H, H, FF, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res)):.1f}us")
