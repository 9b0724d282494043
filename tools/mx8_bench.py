"""MXFP8 forward GEMM vs the bf16 GEMM on the ViLT layer shapes at batch B (development tool)."""
import sys, torch
sys.path.insert(0, ".")
from vault_amd import ops
from tests.test_gpu_gemm import _gemm, EPI_BF16, EPI_GELU, EPI_RES
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = ((B * 185 + 255) // 256) * 256
H, FF = 768, 3072
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
def rb(*s): return torch.randn(*s, device="cuda").bfloat16()
for name, N, K, epi in (("qkv", 3 * H, H, EPI_BF16), ("proj", H, H, EPI_RES), ("ffn1", FF, H, EPI_GELU), ("ffn2", H, FF, EPI_RES)):
    X = rb(M, K); W = rb(N, K) * 0.05; bias = torch.randn(N, device="cuda")
    f32 = epi == EPI_RES
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
    res = torch.randn(M, N, device="cuda") if f32 else None
    out2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda") if epi == EPI_GELU else None
    xq = torch.empty(M, K, dtype=torch.uint8, device="cuda"); xs = torch.empty(M, K // 32, dtype=torch.uint8, device="cuda")
    wq = torch.empty(N, K, dtype=torch.uint8, device="cuda"); wsc = torch.empty(N, K // 32, dtype=torch.uint8, device="cuda")
    ops.quant_mxfp8(W, N, K, K, wq, wsc)
    tq = t(lambda: ops.quant_mxfp8(X, M, K, K, xq, xs))
    t8s = t(lambda: ops.gemm_mxfp8(xq, xs, wq, wsc, out, M, N, K, N, epi, bias=bias, res=res, out2=out2, cfg=0))
    t8 = t(lambda: ops.gemm_mxfp8(xq, xs, wq, wsc, out, M, N, K, N, epi, bias=bias, res=res, out2=out2, cfg=-1))
    t16 = t(lambda: _gemm(X, W, out, M, N, K, K, K, N, 0, 0, epi, cfg=-1, bias=bias, res=res, out2=out2))
    fl = 2.0 * M * N * K
    print(f"{name:5s} M={M} N={N} K={K}: quantise A {tq:6.1f} us ({M*K*3/tq/1e6:5.2f} TB/s) | mxfp8 simple {t8s:6.1f} us | mxfp8 8-wave {t8:6.1f} us {fl/t8/1e6:7.1f} TF/s | bf16 GEMM {t16:6.1f} us {fl/t16/1e6:7.1f} TF/s")
