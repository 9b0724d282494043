"""What the epilogue extras of the gelu'-product data gradient cost (development tool): the same GEMM with and without the
bias-gradient column sums, and as a plain bf16 epilogue."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_BF16, EPI_DGELU

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 185
CFG = int(sys.argv[3]) if len(sys.argv) > 3 else 5      # 5: 256-wide tiles, 6: 192-wide
M = ((B * SEQ + 255) // 256) * 256
H, FF = 768, 3072


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


X = torch.randn(M, H, device="cuda").bfloat16()
W2t = (torch.randn(FF, H, device="cuda") * 0.05).bfloat16()
o_f = torch.empty(M, FF, dtype=torch.bfloat16, device="cuda")
aux = torch.empty(M, FF, dtype=torch.bfloat16, device="cuda")
csum = torch.zeros(FF, device="cuda")
flops = 2 * M * FF * H
for name, fn in [
    ("plain bf16 epilogue", lambda: _gemm(X, W2t, o_f, M, FF, H, H, H, FF, 0, 0, EPI_BF16, cfg=CFG)),
    ("gelu' u8, no colsum", lambda: _gemm(X, W2t, o_f, M, FF, H, H, H, FF, 0, 0, EPI_DGELU, cfg=CFG, aux=aux, aux_u8=1)),
    ("gelu' u8 + colsum", lambda: _gemm(X, W2t, o_f, M, FF, H, H, H, FF, 0, 0, EPI_DGELU, cfg=CFG, aux=aux, colsum=csum, aux_u8=1)),
    ("gelu' bf16 + colsum", lambda: _gemm(X, W2t, o_f, M, FF, H, H, H, FF, 0, 0, EPI_DGELU, cfg=CFG, aux=aux, colsum=csum)),
]:
    t = timeit(fn)
    print(f"M={M} cfg{CFG} {name:24s} {t*1e6:8.1f} us {flops/t/1e12:7.1f} TF/s")
