"""256 x 192 against 256 x 128 tiles of the ring GEMM at the N = 768 shapes of the LM stack (development tool).
usage: python tools/tile128_bench.py [rows ...]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from vault_amd import lib as L

def gemm(A, B, out, M, N, K, lda, ldb, b_mode, epi, cfg, bias=None, res=None):
    a = L.GemmArgs()
    a.A, a.B, a.out = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.res = None if res is None else res.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, lda, ldb, N, M
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits = 0, b_mode, epi, cfg, 1
    L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")

def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for rows in [int(x) for x in sys.argv[1:]] or [10240, 5120, 11776, 12032]:
    for K, what in ((768, "attention-out"), (3072, "FFN-out")):
        A = torch.randn(rows, K, device="cuda").bfloat16(); W = (torch.randn(768, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(768, device="cuda"); res = torch.randn(rows, 768, device="cuda"); o = torch.empty(rows, 768, device="cuda")
        t = {c: timeit(lambda: gemm(A, W, o, rows, 768, K, K, K, 0, 3, c, bias, res)) for c in (4, 8, -1)}
        print(f"rows {rows:6d} {what:14s} forward  (K = {K:4d}): 192-wide {t[4]:6.1f} us  128-wide {t[8]:6.1f} us  auto {t[-1]:6.1f} us")
    for K, what in ((3072, "FFN-in dgrad"), (2304, "QKV dgrad")):
        dY = torch.randn(rows, K, device="cuda").bfloat16(); W = (torch.randn(K, 768, device="cuda") * 0.05).bfloat16()
        o = torch.empty(rows, 768, device="cuda", dtype=torch.bfloat16)
        t = {c: timeit(lambda: gemm(dY, W, o, rows, 768, K, K, 768, 1, 0, c)) for c in (4, 8, -1)}
        print(f"rows {rows:6d} {what:14s} dgrad    (K = {K:4d}): 192-wide {t[4]:6.1f} us  128-wide {t[8]:6.1f} us  auto {t[-1]:6.1f} us")
