"""Kernel choices at the LM stack's GEMM shapes (M = 10240 at B = 256; also 2560 / 5120), forward-form 16-bit-output Linears:
explicit kernel configurations against the automatic one (development).   python tools/lm_shapes_bench.py [rows ...]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from vault_amd import lib as L

def gemm(A, B, out, M, N, K, epi, cfg, bias=None, aux=None, out2=None, b_mode=0, ldb=None, res=None):
    a = L.GemmArgs()
    a.A, a.B, a.out = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.aux = None if aux is None else aux.data_ptr()
    a.out2 = None if out2 is None else out2.data_ptr()
    a.res = None if res is None else res.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, K, (K if ldb is None else ldb), N, M
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits = 0, b_mode, epi, cfg, 1
    L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")

def timeit(fn, iters=30):
    try:
        fn()
    except RuntimeError:
        return float("nan")
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

H, FF = 768, 3072
for rows in [int(x) for x in sys.argv[1:]] or [10240, 5120, 2560]:
    X = torch.randn(rows, H, device="cuda").bfloat16(); XF = torch.randn(rows, FF, device="cuda").bfloat16()
    cases = [("QKV forward", X, (torch.randn(3 * H, H, device="cuda") * 0.05).bfloat16(), 3 * H, H, 0, {}),
             ("attention-out dgrad (W^T)", X, (torch.randn(H, H, device="cuda") * 0.05).bfloat16(), H, H, 0, {}),
             ("FFN-in forward + gelu'", X, (torch.randn(FF, H, device="cuda") * 0.05).bfloat16(), FF, H, 1, {"out2": True}),
             ("gelu'-product dgrad (W^T)", X, (torch.randn(FF, H, device="cuda") * 0.05).bfloat16(), FF, H, 2, {"aux": True})]
    for name, A, W, N, K, epi, kw in cases:
        out = torch.empty(rows, N, device="cuda", dtype=torch.bfloat16)
        extra = {}
        if kw.get("out2"):
            extra["out2"] = torch.empty_like(out)
        if kw.get("aux"):
            extra["aux"] = (torch.rand(rows, N, device="cuda")).bfloat16()
        bias = torch.randn(N, device="cuda") if epi != 2 else None
        t = {c: timeit(lambda: gemm(A, W, out, rows, N, K, epi, c, bias=bias, **extra)) for c in (-1, 5, 6, 3, 4, 0, 1)}
        print(f"rows {rows:6d} {name:28s} N = {N:4d}: " + "  ".join(f"{('auto' if c < 0 else 'cfg' + str(c))} {v:6.1f}" for c, v in t.items()), flush=True)
