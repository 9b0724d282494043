"""Split-K with the in-launch reduction (ring kernel, 192-wide tiles: gemm256.hip SK) against today's automatic choice and the
un-split ring kernel, at the N = 768 Linears with long contractions of small batches (development tool; model constants of
gemm.hip gemm_sk_splits come from this table).  usage: python tools/splitk_bench.py [rows ...]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from vault_amd import lib as L

def gemm(A, B, out, M, N, K, lda, ldb, b_mode, epi, cfg, splits=1, bias=None, res=None, ws=None):
    a = L.GemmArgs()
    a.A, a.B, a.out = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.bias = None if bias is None else bias.data_ptr()
    a.res = None if res is None else res.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, lda, ldb, N, M
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits = 0, b_mode, epi, cfg, splits
    if ws is not None:
        a.splitk_ws, a.splitk_bytes = ws.data_ptr(), ws.numel()
    L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")

def timeit(fn, iters=30):
    try:
        fn()
    except RuntimeError:
        return float("nan")      # (a combination the library refuses: workspace too small, ...)
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

ws = torch.zeros(16384 + 256 * 1024 * 1024, dtype=torch.uint8, device="cuda")
for rows in [int(x) for x in sys.argv[1:]] or [1280, 2560, 3072, 6144, 9216, 12032, 23808, 35584]:
    tiles = rows // 256 * 4
    for K, what in ((3072, "FFN-out forward"), ):
        A = torch.randn(rows, K, device="cuda").bfloat16(); W = (torch.randn(768, K, device="cuda") * 0.05).bfloat16()
        bias = torch.randn(768, device="cuda"); res = torch.randn(rows, 768, device="cuda"); o = torch.empty(rows, 768, device="cuda")
        t = {"auto": timeit(lambda: gemm(A, W, o, rows, 768, K, K, K, 0, 3, -1, 1, bias, res)),
             "ring": timeit(lambda: gemm(A, W, o, rows, 768, K, K, K, 0, 3, 4, 1, bias, res)),
             "auto+ws": timeit(lambda: gemm(A, W, o, rows, 768, K, K, K, 0, 3, -1, 1, bias, res, ws))}
        for sp in (2, 3, 4, 6):
            t[f"s{sp}"] = timeit(lambda: gemm(A, W, o, rows, 768, K, K, K, 0, 3, 4, sp, bias, res, ws))
        print(f"rows {rows:6d} ({tiles:3d} tiles) {what:16s} K = {K}: " + "  ".join(f"{k} {v:6.1f}" for k, v in t.items()))
    for K, what in ((3072, "FFN-in dgrad"), (2304, "QKV dgrad")):
        dY = torch.randn(rows, K, device="cuda").bfloat16(); W = (torch.randn(K, 768, device="cuda") * 0.05).bfloat16()
        o = torch.empty(rows, 768, device="cuda", dtype=torch.bfloat16)
        t = {"auto": timeit(lambda: gemm(dY, W, o, rows, 768, K, K, 768, 1, 0, -1)),
             "ring": timeit(lambda: gemm(dY, W, o, rows, 768, K, K, 768, 1, 0, 4)),
             "auto+ws": timeit(lambda: gemm(dY, W, o, rows, 768, K, K, 768, 1, 0, -1, 1, None, None, ws))}
        for sp in (2, 3, 4, 6):
            t[f"s{sp}"] = timeit(lambda: gemm(dY, W, o, rows, 768, K, K, 768, 1, 0, 4, sp, None, None, ws))
        print(f"rows {rows:6d} ({tiles:3d} tiles) {what:16s} K = {K}: " + "  ".join(f"{k} {v:6.1f}" for k, v in t.items()))

# the row-panel tail of the last round (gemm.hip vault_gemm_launch): automatic (cut) against the un-cut kernel, FFN-in shapes
for rows in (6144, 8960, 12032):
    A = torch.randn(rows, 768, device="cuda").bfloat16(); W = (torch.randn(3072, 768, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(3072, device="cuda"); o = torch.empty(rows, 3072, device="cuda", dtype=torch.bfloat16); o2 = torch.empty_like(o)
    def ffn(cfg):
        a = L.GemmArgs()
        a.A, a.B, a.out, a.out2, a.bias = A.data_ptr(), W.data_ptr(), o.data_ptr(), o2.data_ptr(), bias.data_ptr()
        a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = rows, 3072, 768, 768, 768, 3072, rows
        a.a_mode, a.b_mode, a.epi, a.cfg, a.splits = 0, 0, 1, cfg, 1
        L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
    print(f"rows {rows:6d} FFN-in forward + gelu' (8-wave, N = 3072, K = 768): auto {timeit(lambda: ffn(-1)):6.1f}  256-wide {timeit(lambda: ffn(5)):6.1f}  "
          f"192-wide {timeit(lambda: ffn(6)):6.1f}")
