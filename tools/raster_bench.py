"""GEMMs of the step under raster group widths gn (n-tiles per column group; 0 = the launcher's default) - development.
   python tools/raster_bench.py [M]"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, ".")
from vault_amd import lib as L
from tests.test_gpu_gemm import EPI_BF16, EPI_GELU, EPI_DGELU, EPI_RES

M = int(sys.argv[1]) if len(sys.argv) > 1 else 47360
H, FF = 768, 3072
rb = lambda *s: torch.randn(*s, device="cuda").bfloat16()   # noqa: E731
X = [rb(M, H) for _ in range(2)]; XF = [rb(M, FF) for _ in range(2)]; XQ = [rb(M, 3 * H) for _ in range(2)]
W1 = rb(FF, H) * 0.05; Wq = rb(3 * H, H) * 0.05; W2t = rb(FF, H) * 0.05; Wo = rb(H, H) * 0.05; W2 = rb(H, FF) * 0.05; WqT = rb(H, 3 * H) * 0.05
of = [torch.empty(M, FF, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
oh = [torch.empty(M, H, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
o32 = [torch.empty(M, H, device="cuda") for _ in range(2)]; res = [torch.randn(M, H, device="cuda") for _ in range(2)]
# 8-bit gelu' slots: 64 KiB per tile (8 waves x 8 row tiles x 64 lanes x 16 B), 192-wide tiles: FF / 192 of them per row panel
o8 = [torch.empty((M // 256) * (FF // 192) * 65536, dtype=torch.uint8, device="cuda") for _ in range(2)]
oq = [torch.empty(M, 3 * H, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
bf, bq, bh = torch.randn(FF, device="cuda"), torch.randn(3 * H, device="cuda"), torch.randn(H, device="cuda")
lib = L.load()


def run(A, B, out, N, K, epi, gn, cfg, bias=None, out2=None, aux=None, u8=0, res=None, b_mode=0, ldb=None):
    a = L.GemmArgs()
    a.A, a.B, a.out = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.out2 = out2.data_ptr() if out2 is not None else None
    a.aux = aux.data_ptr() if aux is not None else None
    a.bias = bias.data_ptr() if bias is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo = M, N, K, K, (K if ldb is None else ldb), N
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.gn, a.aux_u8 = 0, b_mode, epi, cfg, 1, gn, u8
    L.check(lib.vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vault_gemm")


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


CASES = {
    "8w FFN-in fwd (12 n-tiles)": lambda gn: (lambda i: run(X[i & 1], W1, of[i & 1], FF, H, EPI_GELU, gn, -1, bias=bf, out2=o8[i & 1], u8=1)),
    "8w QKV fwd (9)": lambda gn: (lambda i: run(X[i & 1], Wq, oq[i & 1], 3 * H, H, EPI_BF16, gn, -1, bias=bq)),
    "8w gelu' dgrad (12)": lambda gn: (lambda i: run(X[i & 1], W2t, of[i & 1], FF, H, EPI_DGELU, gn, -1, aux=o8[i & 1], u8=1)),
    "8w attention-out dgrad": lambda gn: (lambda i: run(X[i & 1], Wo, oh[i & 1], H, H, EPI_BF16, gn, -1)),
    "ring attention-out fwd (4)": lambda gn: (lambda i: run(X[i & 1], Wo, o32[i & 1], H, H, EPI_RES, gn, -1, bias=bh, res=res[i & 1])),
    "ring FFN-out fwd (4)": lambda gn: (lambda i: run(XF[i & 1], W2, o32[i & 1], H, FF, EPI_RES, gn, -1, bias=bh, res=res[i & 1])),
    "res8w attention-out fwd (cfg 6)": lambda gn: (lambda i: run(X[i & 1], Wo, o32[i & 1], H, H, EPI_RES, gn, 6, bias=bh, res=res[i & 1])),
    "res8w FFN-out fwd (cfg 6)": lambda gn: (lambda i: run(XF[i & 1], W2, o32[i & 1], H, FF, EPI_RES, gn, 6, bias=bh, res=res[i & 1])),
    "r8wT FFN-in dgrad on W^T (cfg 6)": lambda gn: (lambda i: run(XF[i & 1], W2, oh[i & 1], H, FF, EPI_BF16, gn, 6)),
    "r8wT FFN-in dgrad on W^T (cfg 5)": lambda gn: (lambda i: run(XF[i & 1], W2, oh[i & 1], H, FF, EPI_BF16, gn, 5)),
    "r8wT QKV dgrad on W^T (cfg 6)": lambda gn: (lambda i: run(XQ[i & 1], WqT, oh[i & 1], H, 3 * H, EPI_BF16, gn, 6)),
    "ring FFN-in dgrad (4)": lambda gn: (lambda i: run(XF[i & 1], W1, oh[i & 1], H, FF, EPI_BF16, gn, -1, b_mode=1, ldb=H)),
    "ring QKV dgrad (4)": lambda gn: (lambda i: run(XQ[i & 1], Wq, oh[i & 1], H, 3 * H, EPI_BF16, gn, -1, b_mode=1, ldb=H)),
}
t(CASES["8w FFN-in fwd (12 n-tiles)"](0), n=200)   # clocks
for name, mk in CASES.items():
    gns = (0, 1, 2, 3, 4, 6, 0, 2, 3, 6) if name.startswith("8w") else (0, 1, 2, 3, 0, 1, 2)
    if os.environ.get("RASTER_BENCH_GNS"):
        gns = tuple(int(x) for x in os.environ["RASTER_BENCH_GNS"].split(","))
        if not name.startswith(os.environ.get("RASTER_BENCH_KIND", "8w")):
            continue
    print(f"M {M} {name:28s}: " + "  ".join(f"gn{g} {t(mk(g)):6.1f}" for g in gns))
