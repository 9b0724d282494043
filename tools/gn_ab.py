import sys, torch, ctypes as C
sys.path.insert(0, ".")
from vault_amd import lib as L
M, H, FF = 47360, 768, 3072
def rb(*s): return torch.randn(*s, device="cuda").bfloat16()
X = rb(M, H); W = rb(3 * H, H) * 0.05; W1 = rb(FF, H) * 0.05; out = torch.empty(M, FF, dtype=torch.bfloat16, device="cuda")
bias = torch.randn(FF, device="cuda")
def run(N, Wt, cfg, gn, epi=0):
    a = L.GemmArgs(); a.A, a.B, a.out, a.bias = X.data_ptr(), Wt.data_ptr(), out.data_ptr(), bias.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, H, H, H, N, M
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.gn = 0, 0, epi, cfg, 1, gn
    L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for rnd in range(3):
    for name, N, Wt, cfg in (("qkv cfg2", 2304, W, 2), ("ffn1 cfg2", 3072, W1, 2), ("qkv cfg3", 2304, W, 3)):
        r = [(gn, t(lambda: run(N, Wt, cfg, gn))) for gn in (0, 2, 3, 4, 6)]
        print(name, " ".join(f"gn{g}:{x:.0f}us" for g, x in r))
# long-K shapes: ffn2 fwd (cfg4, N=768, K=3072) and dgrad ffn2 (N=3072, K=768, B mode 1)
Xf = rb(M, FF); W2 = rb(H, FF) * 0.05; o32 = torch.empty(M, H, device="cuda"); res = torch.randn(M, H, device="cuda"); bh = torch.randn(H, device="cuda")
def run2(cfg, gn):
    a = L.GemmArgs(); a.A, a.B, a.out, a.bias, a.res = Xf.data_ptr(), W2.data_ptr(), o32.data_ptr(), bh.data_ptr(), res.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, H, FF, FF, FF, H, M
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.gn = 0, 0, 3, cfg, 1, gn
    L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
def run3(cfg, gn):   # dgrad ffn2: dU[M,FF] = dY[M,H] . W2[H,FF]
    a = L.GemmArgs(); a.A, a.B, a.out = X.data_ptr(), W2.data_ptr(), out.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, FF, H, H, FF, FF, M
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.gn = 0, 1, 0, cfg, 1, gn
    L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
for rnd in range(2):
    print("ffn2 fwd cfg4", " ".join(f"gn{g}:{t(lambda: run2(4, g)):.0f}us" for g in (0, 1, 2)))
    print("dgrad ffn2 cfg2", " ".join(f"gn{g}:{t(lambda: run3(2, g)):.0f}us" for g in (0, 2, 4, 6)))
    print("dgrad ffn2 cfg3", " ".join(f"gn{g}:{t(lambda: run3(3, g)):.0f}us" for g in (0, 2, 4, 6)))
