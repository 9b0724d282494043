import sys, torch, ctypes as C
sys.path.insert(0, ".")
from vault_amd import lib as L
M, N = 47360, 2304
def rb(*s): return torch.randn(*s, device="cuda").bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); bias = torch.randn(N, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for K in (64, 128, 256, 768, 1536, 3072):
    X = rb(M, K); W = rb(N, K) * 0.05
    def run(cfg, persist=0, mv=M):
        a = L.GemmArgs(); a.A, a.B, a.out, a.bias = X.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr()
        a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, K, K, N, mv
        a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.persist = 0, 0, 0, cfg, 1, persist
        L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
    print(f"K={K:5d}", " ".join(f"cfg{c}:{t(lambda: run(c)):7.1f}us" for c in (0, 2, 3)), "| persistent",
          " ".join(f"cfg{c}:{t(lambda: run(c, 1)):7.1f}us" for c in (0, 2)),
          "| no-store (m_valid=256)", " ".join(f"cfg{c}:{t(lambda: run(c, 0, 256)):7.1f}us" for c in (2, 3)),
          f"| cfg3 dynamic scheduler: {t(lambda: run(3, 1)):7.1f}us")
