"""GEMM under CU contention (development tool): a hog kernel holds N CUs on a second stream (as an RCCL collective
would in a data-parallel step) while the ring GEMM runs; static walk vs dynamic ticket scheduler.
Build first:  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libhog.so tools/hog.hip"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from vault_amd import lib as L
hog = C.CDLL("/tmp/libhog.so")
M, N, K = 47360, 2304, 768
X = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"); bias = torch.randn(N, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
def gemm(persist, cfg=3):
    a = L.GemmArgs(); a.A, a.B, a.out, a.bias = X.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr()
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo, a.m_valid = M, N, K, K, K, N, M
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.persist = 0, 0, 0, cfg, 1, persist
    L.check(L.load().vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "gemm")
def run(persist, hog_blocks, n=10, cfg=3):
    for _ in range(2): gemm(persist, cfg)
    torch.cuda.synchronize()
    if hog_blocks:
        hog.hog_launch(C.c_int(hog_blocks), C.c_longlong(int(0.02 * 1e8)), C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream))  # 20 ms
        torch.cuda._sleep(2_000_000)   # let the hog take its CUs first
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): gemm(persist, cfg)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for hb in (0, 8, 16, 32, 64):
    print(f"hog on {hb:3d} CUs: ring static {run(0, hb):7.1f} us   ring dynamic {run(1, hb):7.1f} us   8-wave (static) {run(0, hb, cfg=5):7.1f} us")
