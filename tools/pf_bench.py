"""Ring-kernel GEMMs of the step (development): data gradients, residual forwards and a grouped weight-gradient launch at the
bench shapes, rotating buffers.  Run under several builds with tools/ab_libs.sh.   python tools/pf_bench.py [M] [kinds]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from vault_amd import lib as L, ops
from tests.test_gpu_gemm import EPI_BF16, EPI_RES

M = int(sys.argv[1]) if len(sys.argv) > 1 else 47360
kinds = sys.argv[2] if len(sys.argv) > 2 else "dgrad,res,wgrad,lm"
H, FF = 768, 3072
rb = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()   # noqa: E731
lib = L.load()
NB = 3


def run(A, B, out, Mr, N, K, epi, cfg=-1, bias=None, res=None, b_mode=0, ldb=None):
    a = L.GemmArgs()
    a.A, a.B, a.out = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo = Mr, N, K, K, (K if ldb is None else ldb), N
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits = 0, b_mode, epi, cfg, 1
    L.check(lib.vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vault_gemm")


def t(fn, n=20):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def gemm_cases(Mr, tag):
    X = [rb(Mr, H) for _ in range(NB)]; XF = [rb(Mr, FF) for _ in range(NB)]; XQ = [rb(Mr, 3 * H) for _ in range(NB)]
    W1 = rb(FF, H) * 0.05; Wq = rb(3 * H, H) * 0.05; Wo = rb(H, H) * 0.05; W2 = rb(H, FF) * 0.05
    oh = [torch.empty(Mr, H, dtype=torch.bfloat16, device="cuda") for _ in range(NB)]
    o32 = [torch.empty(Mr, H, device="cuda") for _ in range(NB)]; res = [torch.randn(Mr, H, device="cuda") for _ in range(NB)]
    bh = torch.randn(H, device="cuda")
    out = []
    if "dgrad" in kinds or tag == "lm":
        out.append(("FFN-in dgrad K=3072", t(lambda i: run(XF[i % NB], W1, oh[i % NB], Mr, H, FF, EPI_BF16, b_mode=1, ldb=H)), 2.0 * Mr * H * FF))
        out.append(("QKV dgrad K=2304", t(lambda i: run(XQ[i % NB], Wq, oh[i % NB], Mr, H, 3 * H, EPI_BF16, b_mode=1, ldb=H)), 2.0 * Mr * H * 3 * H))
    if "res" in kinds or tag == "lm":
        out.append(("attention-out fwd K=768", t(lambda i: run(X[i % NB], Wo, o32[i % NB], Mr, H, H, EPI_RES, bias=bh, res=res[i % NB])), 2.0 * Mr * H * H))
        out.append(("FFN-out fwd K=3072", t(lambda i: run(XF[i % NB], W2, o32[i % NB], Mr, H, FF, EPI_RES, bias=bh, res=res[i % NB])), 2.0 * Mr * H * FF))
    for name, us, fl in out:
        print(f"{tag} M {Mr} ring {name:24s} {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s = {fl / us / 1e6 / 2500:.3f}", flush=True)


t(lambda i: run(rb(4096, H), rb(FF, H), torch.empty(4096, FF, dtype=torch.bfloat16, device="cuda"), 4096, FF, H, EPI_BF16), n=50)   # clocks
if "dgrad" in kinds or "res" in kinds:
    gemm_cases(M, "vilt")
if "lm" in kinds:
    gemm_cases(10240, "lm")
if "wgrad" in kinds:
    # one full launch of 256 items: FFN-out dW tiles of 8 layers (36 per layer), as the step's grouped launches
    G = 8
    for (No, Ki, nm) in ((H, FF, "FFN-out dW [768 x 3072]"), (FF, H, "FFN-in dW [3072 x 768]")):
        dY = (torch.randn(G, M, No, device="cuda") * 0.1).bfloat16(); X = (torch.randn(G, M, Ki, device="cuda") * 0.1).bfloat16()
        dW = torch.zeros(G, No, Ki, device="cuda")
        seg = [dict(dy=dY[0], x=X[0], dw=dW[0], n_out=No, n_in=Ki, batch=G, first=0, count=256, batch_dy=dY.stride(0),
                    batch_x=X.stride(0), batch_dw=dW.stride(0))]
        us = t(lambda i: ops.wgrad_grouped(seg, M, splits=1, accumulate=0), n=10)
        fl = 2.0 * M * 65536.0 * 256
        print(f"vilt M {M} grouped wgrad 256 items {nm:24s} {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s = {fl / us / 1e6 / 2500:.3f}", flush=True)
        del dY, X, dW
