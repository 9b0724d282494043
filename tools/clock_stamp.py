"""The clock the chip HOLDS inside the three long GEMM kernels of the step (development; DESIGN 5's ceiling table): a diagnostic
build (`build_variant.py stamp gemm256.hip:-DR256_STAMP=1 gemm8w.hip:-DW8_STAMP=1`) stamps s_memtime / s_memrealtime at the
start and the end of every block; clock = shader cycles / (real-time ticks x 10 ns).  Each kernel runs back to back on random
data for ~2 s before the stamps are read (MI355X_MICROARCH.md, DVFS give-back item 6).
    VAULT_HIP_LIB=build_ab/libvault_hip_stamp.so python tools/clock_stamp.py"""
import ctypes as C
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from vault_amd import lib as L, ops
from tests.test_gpu_gemm import EPI_BF16, EPI_GELU

M, H, FF = 47360, 768, 3072
lib = L.load()
rb = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()   # noqa: E731


def run(A, B, out, N, K, epi, b_mode=0, ldb=None, bias=None, out2=None, u8=0):
    a = L.GemmArgs()
    a.A, a.B, a.out = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.out2 = out2.data_ptr() if out2 is not None else None
    a.M, a.N, a.K, a.lda, a.ldb, a.ldo = M, N, K, K, (K if ldb is None else ldb), N
    a.a_mode, a.b_mode, a.epi, a.cfg, a.splits, a.aux_u8 = 0, b_mode, epi, -1, 1, u8
    L.check(lib.vault_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "vault_gemm")


def stamps(fn_name):
    buf = (C.c_ulonglong * 512)()
    rc = getattr(lib, fn_name)(buf)
    assert rc == 0, rc
    a = np.array(buf[:], dtype=np.float64).reshape(256, 2)
    a = a[a[:, 1] > 0]
    ghz = a[:, 0] / (a[:, 1] * 10.0)           # cycles per ns
    us = a[:, 1] * 0.01
    return float(np.median(ghz)), float(ghz.min()), float(ghz.max()), float(np.median(us))


def soak(fn, seconds=2.0):
    fn(); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize(); n += 10
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10 * 1e3


X = rb(M, H); XF = rb(M, FF); W1 = rb(FF, H) * 0.05
of = torch.empty(M, FF, dtype=torch.bfloat16, device="cuda"); oh = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
o8 = torch.empty((M // 256) * (FF // 192) * 65536, dtype=torch.uint8, device="cuda"); bf = torch.randn(FF, device="cuda")
G = 8
dY = (torch.randn(G, M, H, device="cuda") * 0.1).bfloat16(); Xs = (torch.randn(G, M, FF, device="cuda") * 0.1).bfloat16()
dW = torch.zeros(G, H, FF, device="cuda")
seg = [dict(dy=dY[0], x=Xs[0], dw=dW[0], n_out=H, n_in=FF, batch=G, first=0, count=256, batch_dy=dY.stride(0), batch_x=Xs.stride(0),
            batch_dw=dW.stride(0))]
cases = [("weight gradients, grouped launch of 256 items (gemm256<1,1,5,4>)", lambda: ops.wgrad_grouped(seg, M, splits=1, accumulate=0),
          "vault_debug_r256_stamps", 2.0 * M * 65536.0 * 256),
         ("FFN-in data gradient K = 3072 (gemm256<0,1,0,3>)", lambda: run(XF, W1, oh, H, FF, EPI_BF16, b_mode=1, ldb=H), "vault_debug_r256_stamps",
          2.0 * M * H * FF),
         ("FFN-in forward, GELU + 8-bit gelu' (gemm8w<7,4>)", lambda: run(X, W1, of, FF, H, EPI_GELU, bias=bf, out2=o8, u8=1), "vault_debug_w8_stamps",
          2.0 * M * H * FF)]
for name, fn, sym, fl in cases:
    us = soak(fn)
    med, lo, hi, blk_us = stamps(sym)
    print(f"{name}: {us:7.1f} us per launch = {fl / us / 1e6 / 2500:.3f} of 2.5 PF; in-kernel clock median {med:.3f} GHz (blocks {lo:.3f} .. {hi:.3f}), "
          f"median block lifetime {blk_us:.1f} us; at that clock the matrix peak is {2.5 * med / 2.4:.3f} PF -> {fl / us / 1e6 / (2500 * med / 2.4):.3f} of it", flush=True)
