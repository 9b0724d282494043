"""Does a CU-masked stream let an HBM-bound kernel (the optimizer) run beside the GEMMs of a small batch?  (development probe)
hipExtStreamCreateWithCUMask through ctypes, wrapped as a torch ExternalStream."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_BF16

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(8), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


M, N, K = 11776, 768, 3072          # 46 x 4 = 184 tiles of 256 x 192: one partial round
X = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
O = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
big = torch.zeros(256 * 1024 * 1024, device=dev)    # 1 GiB: one add_ = 2 GiB of traffic


def gemms(n=40):
    for _ in range(n):
        _gemm(X, W, O, M, N, K, K, K, N, 0, 0, EPI_BF16, cfg=4)


def mem(n=6):
    for _ in range(n):
        big.add_(1.0)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


gemms(5); mem(2)
tg, tm = timed(gemms), timed(mem)
print(f"alone: 40 GEMMs {tg:.2f} ms, 6 x 2 GiB memory passes {tm:.2f} ms ({6 * 2.147 / tm:.2f} TB/s)")
plain = torch.cuda.Stream(device=dev)
work = torch.cuda.Stream(device=dev)
for name, bits in [("all 256 CUs (plain side stream)", None)] + [(f"mask {n} CUs ({pat})", b) for n, pat, b in [
        (64, "low bits", (1 << 64) - 1), (64, "every 4th", int("1000" * 64, 2)), (96, "low bits", (1 << 96) - 1),
        (128, "every 2nd", int("10" * 128, 2)), (64, "high bits", ((1 << 64) - 1) << 192)]]:
    st = plain if bits is None else masked_stream(bits)
    with torch.cuda.stream(st):
        mem(2)
    tm_s = timed(lambda: (st.wait_stream(torch.cuda.current_stream()), [None for _ in [0] if not torch.cuda.stream(st).__enter__()], mem(), torch.cuda.set_stream(torch.cuda.default_stream(dev))))

    def both():
        with torch.cuda.stream(st):
            mem()
        with torch.cuda.stream(work):       # (a masked stream is a BLOCKING stream: it serialises with the NULL stream)
            gemms()
    tb = timed(both)
    print(f"{name}: memory passes alone on it {tm_s:.2f} ms; beside the GEMMs {tb:.2f} ms (serial sum {tg + tm_s:.2f}, ideal {max(tg, tm_s):.2f})")
