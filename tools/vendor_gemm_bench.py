"""What the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16, f32 accumulate) takes for the GEMM shapes of the ViLT layer
at B = 256 - plain GEMMs, no epilogue - beside this repository's kernels for the same shapes WITH their epilogues (development;
a yardstick for DESIGN 5.1, not part of the product path).   python tools/vendor_gemm_bench.py"""
import sys
import torch
sys.path.insert(0, ".")

def timeit(fn, iters=20):
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

M, H, FF = 47360, 768, 3072
rb = lambda *s: (torch.randn(*s, device="cuda") * 0.5).bfloat16()   # noqa: E731
cases = [("QKV forward        [M x 2304 x 768]", (M, 768), (768, 2304), False),
         ("FFN-in forward     [M x 3072 x 768]", (M, 768), (768, 3072), False),
         ("attention-out fwd  [M x 768 x 768]", (M, 768), (768, 768), False),
         ("FFN-out forward    [M x 768 x 3072]", (M, 3072), (3072, 768), False),
         ("FFN-in weight grad [3072 x 768 x M]", (M, 3072), (M, 768), True),
         ("QKV weight grad    [2304 x 768 x M]", (M, 2304), (M, 768), True)]
for name, sa, sb, tn in cases:
    # rotate three operand sets so that a launch does not find its operands in L2 / MALL from the previous one
    As = [rb(*sa) for _ in range(3)]; Bs = [rb(*sb) for _ in range(3)]
    k = [0]
    if tn:
        def run():
            k[0] += 1
            return As[k[0] % 3].t() @ Bs[k[0] % 3]
        fl = 2.0 * sa[1] * sb[1] * sa[0]
    else:
        def run():
            k[0] += 1
            return As[k[0] % 3] @ Bs[k[0] % 3]
        fl = 2.0 * sa[0] * sa[1] * sb[1]
    us = timeit(run)
    print(f"{name}: {us:7.1f} us  {fl / us / 1e6:7.1f} TF/s = {fl / us / 1e6 / 2500:.3f} of 2.5 PF", flush=True)
