"""128 x 128 against 64 x 128 tiles (both four-stage) for the launches that cover fewer than half of the CUs (development tool).
--cold: every launch reads weights / activations that are not in the caches (24 rotating sets, > the 256 MB of MALL), as the
layers of a step do."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_BF16, EPI_RES

H, FF = 768, 3072
COLD = "--cold" in sys.argv
if COLD:
    sys.argv.remove("--cold")
R = 24 if COLD else 1


def timeit(fn, iters=48):
    for i in range(3):
        fn(i % R)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i % R)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def rb(*shape, scale=1.0):
    return [(torch.randn(*shape, device="cuda") * scale).bfloat16() for _ in range(R)]


for M in [int(a) for a in sys.argv[1:]] or [2560]:
    Xf, Xq, Xh = rb(M, FF), rb(M, 3 * H), rb(M, H)
    W1, Wq, W2, Wo = rb(FF, H, scale=0.05), rb(3 * H, H, scale=0.05), rb(H, FF, scale=0.05), rb(H, H, scale=0.05)
    res = [torch.randn(M, H, device="cuda") for _ in range(R)]; bias = torch.randn(H, device="cuda")
    for name, K, fn in [
        ("dgrad ffn1 (0,1) K=3072", FF, lambda cfg, o, i: _gemm(Xf[i], W1[i], o, M, H, FF, FF, H, H, 0, 1, EPI_BF16, cfg=cfg)),
        ("dgrad qkv  (0,1) K=2304", 3 * H, lambda cfg, o, i: _gemm(Xq[i], Wq[i], o, M, H, 3 * H, 3 * H, H, H, 0, 1, EPI_BF16, cfg=cfg)),
        ("fwd ffn2 (0,0) res K=3072", FF, lambda cfg, o, i: _gemm(Xf[i], W2[i], o, M, H, FF, FF, FF, H, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res[i])),
        ("fwd proj (0,0) res K=768", H, lambda cfg, o, i: _gemm(Xh[i], Wo[i], o, M, H, H, H, H, H, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res[i])),
        ("dgrad proj (0,1) K=768", H, lambda cfg, o, i: _gemm(Xh[i], Wo[i], o, M, H, H, H, H, H, 0, 1, EPI_BF16, cfg=cfg)),
        ("dgrad proj(T) (0,0) K=768", H, lambda cfg, o, i: _gemm(Xh[i], Wo[i], o, M, H, H, H, H, H, 0, 0, EPI_BF16, cfg=cfg)),
    ]:
        outs = {}
        line = f"M={M} {name:28s}"
        for cfg in (0, 7):
            o = torch.empty(M, H, dtype=torch.float32 if "res" in name else torch.bfloat16, device="cuda")
            t = timeit(lambda i: fn(cfg, o, i))
            fn(cfg, o, 0)
            outs[cfg] = o.float().clone()
            line += f"  cfg{cfg} {t:7.1f} us ({2 * M * H * K / t / 1e6:6.0f} TF/s)"
        d = float((outs[0] - outs[7]).abs().max())
        print(line, f" max|diff| {d:.2e}")
        assert d <= 1e-2 * float(outs[0].abs().max())
