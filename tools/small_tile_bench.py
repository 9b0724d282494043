"""128 x 128 against 64 x 128 tiles (both four-stage) for the launches that cover fewer than half of the CUs (development tool)."""
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_BF16, EPI_RES

H, FF = 768, 3072


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M in [int(a) for a in sys.argv[1:]] or [2560]:
    Xf = torch.randn(M, FF, device="cuda").bfloat16(); Xq = torch.randn(M, 3 * H, device="cuda").bfloat16()
    W1 = (torch.randn(FF, H, device="cuda") * 0.05).bfloat16(); Wq = (torch.randn(3 * H, H, device="cuda") * 0.05).bfloat16()
    W2 = (torch.randn(H, FF, device="cuda") * 0.05).bfloat16()
    res = torch.randn(M, H, device="cuda"); bias = torch.randn(H, device="cuda")
    Xh = torch.randn(M, H, device="cuda").bfloat16(); Wo = (torch.randn(H, H, device="cuda") * 0.05).bfloat16()
    for name, K, fn in [
        ("dgrad ffn1 (0,1) K=3072", FF, lambda cfg, o: _gemm(Xf, W1, o, M, H, FF, FF, H, H, 0, 1, EPI_BF16, cfg=cfg)),
        ("dgrad qkv  (0,1) K=2304", 3 * H, lambda cfg, o: _gemm(Xq, Wq, o, M, H, 3 * H, 3 * H, H, H, 0, 1, EPI_BF16, cfg=cfg)),
        ("fwd ffn2 (0,0) res K=3072", FF, lambda cfg, o: _gemm(Xf, W2, o, M, H, FF, FF, FF, H, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res)),
        ("fwd proj (0,0) res K=768", H, lambda cfg, o: _gemm(Xh, Wo, o, M, H, H, H, H, H, 0, 0, EPI_RES, cfg=cfg, bias=bias, res=res)),
        ("dgrad proj (0,1) K=768", H, lambda cfg, o: _gemm(Xh, Wo, o, M, H, H, H, H, H, 0, 1, EPI_BF16, cfg=cfg)),
        ("dgrad proj(T) (0,0) K=768", H, lambda cfg, o: _gemm(Xh, Wo, o, M, H, H, H, H, H, 0, 0, EPI_BF16, cfg=cfg)),
    ]:
        outs = {}
        line = f"M={M} {name:28s}"
        for cfg in (0, 7):
            o = torch.empty(M, H, dtype=torch.float32 if "res" in name else torch.bfloat16, device="cuda")
            t = timeit(lambda: fn(cfg, o))
            outs[cfg] = o.float().clone()
            line += f"  cfg{cfg} {t:7.1f} us ({2 * M * H * K / t / 1e6:6.0f} TF/s)"
        d = float((outs[0] - outs[7]).abs().max())
        print(line, f" max|diff| {d:.2e}")
        assert d <= 1e-2 * float(outs[0].abs().max())
