"""Micro-benchmark of the GEMM shapes of one ViLT layer at batch B (development tool)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_BF16, EPI_GELU, EPI_DGELU, EPI_RES, EPI_ATOMIC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 185      # 185: ViLT fused sequence, 40: LM tokens
CFGS = tuple(int(c) for c in sys.argv[3].split(",")) if len(sys.argv) > 3 else (2, 3, 4)
M = ((B * SEQ + 255) // 256) * 256
H, FF = 768, 3072


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rb(*s):
    return torch.randn(*s, device="cuda").bfloat16()


X = rb(M, H); Xf = rb(M, FF)
Wqkv = rb(3 * H, H) * 0.05; Wo = rb(H, H) * 0.05; W1 = rb(FF, H) * 0.05; W2 = rb(H, FF) * 0.05
bias_q = torch.randn(3 * H, device="cuda"); bias_h = torch.randn(H, device="cuda"); bias_f = torch.randn(FF, device="cuda")
o_qkv = torch.empty(M, 3 * H, dtype=torch.bfloat16, device="cuda")
o_h32 = torch.empty(M, H, device="cuda"); res = torch.randn(M, H, device="cuda")
o_f = torch.empty(M, FF, dtype=torch.bfloat16, device="cuda"); o_f2 = torch.empty(M, FF, dtype=torch.bfloat16, device="cuda")
o_h = torch.empty(M, H, dtype=torch.bfloat16, device="cuda")
dW = torch.zeros(FF, H, device="cuda")

o_dqkv = rb(M, 3 * H)
cases = []
for cfg in [c for c in CFGS if c not in (5, 6)]:
    cases += [
        (f"fwd qkv   cfg{cfg}", 2 * M * 3 * H * H, lambda cfg=cfg: _gemm(X, Wqkv, o_qkv, M, 3 * H, H, H, H, 3 * H, 0, 0, EPI_BF16, cfg=cfg, bias=bias_q)),
        (f"fwd proj  cfg{cfg}", 2 * M * H * H, lambda cfg=cfg: _gemm(X, Wo, o_h32, M, H, H, H, H, H, 0, 0, EPI_RES, cfg=cfg, bias=bias_h, res=res)),
        (f"fwd ffn1  cfg{cfg}", 2 * M * FF * H, lambda cfg=cfg: _gemm(X, W1, o_f, M, FF, H, H, H, FF, 0, 0, EPI_GELU, cfg=cfg, bias=bias_f, out2=o_f2)),
        (f"fwd ffn2  cfg{cfg}", 2 * M * FF * H, lambda cfg=cfg: _gemm(Xf, W2, o_h32, M, H, FF, FF, FF, H, 0, 0, EPI_RES, cfg=cfg, bias=bias_h, res=res)),
        (f"dgrad ffn2 cfg{cfg}", 2 * M * FF * H, lambda cfg=cfg: _gemm(X, W2, o_f, M, FF, H, H, FF, FF, 0, 1, EPI_DGELU, cfg=cfg, aux=o_f2)),
        (f"dgrad ffn1 cfg{cfg}", 2 * M * FF * H, lambda cfg=cfg: _gemm(Xf, W1, o_h, M, H, FF, FF, H, H, 0, 1, EPI_BF16, cfg=cfg)),
        (f"dgrad proj cfg{cfg}", 2 * M * H * H, lambda cfg=cfg: _gemm(X, Wo, o_h, M, H, H, H, H, H, 0, 1, EPI_BF16, cfg=cfg)),
        (f"dgrad qkv  cfg{cfg}", 2 * M * 3 * H * H, lambda cfg=cfg: _gemm(o_dqkv, Wqkv, o_h, M, H, 3 * H, 3 * H, H, H, 0, 1, EPI_BF16, cfg=cfg)),
    ]
# 8-wave register-direct kernel (cfg 5 / 6): forward-form operands only - the data gradients run on the transposed weight shadow
W2t = W2.t().contiguous(); W1t = W1.t().contiguous(); Wot = Wo.t().contiguous(); Wqkvt = Wqkv.t().contiguous()
csum_f = torch.zeros(FF, device="cuda")
for c8 in [c for c in CFGS if c in (5, 6)]:
    for persist in (0,):
        tag = f"cfg{c8}"
        cases += [
            (f"dgrad ffn2(T) {tag}", 2 * M * FF * H, lambda persist=persist, c8=c8: _gemm(X, W2t, o_f, M, FF, H, H, H, FF, 0, 0, EPI_DGELU, cfg=c8, aux=o_f2, colsum=csum_f, persist=persist)),
            (f"dgrad ffn1(T) {tag}", 2 * M * FF * H, lambda persist=persist, c8=c8: _gemm(Xf, W1t, o_h, M, H, FF, FF, FF, H, 0, 0, EPI_BF16, cfg=c8, persist=persist)),
            (f"dgrad proj(T) {tag}", 2 * M * H * H, lambda persist=persist, c8=c8: _gemm(X, Wot, o_h, M, H, H, H, H, H, 0, 0, EPI_BF16, cfg=c8, persist=persist)),
            (f"dgrad qkv(T)  {tag}", 2 * M * 3 * H * H, lambda persist=persist, c8=c8: _gemm(o_dqkv, Wqkvt, o_h, M, H, 3 * H, 3 * H, 3 * H, H, 0, 0, EPI_BF16, cfg=c8, persist=persist)),
            (f"fwd qkv   {tag}", 2 * M * 3 * H * H, lambda persist=persist, c8=c8: _gemm(X, Wqkv, o_qkv, M, 3 * H, H, H, H, 3 * H, 0, 0, EPI_BF16, cfg=c8, bias=bias_q, persist=persist)),
            (f"fwd proj  {tag}", 2 * M * H * H, lambda persist=persist, c8=c8: _gemm(X, Wo, o_h32, M, H, H, H, H, H, 0, 0, EPI_RES, cfg=c8, bias=bias_h, res=res, persist=persist)),
            (f"fwd ffn1  {tag}", 2 * M * FF * H, lambda persist=persist, c8=c8: _gemm(X, W1, o_f, M, FF, H, H, H, FF, 0, 0, EPI_GELU, cfg=c8, bias=bias_f, out2=o_f2, persist=persist)),
            (f"fwd ffn2  {tag}", 2 * M * FF * H, lambda persist=persist, c8=c8: _gemm(Xf, W2, o_h32, M, H, FF, FF, FF, H, 0, 0, EPI_RES, cfg=c8, bias=bias_h, res=res, persist=persist)),
            (f"fwd ffn1  {tag} u8", 2 * M * FF * H, lambda persist=persist, c8=c8: _gemm(X, W1, o_f, M, FF, H, H, H, FF, 0, 0, EPI_GELU, cfg=c8, bias=bias_f, out2=o_f2, persist=persist, aux_u8=1)),
            (f"dgrad ffn2(T) {tag} u8", 2 * M * FF * H, lambda persist=persist, c8=c8: _gemm(X, W2t, o_f, M, FF, H, H, H, FF, 0, 0, EPI_DGELU, cfg=c8, aux=o_f2, colsum=csum_f, persist=persist, aux_u8=1)),
        ]
for cfg in (3,):
    for splits in (4, 7):
        cases.append((f"wgrad ffn1 cfg{cfg} s{splits}", 2 * M * FF * H,
                      lambda cfg=cfg, splits=splits: _gemm(Xf, X, dW, FF, H, M, FF, H, H, 1, 1, EPI_ATOMIC, cfg=cfg, splits=splits, accumulate=1)))
        cases.append((f"wgrad proj cfg{cfg} s{splits*4}", 2 * M * H * H,
                      lambda cfg=cfg, splits=splits: _gemm(X, X, dW, H, H, M, H, H, H, 1, 1, EPI_ATOMIC, cfg=cfg, splits=splits * 4, accumulate=1)))
# split-K data gradients into an f32 accumulator (small M: the K = 3072 / 2304 loops of a single partial round of tiles)
o_acc = torch.zeros(M, H, device="cuda")
for splits in (2, 3, 4, 6, 8):
    cases.append((f"dgrad ffn1 atomic cfg0 s{splits}", 2 * M * FF * H,
                  lambda splits=splits: _gemm(Xf, W1, o_acc, M, H, FF, FF, H, H, 0, 1, EPI_ATOMIC, cfg=0, splits=splits, accumulate=1)))
    cases.append((f"dgrad qkv  atomic cfg0 s{splits}", 2 * M * 3 * H * H,
                  lambda splits=splits: _gemm(o_dqkv, Wqkv, o_acc, M, H, 3 * H, 3 * H, H, H, 0, 1, EPI_ATOMIC, cfg=0, splits=splits, accumulate=1)))
for name, flops, fn in cases:
    t = timeit(fn)
    print(f"{name:28s} {t*1e6:9.1f} us  {flops/t/1e12:8.1f} TF/s")
