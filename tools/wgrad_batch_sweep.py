"""Batched weight-gradient launch (ring kernel, G layers per launch) over the split count (development tool):
python tools/wgrad_batch_sweep.py [B] [SEQ] [G]   - prints us per launch and the engine's cost-model pick."""
import sys, torch
sys.path.insert(0, ".")
from tests.test_gpu_gemm import _gemm, EPI_ATOMIC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 185
G = int(sys.argv[3]) if len(sys.argv) > 3 else 6
M = ((B * SEQ + 255) // 256) * 256
nk = M // 64
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for (N, K) in ((768, 3072), (3072, 768), (2304, 768), (768, 768)):
    dY = (torch.randn(G, M, N, device="cuda") * 0.1).bfloat16(); X = (torch.randn(G, M, K, device="cuda") * 0.1).bfloat16()
    dW = torch.zeros(G, N, K, device="cuda")
    tiles = (N // 256) * (K // 256) * G
    cost = lambda sp: -(-tiles * sp // 256) * (1.67 * -(-nk // sp) + 40.0)
    pick = min((sp for sp in range(1, 9) if nk // sp >= 2), key=cost)
    row = []
    t(lambda: _gemm(dY, X, dW, N, K, M, N, K, K, 1, 1, EPI_ATOMIC, cfg=3, splits=1, accumulate=1, batch=G,
                    batch_a=M * N, batch_b=M * K, batch_o=N * K), n=10)   # (the first timed launches of a process run ~25 % slow)
    for sp in range(1, 9):
        us = t(lambda: _gemm(dY, X, dW, N, K, M, N, K, K, 1, 1, EPI_ATOMIC, cfg=3, splits=sp, accumulate=1, batch=G,
                             batch_a=M * N, batch_b=M * K, batch_o=N * K))
        row.append(f"s{sp}:{us:7.0f}{'*' if sp == pick else ' '}")
    print(f"M={M} G={G} dW[{N}x{K}] {2*M*N*K*G/1e9:6.0f} GF  " + " ".join(row))
    del dY, X, dW
