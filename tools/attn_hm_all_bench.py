"""Attention forward / backward with EVERY operand contiguous per (batch, head) item, emulated by calling the kernels with
heads = 1, H = 64 over B x 12 'samples' (same arithmetic per item, same item count): what head-major ctx / dctx would buy."""
import sys, torch
sys.path.insert(0, ".")
from vault_amd import ops
B, S, H, heads = 256, 185, 768, 12
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
M = ((B * S + 255) // 256) * 256
km = torch.ones(B, S, device="cuda")
ctx = torch.zeros(M, H, dtype=torch.bfloat16, device="cuda"); dctx = (torch.randn(M, H, device="cuda") * 0.1).bfloat16()
lse = torch.zeros(B, heads, S, device="cuda")
qh = (torch.randn(3 * heads, M, 64, device="cuda") * 0.5).bfloat16()
dqh = torch.zeros(3 * heads, M, 64, dtype=torch.bfloat16, device="cuda")
for rep in range(3):
    tf = t(lambda: ops.attention_fwd(qh, km, ctx, lse, B, S, H, heads, qkv_hm=M))
    tb = t(lambda: ops.attention_bwd(qh, km, ctx, lse, dctx, dqh, B, S, H, heads, qkv_hm=M))
    print(f"head-major qkv / dqkv, row-major ctx / dctx: fwd {tf:6.1f} us  bwd {tb:6.1f} us")
    B2 = B * heads
    M2 = ((B2 * S + 255) // 256) * 256
    km2 = torch.ones(B2, S, device="cuda")
    ctx2 = torch.zeros(M2, 64, dtype=torch.bfloat16, device="cuda"); dctx2 = (torch.randn(M2, 64, device="cuda") * 0.1).bfloat16()
    lse2 = torch.zeros(B2, 1, S, device="cuda")
    q2 = (torch.randn(3, M2, 64, device="cuda") * 0.5).bfloat16(); dq2 = torch.zeros(3, M2, 64, dtype=torch.bfloat16, device="cuda")
    tf = t(lambda: ops.attention_fwd(q2, km2, ctx2, lse2, B2, S, 64, 1, qkv_hm=M2))
    tb = t(lambda: ops.attention_bwd(q2, km2, ctx2, lse2, dctx2, dq2, B2, S, 64, 1, qkv_hm=M2))
    print(f"everything contiguous per item (heads = 1 emulation): fwd {tf:6.1f} us  bwd {tb:6.1f} us")
    del km2, ctx2, dctx2, lse2, q2, dq2
