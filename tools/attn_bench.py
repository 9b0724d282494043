import sys, torch
sys.path.insert(0, ".")
from vault_amd import ops
B, S, heads = 256, 185, 12
H = heads * 64; M = B * S
qkv = torch.randn(M, 3 * H, device="cuda").bfloat16()
km = torch.ones(B, S, device="cuda"); km[:, 30:40] = 0
ctx = torch.zeros(M, H, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, heads, S, device="cuda")
dctx = torch.randn(M, H, device="cuda").bfloat16(); dqkv = torch.zeros(M, 3 * H, dtype=torch.bfloat16, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
print("attn fwd  S=185: %.1f us" % t(lambda: ops.attention_fwd(qkv, km, ctx, lse, B, S, H, heads)))
print("attn bwd  S=185: %.1f us" % t(lambda: ops.attention_bwd(qkv, km, ctx, lse, dctx, dqkv, B, S, H, heads)))
S2 = 40; M2 = B * S2
qkv2 = torch.randn(M2, 3 * H, device="cuda").bfloat16(); km2 = torch.ones(B, S2, device="cuda")
ctx2 = torch.zeros(M2, H, dtype=torch.bfloat16, device="cuda"); lse2 = torch.zeros(B, heads, S2, device="cuda")
dctx2 = torch.randn(M2, H, device="cuda").bfloat16(); dqkv2 = torch.zeros(M2, 3 * H, dtype=torch.bfloat16, device="cuda")
print("attn fwd  S=40 : %.1f us" % t(lambda: ops.attention_fwd(qkv2, km2, ctx2, lse2, B, S2, H, heads)))
print("attn bwd  S=40 : %.1f us" % t(lambda: ops.attention_bwd(qkv2, km2, ctx2, lse2, dctx2, dqkv2, B, S2, H, heads)))
