"""Attention forward / backward kernel times at the ViLT and LM shapes (development tool): python tools/attn_bench.py [B]"""
import sys, torch
sys.path.insert(0, ".")
from vault_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SEQS = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (185, 40)    # 281: padded batches (two-phase backward)
H, heads = 768, 12
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for S in SEQS:
    M = ((B * S + 255) // 256) * 256
    qkv = (torch.randn(M, 3 * H, device="cuda") * 0.5).bfloat16()
    km = torch.ones(B, S, device="cuda")
    ctx = torch.zeros(M, H, dtype=torch.bfloat16, device="cuda"); dctx = (torch.randn(M, H, device="cuda") * 0.1).bfloat16()
    lse = torch.zeros(B, heads, S, device="cuda"); dqkv = torch.zeros(M, 3 * H, dtype=torch.bfloat16, device="cuda")
    tf = t(lambda: ops.attention_fwd(qkv, km, ctx, lse, B, S, H, heads))
    tb = t(lambda: ops.attention_bwd(qkv, km, ctx, lse, dctx, dqkv, B, S, H, heads))
    print(f"S={S:4d} B={B}: fwd {tf:7.1f} us   bwd {tb:7.1f} us   checksum {dqkv.float().abs().sum().item():.6e}")
    # the same in the head-major layout of qkv / dqkv ([3][heads][M][64]: vault_attn_args.qkv_hm)
    qh = qkv.view(M, 3 * heads, 64).permute(1, 0, 2).contiguous()
    dqh = torch.zeros(3 * heads, M, 64, dtype=torch.bfloat16, device="cuda")
    tf = t(lambda: ops.attention_fwd(qh, km, ctx, lse, B, S, H, heads, qkv_hm=M))
    tb = t(lambda: ops.attention_bwd(qh, km, ctx, lse, dctx, dqh, B, S, H, heads, qkv_hm=M))
    print(f"S={S:4d} B={B}: fwd {tf:7.1f} us   bwd {tb:7.1f} us   checksum {dqh.float().abs().sum().item():.6e}   (head-major qkv / dqkv)")
