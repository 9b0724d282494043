"""LayerNorm forward / backward kernel times at the ViLT and LM shapes (development tool): python tools/ln_bench.py [B]
Backward in the encoder's form: bf16 dy + fp32 residual-stream gradient in, fp32 + bf16 dx out, d gamma / d beta /
the next Linear's bias gradient accumulated."""
import sys, torch
sys.path.insert(0, ".")
from vault_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = 768
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for S in (185, 40):
    M = B * S
    x = torch.randn(M, H, device="cuda"); dy = (torch.randn(M, H, device="cuda") * 0.1).bfloat16()
    dres = torch.randn(M, H, device="cuda") * 0.1
    mean = x.mean(1).contiguous(); rstd = (1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-12)).contiguous()
    gamma = torch.ones(H, device="cuda"); dx = torch.empty(M, H, device="cuda"); dxb = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    dg, db, dbias = (torch.zeros(H, device="cuda") for _ in range(3))
    tb = t(lambda: ops.layernorm_bwd(x, mean, rstd, gamma, M, H, dy_bf16=dy, dres=dres, dx_f32=dx, dx_bf16=dxb, dgamma=dg,
                                     dbeta=db, dbias=dbias))
    byts = M * H * (4 + 2 + 4 + 4 + 2)
    print(f"S={S:4d} B={B}: ln_bwd {tb:7.1f} us  {byts / tb / 1e6:6.2f} TB/s   checksum {dx.abs().sum().item():.6e}")
