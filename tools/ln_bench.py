"""Micro-benchmark of the LayerNorm backward at the ViLT / LM row counts of batch B (development tool)."""
import sys
import torch
sys.path.insert(0, ".")
from vault_amd import ops

H = 768


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


for rows in [int(a) for a in sys.argv[1:]] or [2560, 10240, 11840, 47360]:
    x = torch.randn(rows, H, device="cuda"); dres = torch.randn(rows, H, device="cuda")
    dy = torch.randn(rows, H, device="cuda").bfloat16(); dx = torch.empty(rows, H, device="cuda")
    mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-12).rsqrt().contiguous()
    g = torch.randn(H, device="cuda"); dg = torch.zeros(H, device="cuda"); db = torch.zeros(H, device="cuda")
    t_full = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, g, rows, H, dy_bf16=dy, dres=dres, dx_f32=dx, dgamma=dg, dbeta=db))
    t_nog = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, g, rows, H, dy_bf16=dy, dres=dres, dx_f32=dx))
    byts = rows * H * 14
    dxb = torch.empty(rows, H, device="cuda", dtype=torch.bfloat16); dbias = torch.zeros(H, device="cuda")
    t_3 = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, g, rows, H, dy_bf16=dy, dres=dres, dx_f32=dx, dx_bf16=dxb, dgamma=dg, dbeta=db, dbias=dbias))
    t_3n = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, g, rows, H, dy_bf16=dy, dres=dres, dx_f32=dx, dx_bf16=dxb))
    print(f"rows {rows:6d}: with dgamma/dbeta {t_full:7.1f} us ({byts / t_full / 1e6:6.2f} TB/s)   without {t_nog:7.1f} us ({byts / t_nog / 1e6:6.2f} TB/s)"
          f"   + bf16 copy and its column sums {t_3:7.1f} us, without the three sums {t_3n:7.1f} us")

# the 16-bit gradient stream case of the pre-LN ViLT stack (dy, residual-gradient stream in and out all 16-bit): the general
# kernel (a build without the stream dispatch in norm.hip) against the straight-line kernel with row prefetch; correctness: both agree
for rows in [int(a) for a in sys.argv[1:]] or [11840, 47360]:
    x = torch.randn(rows, H, device="cuda"); dr = torch.randn(rows, H, device="cuda").bfloat16()
    dy = torch.randn(rows, H, device="cuda").bfloat16(); dxb = torch.empty(rows, H, device="cuda", dtype=torch.bfloat16)
    mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-12).rsqrt().contiguous()
    g = torch.randn(H, device="cuda"); dg = torch.zeros(H, device="cuda"); db = torch.zeros(H, device="cuda"); dbi = torch.zeros(H, device="cuda")
    t = timeit(lambda: ops.layernorm_bwd(x, mean, rstd, g, rows, H, dy_bf16=dy, dres_bf16=dr, dx_bf16=dxb, dgamma=dg, dbeta=db, dbias=dbi))
    byts = rows * H * 10
    # reference result in torch
    dg.zero_(); db.zero_(); dbi.zero_()
    ops.layernorm_bwd(x, mean, rstd, g, rows, H, dy_bf16=dy, dres_bf16=dr, dx_bf16=dxb, dgamma=dg, dbeta=db, dbias=dbi)
    xh = (x - mean[:, None]) * rstd[:, None]; gy = dy.float() * g
    want = rstd[:, None] * (gy - gy.mean(1, keepdim=True) - xh * (gy * xh).mean(1, keepdim=True)) + dr.float()
    err = float((dxb.float() - want).abs().max() / want.abs().max())
    eg = float((dg - (dy.float() * xh).sum(0)).abs().max() / (dy.float() * xh).sum(0).abs().max())
    eb = float((dbi - want.sum(0)).abs().max() / want.sum(0).abs().max())
    import os
    print(f"16-bit stream, rows {rows:6d} : {t:7.1f} us ({byts / t / 1e6:5.2f} TB/s); "
          f"max rel err dx {err:.1e} dgamma {eg:.1e} dbias {eb:.1e}")
