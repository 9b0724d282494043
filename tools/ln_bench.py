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
