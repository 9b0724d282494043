"""Development: where the 16-bit output of the MXFP8 GELU forms differs between calls (emit / no emit, 8-bit gelu' or not)."""
import sys, torch
sys.path.insert(0, ".")
from vault_amd import ops
from tests.test_gpu_mx8 import _quant_gpu, EPI_GELU
M, N, K = 1280, 3072, 768
g = torch.Generator().manual_seed(22)
x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).bfloat16().cuda()
w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
q_a, s_a = _quant_gpu(x); q_w, s_w = _quant_gpu(w)
bias_r = torch.randn(N, generator=g).cuda()
def run(u8, bias, cfg=5, m_valid=M):
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    out2 = torch.zeros(M * N, dtype=torch.uint8, device="cuda") if u8 else torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_mxfp8(q_a, s_a, q_w, s_w, out, M, N, K, N, EPI_GELU, bias=bias, out2=out2, m_valid=m_valid, cfg=cfg, aux_u8=u8)
    torch.cuda.synchronize()
    return out, out2
for bname, bias in (("rand", bias_r), ("zero", torch.zeros(N, device="cuda")), ("none", None)):
    ref, _ = run(False, bias, cfg=0)
    for u8 in (False, True):
        tot = 0
        for rep in range(6):
            o, u = run(u8, bias)
            d = (o != ref)
            tot += int(d.sum())
            if d.any() and rep < 3:
                nz = d.nonzero()
                cols = sorted(set((nz[:, 1] % 64).tolist()))
                r0, c0 = nz[0].tolist()
                print("   bias", bname, "u8", u8, "rep", rep, "n", int(d.sum()), "cols%64", cols, "first", (r0, c0), float(o[r0, c0]), float(ref[r0, c0]),
                      "lanes(row%16)", sorted(set((nz[:, 0] % 16).tolist())))
        print("bias", bname, "u8", u8, "total differing over 6 runs", tot)
