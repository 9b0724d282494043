"""The fused AdamW pass over the flat parameter buffer of ViLT-B/32 + BERTweet (development tool)."""
import sys
import torch
sys.path.insert(0, ".")
from vault_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 246_000_000
fmt = sys.argv[2] if len(sys.argv) > 2 else "bf16"          # operand format of the 16-bit shadow (which library runs)
zero = (sys.argv[3] != "0") if len(sys.argv) > 3 else True   # zero the gradients in the pass
n -= n % 4
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1e-3
m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda"); pb = torch.zeros(n, device="cuda", dtype=ops.HALF_DTYPE[fmt])


def fn():
    with ops.operand_format(fmt):
        ops.adamw_step(p, g, m, v, pb, n, 2e-5, 0.9, 0.999, 1e-8, 0.01, grad_scale=1.0 if fmt == "bf16" else 1.0 / 4096.0,
                       zero_grad=zero)


for _ in range(3):
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    fn()
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / 10
print(f"{fmt} zero_grad={zero} n={n}: {t * 1e3:.1f} us, {34 * n / t / 1e9:.2f} TB/s of 34 B per parameter")

# round 6: elements that never received a gradient (g = m = v = 0, no weight decay) take no stores - the first `frac` of the
# buffer idle (an embedding table's untouched rows), the rest as above
for frac in (0.0, 0.22, 1.0):
    k = int(n * frac) // 4 * 4
    def fz():
        g.normal_(std=1e-3); g[:k] = 0
        m[:k] = 0; v[:k] = 0
    fz()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        g[:k] = 0            # (the pass zeroes g; idle elements must stay idle, the others get their gradient back below)
        g[k:].normal_(std=1e-3)
        torch.cuda.synchronize()
        s.record()
        with ops.operand_format(fmt):
            ops.adamw_step(p, g, m, v, pb, n, 2e-5, 0.9, 0.999, 1e-8, 0.0, grad_scale=1.0, zero_grad=zero)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    print(f"no weight decay, {frac:.0%} of the elements idle (g = m = v = 0): {min(ts):7.1f} us (min of 6), {sorted(ts)[3]:7.1f} (median)")
