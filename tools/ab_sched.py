"""Same-process A/B of the GEMM scheduling mode (ops.GEMM_SCHED: 0 = static tile walks, 3 = what a data-parallel step with more
than one rank sets - dynamic hand-out in the ring kernel, one block per tile in the double-buffered one) in the fused train step
on ONE GPU with nothing else on it: what the scheduler alone costs (development).
   python tools/ab_sched.py [batch] [rounds]"""
import sys
import time
import torch
sys.path.insert(0, ".")
from vault_amd import ops
from vault_amd.engine import VaultEngine
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, synthetic_batch
from vault_amd.train import TrainStep
from bench import resident_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
bn = synthetic_batch(spec, B, seed=1234, n_classes=3)
runs = []
for sched in (0, 3):
    ops.GEMM_SCHED = sched
    eng = VaultEngine(spec, dev, seed=0, classifier_dropout=0.1, half="bf16")
    st = TrainStep(eng, learning_rate=2e-5, warmup_ratio=0.1, total_steps=1000, assume_full_pixel_mask=True)
    batch, _, labels = resident_inputs(eng, spec, bn, dev)
    for _ in range(5):
        st(batch, labels)          # (the tape records the mode of its first step)
    runs.append((sched, st, batch, labels))
torch.cuda.synchronize()
for r in range(rounds):
    for sched, st, batch, labels in runs:
        ops.GEMM_SCHED = sched     # (part of the tape key)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            st(batch, labels)
        torch.cuda.synchronize()
        print(f"GEMM_SCHED={sched}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
