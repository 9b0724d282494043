"""Same-process A/B of a VaultEngine class attribute in the fused train step (development).
   python tools/ab_attr.py ATTR v1,v2[,v3] [batch] [rounds] [fp8]     e.g.  HEAD_MAJOR_MIN_ROWS 16384,8192 256 3"""
import sys
import time
import torch
sys.path.insert(0, ".")
from vault_amd.engine import VaultEngine
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, synthetic_batch
from vault_amd.train import TrainStep
from bench import resident_inputs

attr, vals = sys.argv[1], [eval(v) for v in sys.argv[2].split(",")]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
fp8 = len(sys.argv) > 5 and sys.argv[5] == "fp8"
dev = torch.device("cuda:0")
spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
bn = synthetic_batch(spec, B, seed=1234, n_classes=3)
runs = []
for v in vals:
    eng = VaultEngine(spec, dev, seed=0, classifier_dropout=0.1, half="bf16", fp8_forward=fp8)
    setattr(eng, attr, v)
    st = TrainStep(eng, learning_rate=2e-5, warmup_ratio=0.1, total_steps=1000, assume_full_pixel_mask=True)
    batch, _, labels = resident_inputs(eng, spec, bn, dev)
    for _ in range(5):
        st(batch, labels)
    runs.append((v, st, batch, labels))
torch.cuda.synchronize()
for r in range(rounds):
    for v, st, batch, labels in runs:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            st(batch, labels)
        torch.cuda.synchronize()
        print(f"{attr}={v}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
