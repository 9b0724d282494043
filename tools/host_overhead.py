"""Host enqueue time vs GPU time of one fine-tune step at a small batch (development tool)."""
import sys, time, torch
sys.path.insert(0, ".")
from vault_amd.engine import VaultEngine
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, synthetic_batch
from vault_amd.train import TrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
eng = VaultEngine(spec, "cuda:0", seed=0, classifier_dropout=0.1, half="bf16")
step = TrainStep(eng, total_steps=1000, assume_full_pixel_mask=True)
bn = synthetic_batch(spec, B, seed=1)
db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}; lab = torch.from_numpy(bn["labels"]).cuda()
for _ in range(5): step(db, lab)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step(db, lab)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3*(t1-t0)/n:.2f} ms/step, wall {1e3*(t2-t0)/n:.2f} ms/step, launches/step ~{len(step._tape.calls)}")
