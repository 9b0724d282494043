"""Fine-tune step on padded batches of differently sized images at full model size (development tool):
python tools/ragged_bench.py [B]   - canvas 384 x 640 (12 x 20 patches), image sizes drawn per sample."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vault_amd.engine import VaultEngine
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, synthetic_ragged_batch
from vault_amd.train import TrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
eng = VaultEngine(spec, "cuda:0", seed=0, classifier_dropout=0.1, half="bf16")
step = TrainStep(eng, learning_rate=2e-5, total_steps=100)
rng = np.random.default_rng(0)
hw = [(384, int(rng.integers(12, 21)) * 32) if rng.random() < 0.7 else (int(rng.integers(8, 13)) * 32, 384) for _ in range(B)]
hw[0] = (384, 640)
bn = synthetic_ragged_batch(spec, hw, (384, 640), seed=5)
db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
lab = torch.from_numpy(bn["labels"]).cuda()
for _ in range(3): loss = step(db, lab)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): loss = step(db, lab)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
ws = [w for k, w in eng._ws.items() if len(k) == 7 and k[3:6] != (0, 0, 0)][0]
print(f"B={B} canvas 384x640 S={ws['S']} valid patches/sample mean {ws['n_valid'].mean():.1f}: {dt*1e3:.2f} ms/step, "
      f"{B/dt:.1f} samples/s, loss {float(loss):.4f} (per-step host patch selection + mask check included)")
