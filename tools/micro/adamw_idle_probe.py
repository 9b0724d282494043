"""How much of the flat parameter buffer is idle for the optimizer (g = m = v = 0) after a few fused steps on one resident batch
(development; round 6).   python tools/micro/adamw_idle_probe.py [batch]"""
import sys
import torch
sys.path.insert(0, ".")
from vault_amd.engine import VaultEngine
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, synthetic_batch
from vault_amd.train import TrainStep
from bench import resident_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
bn = synthetic_batch(spec, B, seed=1234, n_classes=3)
eng = VaultEngine(spec, dev, seed=0, classifier_dropout=0.1, half="bf16")
st = TrainStep(eng, learning_rate=2e-5, warmup_ratio=0.1, total_steps=1000, assume_full_pixel_mask=True)
batch, _, labels = resident_inputs(eng, spec, bn, dev)
for _ in range(4):
    st(batch, labels)
torch.cuda.synchronize()
P = eng.params
o, shp = P.offsets["bert.embeddings.word_embeddings.weight"]
n = shp[0] * shp[1]
for name, t in (("g", P.g), ("m", P.m), ("v", P.v)):
    z = t[o:o + n].view(shp[0], shp[1])
    rows_nz = int((z.abs().amax(dim=1) != 0).sum())
    print(f"word embedding table {name}: {rows_nz} of {shp[0]} rows non-zero; whole buffer non-zero share {float((t[:P.n_train] != 0).float().mean()):.4f}")
ids = batch["input_ids"]
print("distinct token ids in the batch:", int(ids.unique().numel()))
