"""Development: N fused train steps at per-GPU batch 256 with engine class attributes set from the command line (to be run under
rocprofv3).  usage: steps_attr.py [fp8] ATTR=value ..."""
import sys
import torch
sys.path.insert(0, ".")
from vault_amd.engine import VaultEngine
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, synthetic_batch
from vault_amd.train import TrainStep
from bench import resident_inputs

args = sys.argv[1:]
fp8 = "fp8" in args
dev = torch.device("cuda:0")
spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
bn = synthetic_batch(spec, 256, seed=1234, n_classes=3)
eng = VaultEngine(spec, dev, seed=0, classifier_dropout=0.1, half="bf16", fp8_forward=fp8)
for a in args:
    if "=" in a:
        k, v = a.split("=")
        setattr(eng, k, eval(v))
st = TrainStep(eng, learning_rate=2e-5, warmup_ratio=0.1, total_steps=1000, assume_full_pixel_mask=True)
batch, _, labels = resident_inputs(eng, spec, bn, dev)
for _ in range(11):
    st(batch, labels)
torch.cuda.synchronize()
