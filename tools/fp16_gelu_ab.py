"""Forward error of the fp16 operand build against the fp32 CPU oracle over many samples (run on the GPU box), to tell a
systematic change from rounding noise when the GELU epilogue's polynomial changes:
    python tools/fp16_gelu_ab.py                                  # the in-tree libvault_hip_f16.so
    VAULT_HIP_LIB_F16=build_ab/libvault_hip_f16_oldgelu.so python tools/fp16_gelu_ab.py
Prints max / rms of |dlogits| per sample at B = 48 (full size, seeds 77 and 78), train and eval mode, and the
reference-golden numbers bench.py reports."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vault_oracle as O                                   # noqa: E402  (a checker tool, not the product path)
from vault_amd.engine import VaultEngine                               # noqa: E402
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, build_state, synthetic_batch   # noqa: E402


def main():
    spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
    spec.lm.hidden_dropout_prob = 0.0
    spec.lm.attention_probs_dropout_prob = 0.0
    state = build_state(spec, 0)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    P = O.to_torch_state(state, requires_grad=False)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="fp16")
    print("library:", os.environ.get("VAULT_HIP_LIB_F16", "in-tree"))
    for seed in (77, 78):
        bn = synthetic_batch(spec, 48, seed=seed, n_classes=3)
        with torch.no_grad():
            loss, ref = O.vault_loss(P, spec, O.torch_batch(bn))
        db = {k: torch.from_numpy(v).to("cuda:0") for k, v in bn.items()}
        for train in (True, False):
            out = eng.forward(db, train=train, labels=db["labels"], need_hidden=False)
            torch.cuda.synchronize()
            d = (out["logits"].cpu() - ref["logits"]).abs()
            print(f"seed {seed} train={train}: |dlogits| max {d.max().item():.2e} rms {d.pow(2).mean().sqrt().item():.2e} "
                  f"|dloss| {abs(float(out['loss']) - float(loss)):.2e}")
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_bertweet_b2.npz"))
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=3)
    db = {k: torch.from_numpy(v).to("cuda:0") for k, v in bn.items()}
    for train in (True, False):
        out = eng.forward(db, train=train, labels=db["labels"], need_hidden=False)
        torch.cuda.synchronize()
        print(f"golden b2 train={train}: |dlogits| {np.abs(out['logits'].cpu().numpy() - g['logits']).max():.2e} "
              f"|dloss| {abs(float(out['loss']) - float(g['loss'])):.2e}")


if __name__ == "__main__":
    main()
