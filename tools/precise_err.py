import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from vault_amd.engine import VaultEngine
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, synthetic_batch
g = np.load("tests/golden/full_bertweet_b2.npz")
spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
spec.lm.hidden_dropout_prob = 0; spec.lm.attention_probs_dropout_prob = 0
bn = synthetic_batch(spec, 2, seed=13, n_classes=3)
eng = VaultEngine(spec, "cuda:0", seed=0, classifier_dropout=0.0, with_grads=False, half="bf16")
db = {k: torch.from_numpy(v).cuda() for k, v in bn.items()}
for pr in (False, True):
    out = eng.forward(db, train=False, labels=db["labels"], need_hidden=True, precise=pr)
    torch.cuda.synchronize()
    T = 40
    h = out["last_hidden_state"][:, :T + 1].cpu().numpy()
    print("precise" if pr else "bf16   ", "logits %.2e  pooled %.2e  hidden %.2e (max %.2f)  loss %.2e" % (
        np.abs(out["logits"].cpu().numpy() - g["logits"]).max(), np.abs(out["pooler_output"].cpu().numpy() - g["pooler_output"]).max(),
        np.abs(h - g["hidden_text_cls"]).max(), np.abs(g["hidden_text_cls"]).max(), abs(float(out["loss"]) - float(g["loss"]))))
    import time
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): eng.forward(db, train=False, need_hidden=False, precise=pr)
    torch.cuda.synchronize(); print("   fwd ms (B=2):", (time.time() - t0) / 5 * 1e3)
