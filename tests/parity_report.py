"""Test-side report (not collected by pytest; run by hand): HIP engine vs CPU oracle on the golden batches; prints error magnitudes."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import vault_oracle as O
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, build_state, synthetic_batch
from vault_amd.engine import VaultEngine

CASES = {
    "tiny_roberta": (lambda: VaultSpec.tiny(3, "roberta"), 3, 11),
    "tiny_bert": (lambda: VaultSpec.tiny(3, "bert"), 3, 12),
    "full": (lambda: VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3), 2, 13),
}
which = sys.argv[1:] or ["tiny_roberta", "tiny_bert"]
for name in which:
    mk, B, seed = CASES[name]
    spec = mk()
    if spec.lm is not None:
        spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, B, seed=seed, n_classes=3)
    state = build_state(spec, 0)
    t0 = time.time()
    P = O.to_torch_state(state, requires_grad=True)
    taps = {}
    batch = O.torch_batch(bn)
    out = O.vault_forward(P, spec, batch, taps=taps)
    loss = torch.nn.functional.cross_entropy(out["logits"], batch["labels"])
    loss.backward()
    print(name, "oracle time %.1fs" % (time.time() - t0))
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items()}
    res = eng.forward(db, train=True, labels=db["labels"], need_hidden=True)
    eng.zero_grad()
    eng.backward()
    torch.cuda.synchronize()
    T = bn["input_ids"].shape[1]
    def err(a, b): return float((a.detach().cpu().float() - b.detach().float()).abs().max())
    print("  logits err %.3e (|logits| %.3f)" % (err(res["logits"], out["logits"]), float(out["logits"].abs().max())))
    print("  pooled err %.3e" % err(res["pooler_output"], out["pooler_output"]))
    print("  hidden err %.3e (max %.2f)" % (err(res["last_hidden_state"], out["last_hidden_state"]), float(out["last_hidden_state"].abs().max())))
    print("  loss %.6f vs %.6f" % (float(res["loss"]), float(loss)))
    ws = eng.last
    S = ws["S"]
    # taps
    for k in sorted(taps, key=lambda s: (s.split("layer")[0], int(s.split("layer")[1]) if "layer" in s else -1)):
        t = taps[k]
        if k == "lm_embed": mine = ws["lm_y"][0][:B * T].view(B, T, -1)
        elif k.startswith("lm_layer"): mine = ws["lm_y"][int(k[8:]) + 1][:B * T].view(B, T, -1)
        elif k == "vilt_embed": mine = ws["x"][0][:B * S].view(B, S, -1)
        else: mine = ws["x"][int(k[10:]) + 1][:B * S].view(B, S, -1)
        print("   tap %-14s err %.3e  (max %.2f)" % (k, err(mine, t), float(t.abs().max())))
    worst = []
    tot_n = tot_d = 0.0
    for n in eng.params.trainable:
        g = eng.params.gr(n).detach().cpu().double()
        r = P[n].grad
        if r is None:
            print("   oracle has no grad for", n); continue
        r = r.double()
        d = float((g - r).norm()); rn = float(r.norm())
        tot_n += d * d; tot_d += rn * rn
        worst.append((d / (rn + 1e-12), n, rn))
    worst.sort(reverse=True)
    print("  global grad rel err %.3e" % (tot_n ** 0.5 / tot_d ** 0.5))
    for w in worst[:8]: print("   grad rel err %.3e  %s (|g|=%.3e)" % w)
