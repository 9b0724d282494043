"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz, written
by oracle/make_goldens.py from /root/reference running on HuggingFace transformers)."""
import os

import numpy as np
import pytest
import torch

from oracle import vault_oracle as O
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, build_state, synthetic_batch

GOLD = os.path.join(os.path.dirname(__file__), "golden")

CASES = {
    "tiny_roberta": lambda: VaultSpec.tiny(3, "roberta"),
    "tiny_bert": lambda: VaultSpec.tiny(3, "bert"),
    "full_bertweet_b2": lambda: VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3),
}


def _run_oracle(spec, g):
    B = int(g["meta_batch"])
    batch = O.torch_batch(synthetic_batch(spec, B, seed=int(g["meta_data_seed"]), n_classes=spec.n_classes))
    P = O.to_torch_state(build_state(spec, 0), requires_grad=True)
    loss, out = O.vault_loss(P, spec, batch)
    loss.backward()
    return P, loss, out, batch


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    path = os.path.join(GOLD, f"{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    g = np.load(path)
    spec = CASES[name]()
    torch.set_num_threads(8)
    P, loss, out, batch = _run_oracle(spec, g)
    T = batch["input_ids"].shape[1]
    # fp32 vs fp32, different op order (sdpa vs eager softmax, fused vs unfused): 2e-5 abs
    np.testing.assert_allclose(out["logits"].detach().numpy(), g["logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["pooler_output"].detach().numpy(), g["pooler_output"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["last_hidden_state"][:, : T + 1].detach().numpy(), g["hidden_text_cls"],
                               atol=1e-4, rtol=0)
    # patch tokens: the reference shuffles them (D3) -> compare sorted per-token norms
    pn = np.sort(out["last_hidden_state"][:, T + 1:].detach().norm(dim=-1).numpy(), axis=1)
    np.testing.assert_allclose(pn, g["hidden_patch_sorted_norms"], rtol=1e-5)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    names = [str(n) for n in g["grad_names"]]
    norms = g["grad_norms"]
    for n, ref_norm in zip(names, norms):
        assert P[n].grad is not None, n
        mine = float(P[n].grad.double().norm())
        # key-bias gradients are analytically zero (softmax shift invariance): fp32 noise ~1e-9
        assert abs(mine - ref_norm) <= 2e-4 * ref_norm + 2e-8, (n, mine, ref_norm)
    # parameters the reference leaves without gradient must be gradient-free (or zero) here too
    for k, p in P.items():
        if k not in names:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(P[k[6:]].grad.numpy(), g[k], atol=2e-6, rtol=1e-3)


def test_adamw_formula_float64():
    """HF-4.48 AdamW (no bias correction) restated in fp32 tracks a float64 run of the same
    formula; pins eps placement (outside the sqrt) and the decay-after-update order."""
    rng = np.random.default_rng(0)
    p32 = rng.standard_normal(1000).astype(np.float32)
    p64 = p32.astype(np.float64)
    m32 = np.zeros_like(p32); v32 = np.zeros_like(p32)
    m64 = np.zeros_like(p64); v64 = np.zeros_like(p64)
    for step in range(1, 6):
        g = rng.standard_normal(1000).astype(np.float32) * 1e-2
        lr = O.linear_schedule_lr(2e-5, step - 1, 2, 10)
        O.hf_adamw_step(p32, g, m32, v32, np.float32(lr), step, weight_decay=0.01)
        O.hf_adamw_step(p64, g.astype(np.float64), m64, v64, lr, step, weight_decay=0.01)
    np.testing.assert_allclose(p32, p64, atol=1e-6)
    # first step with lr>0: update magnitude is lr * m/(sqrt(v)+eps) = lr * 0.1g/(sqrt(0.001)|g| + eps)
    p = np.ones(4, np.float64); m = np.zeros(4); v = np.zeros(4)
    g = np.array([1.0, -1.0, 1e-12, 0.0])
    O.hf_adamw_step(p, g, m, v, 1e-3, 1)
    exp = 1.0 - 1e-3 * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    np.testing.assert_allclose(p, exp, rtol=1e-12)


def test_linear_schedule():
    assert O.linear_schedule_lr(1.0, 0, 10, 100) == 0.0
    assert O.linear_schedule_lr(1.0, 5, 10, 100) == 0.5
    assert O.linear_schedule_lr(1.0, 10, 10, 100) == 1.0
    assert abs(O.linear_schedule_lr(1.0, 55, 10, 100) - 0.5) < 1e-12
    assert O.linear_schedule_lr(1.0, 100, 10, 100) == 0.0
