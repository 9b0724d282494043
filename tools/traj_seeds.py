#!/usr/bin/env python
"""is the bf16 20-step trajectory bound (tests/test_gpu_train.py, 8e-3) noise or drift?  The test's trajectory over
several DATA seeds, against the fp32 CPU oracle, on whichever bf16 library VAULT_HIP_LIB names (tree = shared-exponential GELU,
`build_variant.py gelu_old common.h:-DVAULT_GELU_SHARED_EXP=0` = the round-3 polynomial).  The oracle trajectories are cached
under gpurun_out/ so that the second library re-uses them.

  python tools/traj_seeds.py [seed ...]      (default 300 400 500)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vault_oracle as O  # noqa: E402  (checker tool, like tests/)
from vault_amd.engine import VaultEngine  # noqa: E402
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, build_state, synthetic_batch  # noqa: E402
from vault_amd.train import TrainStep  # noqa: E402


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [300, 400, 500]
    half = os.environ.get("TRAJ_HALF", "bf16")
    spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
    spec.lm.hidden_dropout_prob = 0.0
    spec.lm.attention_probs_dropout_prob = 0.0
    nsteps, B = 20, 4
    state = build_state(spec, 0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for seed in seeds:
        batches = [synthetic_batch(spec, B, seed=seed + i, n_classes=3) for i in range(4)]
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half=half)
        step = TrainStep(eng, learning_rate=2e-5, warmup_ratio=0.1, total_steps=nsteps)
        losses = []
        for i in range(nsteps):
            bn = batches[i % 4]
            db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
            losses.append(float(step(db, torch.from_numpy(bn["labels"]).cuda())))
        del step, eng
        torch.cuda.empty_cache()
        cache = os.path.join(ROOT, "gpurun_out", f"traj_ref_{seed}.json")
        if os.path.exists(cache):
            ref = json.load(open(cache))
        else:
            torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
            P = O.to_torch_state(state, requires_grad=True)
            m = {k: torch.zeros_like(v) for k, v in P.items()}
            v2 = {k: torch.zeros_like(v) for k, v in P.items()}
            ref = []
            for t in range(1, nsteps + 1):
                for p in P.values():
                    p.grad = None
                loss, _ = O.vault_loss(P, spec, O.torch_batch(batches[(t - 1) % 4]))
                loss.backward()
                ref.append(float(loss.detach()))
                lr = O.linear_schedule_lr(2e-5, t - 1, int(0.1 * nsteps), nsteps)
                with torch.no_grad():
                    for k, p in P.items():
                        if p.grad is not None:
                            O.hf_adamw_step(p, p.grad, m[k], v2[k], lr, t)
            json.dump(ref, open(cache, "w"))
        d = [abs(a - b) for a, b in zip(losses, ref)]
        print(f"{half} lib={os.path.basename(os.environ.get('VAULT_HIP_LIB', 'tree'))} data seed {seed}: max |dloss| {max(d):.2e} "
              f"(step {d.index(max(d)) + 1}), mean {sum(d) / len(d):.2e}, final {d[-1]:.2e}; loss {ref[0]:.4f} -> {ref[-1]:.4f}", flush=True)


if __name__ == "__main__":
    main()
