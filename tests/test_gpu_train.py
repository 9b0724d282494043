"""GPU tests of the optimizer kernel and of the fine-tune step (ref: vault/tmsc_utils/trainer.py:353-369)."""
import os

import numpy as np
import pytest
import torch

from oracle import vault_oracle as O
from vault_amd import ops
from vault_amd.engine import VaultEngine
from vault_amd.spec import VaultSpec, build_state, synthetic_batch
from vault_amd.train import TrainStep, linear_schedule

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("correct_bias,wd", [(False, 0.0), (True, 0.01)])
def test_adamw_kernel_matches_hf_formula(correct_bias, wd):
    n = 4096 * 3
    rng = np.random.default_rng(0)
    p = rng.standard_normal(n).astype(np.float32)
    p_ref = p.astype(np.float64)
    m_ref = np.zeros(n); v_ref = np.zeros(n)
    dp = torch.from_numpy(p).cuda(); dm = torch.zeros(n, device="cuda"); dv = torch.zeros(n, device="cuda")
    pb = dp.bfloat16()            # (the 16-bit shadow starts as the image of the parameters, as ParamStore makes it)
    for t in range(1, 5):
        g = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 0, n)).astype(np.float32)
        # rows of an embedding table: the first third never receives a gradient (round 6: the kernel then skips its stores - with
        # weight decay it must not), the second third only in the first two steps (its moments keep decaying, it keeps moving)
        g[: n // 3] = 0.0
        if t > 2:
            g[n // 3: 2 * n // 3] = 0.0
        lr = linear_schedule(2e-5, t - 1, 2, 10)
        dg = torch.from_numpy(g * 4.0).cuda()   # pretend 4 ranks summed -> grad_scale 1/4
        bc = np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) if correct_bias else 1.0
        ops.adamw_step(dp, dg, dm, dv, pb, n, lr, 0.9, 0.999, 1e-8, wd, bias_corr_factor=float(bc), grad_scale=0.25)
        O.hf_adamw_step(p_ref, g.astype(np.float64), m_ref, v_ref, lr, t, weight_decay=wd, correct_bias=correct_bias)
        torch.cuda.synchronize()
        assert float(dg.abs().max()) == 0.0          # gradients cleared by the fused kernel
    np.testing.assert_allclose(dp.cpu().numpy(), p_ref, atol=2e-7, rtol=1e-6)
    np.testing.assert_allclose(dm.cpu().numpy(), m_ref, atol=1e-8, rtol=1e-4)
    assert torch.equal(pb, dp.bfloat16())
    if wd == 0.0:      # never touched, no decay: bit-identical to where they started
        assert np.array_equal(dp[: n // 3].cpu().numpy(), p[: n // 3]) and float(dm[: n // 3].abs().max()) == 0.0
    else:
        assert not np.array_equal(dp[: n // 3].cpu().numpy(), p[: n // 3])
    assert float(np.abs(dp[n // 3: 2 * n // 3].cpu().numpy() - p[n // 3: 2 * n // 3]).min()) > 0.0     # touched once: still moving


@pytest.mark.parametrize("half", ["fp16", "bf16"])      # (the API default format and the bench's, not bf16 alone)
def test_two_step_trajectory_vs_oracle(half):
    spec = VaultSpec.tiny(3, "roberta")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, 4, seed=21, n_classes=3)
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half=half)
    step = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    losses = [float(step(db, labels)) for _ in range(3)]
    # oracle: same three steps in fp32 with the HF-AdamW formula
    P = O.to_torch_state(state, requires_grad=True)
    m = {k: torch.zeros_like(v) for k, v in P.items()}; v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    tb = O.torch_batch(bn)
    ref_losses = []
    for t in range(1, 4):
        for p in P.values():
            p.grad = None
        loss, _ = O.vault_loss(P, spec, tb)
        loss.backward()
        ref_losses.append(float(loss.detach()))
        lr = O.linear_schedule_lr(5e-5, t - 1, 0, 10)
        with torch.no_grad():
            for k, p in P.items():
                if p.grad is not None:
                    O.hf_adamw_step(p, p.grad, m[k], v2[k], lr, t)
    assert abs(losses[0] - ref_losses[0]) < (2e-3 if half == "bf16" else 3e-4)
    # sign-like AdamW updates (no bias correction: |update| ~ 3.2 lr at step 1) amplify bf16 gradient
    # noise a little: the loss trajectory must track the oracle's closely and move the same way
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 5e-3, (losses, ref_losses)
    d_ref, d_mine = ref_losses[2] - ref_losses[0], losses[2] - losses[0]
    assert d_ref < 0 and d_mine < 0 and abs(d_mine - d_ref) < 0.3 * abs(d_ref) + 2e-3, (losses, ref_losses)
    # parameters moved the same way where the reference gradient is not tiny
    w = eng.params.w("pooler.dense.weight").cpu()
    w0 = torch.from_numpy(state["pooler.dense.weight"])
    dref = P["pooler.dense.weight"].detach() - w0
    dmine = w - w0
    big = dref.abs() > 0.5 * dref.abs().max()
    assert float((torch.sign(dref[big]) == torch.sign(dmine[big])).float().mean()) > 0.97


def test_bce_single_logit_fine_tune_step_vs_oracle():
    """n_classes = 1 with float targets: the fused step's BCE-with-logits loss and its gradients (ref:
    vault/models/vault/trainer.py:55-56 over model.py:567-570) against the fp32 oracle, eager and from the tape."""
    spec = VaultSpec.tiny(1, "bert")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, 6, seed=33, n_classes=2)
    bn["labels"] = bn["labels"].astype(np.float32)            # 0. / 1. targets
    state = build_state(spec, 2)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    # forward + backward: loss, logits and head / pooler gradients
    out = eng.forward(db, train=True, labels=labels)
    eng.zero_grad(); eng.backward()
    P = O.to_torch_state(state, requires_grad=True)
    loss_ref, out_ref = O.vault_loss(P, spec, O.torch_batch(bn))
    loss_ref.backward()
    assert out["logits"].shape == (6,)
    np.testing.assert_allclose(out["logits"].cpu().numpy(), out_ref["logits"].detach().squeeze(-1).numpy(), atol=3e-3)
    assert abs(float(out["loss"]) - float(loss_ref.detach())) < 2e-3
    for name in ("classifier.1.weight", "classifier.1.bias", "pooler.dense.weight"):
        g, gr = eng.params.gr(name).cpu().reshape(-1), P[name].grad.reshape(-1)
        cos = float(torch.dot(g, gr) / (g.norm() * gr.norm() + 1e-30))
        assert cos > 0.995 and abs(float(g.norm() / gr.norm()) - 1.0) < 0.02, (name, cos)
    with pytest.raises(ValueError):     # float targets on a multi-class head
        VaultEngine(VaultSpec.tiny(3, "bert"), "cuda:0", state=build_state(VaultSpec.tiny(3, "bert"), 0)).forward(
            {k: torch.from_numpy(v).cuda() for k, v in synthetic_batch(VaultSpec.tiny(3, "bert"), 6, seed=33).items()
             if k != "labels"}, train=True, labels=labels)
    # three optimisation steps (step 2 and 3 replay the tape)
    eng2 = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    step = TrainStep(eng2, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10)
    losses = [float(step(db, labels)) for _ in range(3)]
    P = O.to_torch_state(state, requires_grad=True)
    m = {k: torch.zeros_like(v) for k, v in P.items()}; v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    ref_losses = []
    for t in range(1, 4):
        for p in P.values():
            p.grad = None
        loss, _ = O.vault_loss(P, spec, O.torch_batch(bn))
        loss.backward()
        ref_losses.append(float(loss.detach()))
        with torch.no_grad():
            for k, p in P.items():
                if p.grad is not None:
                    O.hf_adamw_step(p, p.grad, m[k], v2[k], O.linear_schedule_lr(5e-5, t - 1, 0, 10), t)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 5e-3, (losses, ref_losses)
    assert ref_losses[2] < ref_losses[0] and losses[2] < losses[0], (losses, ref_losses)


def test_bce_single_logit_against_reference_golden():
    """The fused head's BCE-with-logits on the golden the reference itself produced (VaultForTMSC(n_classes=1) +
    nn.BCEWithLogitsLoss, ref: vault/models/vault/trainer.py:55-56; tests/golden/tiny_bert_bce_n1.npz): logits, loss and
    per-parameter gradient norms."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_bert_bce_n1.npz"))
    spec = VaultSpec.tiny(1, "bert")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, int(g["meta_batch"]), seed=int(g["meta_data_seed"]), n_classes=2)
    eng = VaultEngine(spec, "cuda:0", state=build_state(spec, 0), classifier_dropout=0.0, half="bf16")
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    out = eng.forward(db, train=True, labels=torch.from_numpy(bn["labels"].astype(np.float32)).cuda())
    eng.zero_grad(); eng.backward()
    torch.cuda.synchronize()
    assert np.abs(out["logits"].cpu().numpy() - g["logits"]).max() < 3e-3
    assert abs(float(out["loss"]) - float(g["loss"])) < 2e-3
    for n, rn in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if ".key.bias" in n:
            continue
        mine = float(eng.params.gr(n).double().norm())
        assert abs(mine - rn) <= 0.08 * rn + 1e-7, (n, mine, rn)


def test_train_mode_dropout_is_active_and_reproducible():
    spec = VaultSpec.tiny(3, "roberta")
    bn = synthetic_batch(spec, 4, seed=22, n_classes=3)
    eng = VaultEngine(spec, "cuda:0", classifier_dropout=0.1, half="bf16")
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    ev = eng.forward(db, train=False, need_hidden=False)["logits"].clone()
    a = eng.forward(db, train=True, labels=labels, need_hidden=False)
    la = a["logits"].clone()
    eng.zero_grad(); eng.backward()
    b = eng.forward(db, train=True, labels=labels, need_hidden=False)["logits"].clone()
    torch.cuda.synchronize()
    assert not torch.equal(la, ev) and not torch.equal(la, b)   # new mask every step
    assert (la - ev).abs().max() < 0.2
    assert torch.isfinite(eng.params.g).all()


def test_tape_replay_matches_eager_steps():
    """TrainStep records the C-ABI call list on its first step and replays it afterwards (new inputs are
    copied into the persistent input buffers, dropout seeds are re-keyed): same trajectory as eager."""
    spec = VaultSpec.tiny(3, "roberta")          # LM + classifier dropout active
    state = build_state(spec, 0)
    batches = [synthetic_batch(spec, 4, seed=40 + i, n_classes=3) for i in range(4)]
    res = {}
    for use_tape in (False, True):
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.1, half="bf16")
        step = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10, use_tape=use_tape)
        losses = []
        for bn in batches:
            db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
            losses.append(float(step(db, torch.from_numpy(bn["labels"]).cuda())))
        res[use_tape] = (losses, eng.params.p.clone())
    torch.cuda.synchronize()
    la, lb = res[False][0], res[True][0]
    d = (res[False][1] - res[True][1]).abs()
    print("eager vs tape: loss diffs", [f"{abs(a - b):.1e}" for a, b in zip(la, lb)], f"param diff mean {float(d.mean()):.2e}",
          f"frac > 1e-5: {float((d > 1e-5).float().mean()):.4f}")
    # float atomics (split-K wgrad, LayerNorm dgamma) make two EAGER runs differ in the last bits too; AdamW's
    # sign-like steps amplify that over steps: identical at step 1, 0 - 1.4e-5 at step 2 (a sign flip of a ~0 gradient in
    # the classifier moves the loss by that much), ~1e-4 by step 4 (printed above)
    assert abs(la[0] - lb[0]) < 1e-6 and abs(la[1] - lb[1]) < 5e-5, (la, lb)
    assert max(abs(a - b) for a, b in zip(la, lb)) < 5e-4, (la, lb)
    assert len(set(round(x, 4) for x in lb)) > 1                          # different batches -> different losses
    # sign-like AdamW steps flip on ~0 gradients whose float-atomic sums differ in the last bit: compare in bulk
    d = (res[False][1] - res[True][1]).abs()
    assert float(d.mean()) < 1e-6 and float((d > 1e-5).float().mean()) < 0.02



def test_train_step_on_padded_image_batches_mixed_with_square_ones():
    """The fine-tune step on batches of differently sized, padded images (general image path, one tape per image
    geometry) interleaved with ordinary square batches: same trajectory with and without tape replay, and the
    loss goes down when one padded batch is repeated."""
    from vault_amd.spec import synthetic_ragged_batch
    spec = VaultSpec.tiny(3, "bert")
    state = build_state(spec, 0)
    geos = [([(96, 160), (128, 64), (80, 80), (128, 160)], (128, 160)),
            ([(128, 160), (64, 64), (96, 96), (32, 160)], (128, 160)),      # same canvas, other image sizes
            None,                                                             # square all-valid batch
            ([(96, 160), (128, 64), (80, 80), (128, 160)], (128, 160))]
    batches = []
    for i, gspec in enumerate(geos):
        if gspec is None:
            batches.append(synthetic_batch(spec, 4, seed=70 + i, n_classes=3))
        else:
            batches.append(synthetic_ragged_batch(spec, gspec[0], gspec[1], seed=70 + i, n_classes=3))
    res = {}
    for use_tape in (False, True):
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
        step = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10, use_tape=use_tape)
        losses = []
        for bn in batches:
            db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
            losses.append(float(step(db, torch.from_numpy(bn["labels"]).cuda())))
        res[use_tape] = losses
    la, lb = res[False], res[True]
    assert max(abs(a - b) for a, b in zip(la, lb)) < 5e-4, (la, lb)
    # repeat one padded batch: the loss must fall
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    step = TrainStep(eng, learning_rate=2e-4, warmup_ratio=0.0, total_steps=100, constant_lr=True)
    bn = batches[0]
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    lab = torch.from_numpy(bn["labels"]).cuda()
    ls = [float(step(db, lab)) for _ in range(12)]
    assert ls[-1] < ls[0] - 0.05, ls



def _dp_spec(kind):
    """``kind``: "roberta" / "bert" = the tiny test model; "full-width" = hidden 768, FFN 3072, BERTweet's 64,001-row
    vocabulary, 2 + 2 layers (80 M parameters: several default-size buckets, the real row-sparse table)."""
    if kind == "full-width":
        from vault_amd.spec import LMSpec, ViltSpec
        spec = VaultSpec(vilt=ViltSpec(num_hidden_layers=2), lm=LMSpec.bertweet_base(), n_classes=3)
        spec.lm.num_hidden_layers = 2
    else:
        spec = VaultSpec.tiny(3, kind)
    spec.lm.hidden_dropout_prob = 0.0
    spec.lm.attention_probs_dropout_prob = 0.0
    return spec


def _dp_worker(rank, world, port, out_path, kind, nsteps, use_tape, wire, sparse, bucket_mb=0.25, half="bf16"):
    """One data-parallel rank (both ranks share cuda:0; gloo carries the device tensors): its half of every batch."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        spec = _dp_spec(kind)
        eng = VaultEngine(spec, "cuda:0", state=build_state(spec, 0), classifier_dropout=0.0, half=half)
        kw = {} if bucket_mb is None else dict(bucket_mb=bucket_mb)          # None: TrainStep's default (64 MB)
        step = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10, use_tape=use_tape,
                         wire=wire, sparse_embedding=sparse, **kw)
        assert step.world == world and step.reducer is not None and ops.GEMM_SCHED == 3
        assert (step.reducer.sparse is not None) == sparse and step.reducer.wire == wire
        losses, wire_bytes = [], []
        for i in range(nsteps):
            bn = synthetic_batch(spec, 8, seed=90 + i, n_classes=3)
            lo, hi = rank * (8 // world), (rank + 1) * (8 // world)
            db = {k: torch.from_numpy(v[lo:hi]).cuda() for k, v in bn.items() if k != "labels"}
            losses.append(float(step(db, torch.from_numpy(bn["labels"][lo:hi]).cuda())))
            wire_bytes.append(step.reducer.wire_bytes)
        torch.cuda.synchronize()
        torch.save({"p": eng.params.p.cpu(), "losses": losses, "wire_bytes": wire_bytes, "n_train": eng.params.n_train,
                    "launched": len(step.reducer.launched) if step.reducer.launched else None,
                    "bucket_elems": step.reducer.bucket_elems, "union_waits": step.reducer.union_waits,
                    "sparse_checks": step.reducer.sparse_checks},
                   f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_tape,wire,sparse,world", [(False, "fp32", True, 2), (True, "fp32", True, 2), (True, "fp32", False, 2),
                                                        (True, "bf16", True, 2), (False, "bf16", False, 2), (True, "bf16", True, 4),
                                                        (True, "fp32", True, 4)])
def test_data_parallel_two_ranks_equal_one_rank_on_the_global_batch(tmp_path, use_tape, wire, sparse, world):
    """The whole N > 1 path on real kernels: two (or four) processes (sharing the one GPU, gloo as the transport) each step
    their share of a global batch of 8 - bucketed gradient exchange from inside backward on a side stream (f32 all-reduce or
    bf16 reduce-scatter + all-gather; the word-embedding table row-sparse or dense), dynamic GEMM scheduling, AdamW
    dividing by the world size - and must land where ONE process stepping the 8 samples lands."""
    import torch.multiprocessing as mp
    nsteps = 3
    out = str(tmp_path / "dp")
    port = 29600 + (1 if use_tape else 0) + (2 if wire == "bf16" else 0) + (4 if sparse else 0) + (8 if world == 4 else 0)
    mp.spawn(_dp_worker, args=(world, port, out, "roberta", nsteps, use_tape, wire, sparse), nprocs=world, join=True)
    rs = [torch.load(out + f".{r}") for r in range(world)]
    r0, r1 = rs[0], rs[1]
    for r in rs[1:]:
        assert torch.equal(r0["p"], r["p"])                    # replicas stay bit-identical
    dense_fp32 = 2 * 4 * r0["n_train"] * (world - 1) // world  # bytes a rank sends in a dense f32 ring all-reduce
    print(f"wire bytes per step and rank: {r0['wire_bytes']} (dense f32 all-reduce: {dense_fp32})")
    if sparse:        # (the tiny model's table is 1 % of its gradient: the saving is small here, 197 MB of 890 MB at full size)
        assert max(r0["wire_bytes"]) < (1.0 if wire == "fp32" else 0.55) * dense_fp32
    elif wire == "bf16":
        assert max(r0["wire_bytes"]) < 0.55 * dense_fp32
    spec = VaultSpec.tiny(3, "roberta")
    spec.lm.hidden_dropout_prob = 0.0
    spec.lm.attention_probs_dropout_prob = 0.0
    eng = VaultEngine(spec, "cuda:0", state=build_state(spec, 0), classifier_dropout=0.0, half="bf16")
    step = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10, use_tape=False)
    ref_losses = []
    for i in range(nsteps):
        bn = synthetic_batch(spec, 8, seed=90 + i, n_classes=3)
        db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
        ref_losses.append(float(step(db, torch.from_numpy(bn["labels"]).cuda())))
    torch.cuda.synchronize()
    # the global mean loss is the mean of the two local means
    for i, c in enumerate(ref_losses):
        mean_local = sum(r["losses"][i] for r in rs) / world
        assert abs(mean_local - c) < 5e-4, (mean_local, c)
    d = (r0["p"] - eng.params.p.cpu()).abs()
    if wire == "fp32":
        # same tolerance as tape-vs-eager: float-atomic summation order + sign-like AdamW steps on ~0 gradients
        assert float(d.mean()) < 2e-6 and float((d > 1e-5).float().mean()) < 0.03
    else:
        # bf16 on the wire: every summed gradient element carries one bf16 rounding (2^-9 relative) into AdamW
        print(f"bf16 wire: mean |dp| {float(d.mean()):.2e}, share above 2e-5: {float((d > 2e-5).float().mean()):.4f}")
        assert float(d.mean()) < 4e-6 and float((d > 2e-5).float().mean()) < 0.03


def test_exchange_kernels_match_their_host_restatements():
    """csrc/exchange.hip against the torch restatements the gloo CPU test drives the reducer with (tests/test_host.py
    HostKernels): union of token ids, row gather / scatter, f32 sum of bf16 chunks in rank order, widening."""
    from tests.test_host import HostKernels as HK
    from vault_amd.train import ExchangeKernels as DK
    g = torch.Generator().manual_seed(5)
    V, H, n = 64001, 768, 4 * 1280
    keys = torch.randint(0, V, (n,), generator=g, dtype=torch.int64)
    keys[::7] = 1
    keys[5], keys[6] = -3, V + 2                                 # outside the table: ignored
    flags_h, uniq_h, cnt_h = torch.zeros(V, dtype=torch.int32), torch.zeros(n, dtype=torch.int64), torch.zeros(2, dtype=torch.int32)
    HK.rows_union(keys, n, V, flags_h, uniq_h, cnt_h)
    flags_d = torch.zeros(V, dtype=torch.int32, device="cuda"); uniq_d = torch.zeros(n, dtype=torch.int64, device="cuda")
    cnt_d = torch.zeros(2, dtype=torch.int32, device="cuda")
    for _ in range(2):                                           # (the scratch flags are left zeroed: second call equal)
        DK.rows_union(keys.cuda(), n, V, flags_d, uniq_d, cnt_d)
        U = int(cnt_d[0].item())
        assert int(cnt_d[1].item()) == int(cnt_h[1]) == 2           # the two ids outside the table are counted (-1 would not be)
        assert U == int(cnt_h[0]) and torch.equal(uniq_d[:U].cpu(), uniq_h[:U]) and int(flags_d.abs().sum()) == 0
    table = torch.randn(V, H, generator=g)
    td = table.cuda()
    comp_d = torch.zeros(U * H, device="cuda")
    DK.rows_gather(td, uniq_d, U, H, comp_d)
    assert torch.equal(comp_d.cpu().view(U, H), table[uniq_h[:U]])
    comp_d.mul_(2.0)
    DK.rows_scatter(comp_d, uniq_d, U, H, td)
    want = table.clone(); want[uniq_h[:U]] *= 2.0
    assert torch.equal(td.cpu(), want)
    W, chunk = 8, 8 * 1237
    src = (torch.randn(W * chunk, generator=g) * 3.0).to(torch.bfloat16)
    out_h = torch.zeros(chunk, dtype=torch.bfloat16); out_d = torch.zeros(chunk, dtype=torch.bfloat16, device="cuda")
    HK.sum_chunks(src, W, chunk, out_h)
    DK.sum_chunks(src.cuda(), W, chunk, out_d)
    assert torch.equal(out_d.cpu(), out_h)
    wide = torch.zeros(chunk, device="cuda")
    DK.widen(out_d, wide, chunk)
    assert torch.equal(wide.cpu(), out_h.float())


def test_evaluate_pass_matches_oracle_predictions():
    """The reference's evaluation pass (eval-mode forward, mean CE, argmax, accuracy, macro-F1) on the HIP engine vs
    the same quantities from the CPU oracle."""
    from vault_amd.train import evaluate, evaluation_metrics
    spec = VaultSpec.tiny(3, "roberta")
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, with_grads=False, half="bf16")
    batches, ref_pred, ref_true, ref_loss, n = [], [], [], 0.0, 0
    P = O.to_torch_state(state)
    for i, B in enumerate((5, 3, 4)):
        bn = synthetic_batch(spec, B, seed=200 + i, n_classes=3)
        db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
        batches.append((db, torch.from_numpy(bn["labels"]).cuda()))
        loss, out = O.vault_loss(P, spec, O.torch_batch(bn))
        ref_pred += out["logits"].argmax(-1).tolist(); ref_true += bn["labels"].tolist()
        ref_loss += float(loss) * B; n += B
    res = evaluate(eng, batches)
    want = evaluation_metrics(ref_true, ref_pred)
    assert abs(res["eval_loss"] - ref_loss / n) < 2e-3
    # random-init logits are close to each other: allow one flipped argmax out of 12
    assert abs(res["eval_accuracy"] - want["eval_accuracy"]) <= 1.0 / n + 1e-9
    assert 0.0 <= res["macro_f1_score"] <= 1.0


@pytest.mark.parametrize("half", ["fp16", "bf16"])
def test_frozen_lm_train_step_trajectory_vs_oracle(half):
    """BASELINE config 4's step: ``TrainStep`` on an engine with ``freeze_lm=True`` (LM forward only, gradient buckets
    ending at the ViLT embeddings, AdamW over the shortened trainable range) for three steps, against the oracle
    stepping only the non-LM parameters with the HF-AdamW formula; the frozen parameters must not move at all."""
    spec = VaultSpec.tiny(3, "bert")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, 4, seed=31, n_classes=3)
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, freeze_lm=True, half=half)
    step = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    losses = [float(step(db, labels)) for _ in range(3)]
    P = O.to_torch_state(state, requires_grad=True)
    m = {k: torch.zeros_like(v) for k, v in P.items()}; v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    tb = O.torch_batch(bn)
    ref_losses = []
    for t in range(1, 4):
        for p in P.values():
            p.grad = None
        loss, _ = O.vault_loss(P, spec, tb)
        loss.backward()
        ref_losses.append(float(loss.detach()))
        lr = O.linear_schedule_lr(5e-5, t - 1, 0, 10)
        with torch.no_grad():
            for k, p in P.items():
                if p.grad is not None and not k.startswith("bert."):
                    O.hf_adamw_step(p, p.grad, m[k], v2[k], lr, t)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 5e-3, (losses, ref_losses)
    d_ref, d_mine = ref_losses[2] - ref_losses[0], losses[2] - losses[0]
    assert d_ref < 0 and d_mine < 0 and abs(d_mine - d_ref) < 0.3 * abs(d_ref) + 2e-3, (losses, ref_losses)
    for n in ("bert.embeddings.word_embeddings.weight", "bert.encoder.layer.1.output.dense.weight"):
        assert torch.equal(eng.params.w(n).cpu(), torch.from_numpy(state[n]))      # frozen: bit-identical
    w = eng.params.w("pooler.dense.weight").cpu()
    w0 = torch.from_numpy(state["pooler.dense.weight"])
    dref = P["pooler.dense.weight"].detach() - w0
    big = dref.abs() > 0.5 * dref.abs().max()
    assert float((torch.sign(dref[big]) == torch.sign((w - w0)[big])).float().mean()) > 0.97


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_train_step_over_rccl_single_rank(wire, monkeypatch):
    """The data-parallel path on its production transport: ``torch.distributed`` backend "nccl" (= RCCL on ROCm) with one
    rank and VAULT_FORCE_DP=1 - bucketed all-reduce launched from inside backward on a side stream, split optimizer step
    (upper range while the last bucket is on the wire).  With one rank the all-reduce is the identity: the parameters
    must land exactly where the plain single-process step lands."""
    import os
    import torch.distributed as dist
    from vault_amd.train import BucketReducer
    monkeypatch.setattr(BucketReducer, "SINGLE_RANK_COLLECTIVES", True)      # (the RCCL calls themselves, with one rank)
    spec = VaultSpec.tiny(3, "roberta")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, 8, seed=41, n_classes=3)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()

    def run(dp):
        eng = VaultEngine(spec, "cuda:0", state=build_state(spec, 0), classifier_dropout=0.0, half="bf16")
        st = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10, bucket_mb=0.05, wire=wire)
        assert (st.reducer is not None) == dp
        if dp:   # RCCL's all_gather_into_tensor (token ids) + all_reduce / all_to_all_single + all_gather_into_tensor
            assert st.reducer.native_a2a and st.reducer.sparse is not None and st.reducer.wire == wire
        ls = [float(st(db, labels)) for _ in range(3)]
        torch.cuda.synchronize()
        if dp:
            assert len(st.reducer.launched) == 0 and st.reducer.hi == st.reducer.n     # every bucket waited for
        return ls, eng.params.p.clone()

    ref_l, ref_p = run(False)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29641" if wire == "fp32" else "29642", VAULT_FORCE_DP="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        dp_l, dp_p = run(True)
    finally:
        dist.destroy_process_group()
        os.environ.pop("VAULT_FORCE_DP", None)
    # (float-atomic summation order in the weight gradients + sign-like AdamW steps: two runs of the SAME path differ by
    #  a few 1e-5 in the later losses)
    assert max(abs(a - b) for a, b in zip(dp_l, ref_l)) < 5e-4
    d = (dp_p - ref_p).abs()
    if wire == "fp32":
        assert float(d.mean()) < 2e-6 and float((d > 1e-5).float().mean()) < 0.03   # float-atomic summation order only
    else:                                                                          # + one bf16 rounding of every gradient
        assert float(d.mean()) < 4e-6 and float((d > 2e-5).float().mean()) < 0.03


_TRAJ = {}


def _fp32_oracle_trajectory(spec, state, batches, nsteps):
    """20 fine-tune steps of the fp32 CPU oracle (computed once per session: both operand formats compare with it)."""
    import os
    if "ref" not in _TRAJ:
        torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
        P = O.to_torch_state(state, requires_grad=True)
        m = {k: torch.zeros_like(v) for k, v in P.items()}; v2 = {k: torch.zeros_like(v) for k, v in P.items()}
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
        _TRAJ["ref"] = ref
    return _TRAJ["ref"]


@pytest.mark.parametrize("half", ["fp16", "bf16"])
def test_full_size_loss_trajectory_20_steps_vs_fp32_oracle(half):
    """What a fine-tune user sees: 20 optimisation steps of the full-size model (12 + 12 layers, B = 4, lr 2e-5 with the
    reference's 10 % linear warm-up, HF-AdamW without bias correction) against the same 20 steps of the fp32 CPU oracle (the
    measured deviations are printed).  fp16 operands (the API default): every loss of the trajectory inside the north star's
    1e-3.  bf16 operands: the FIRST step carries the format's own error (|dloss| 1.7e-3 on one forward: bound 3e-3); along the
    trajectory AdamW's sign-like early steps (no bias correction: m / sqrt(v) = +-3.2 at step 1 whatever the gradient's size)
    amplify rounding noise chaotically - over 11 data seeds the per-trajectory MAXIMUM lies anywhere between 1.7e-3 and 2.3e-2
    with either form of the GELU epilogue (round-3 polynomial: 2.1e-3 .. 1.4e-2; round 4's shared exponential: 1.7e-3 ..
    2.3e-2; higher in 6 of the 11 seeds, lower in 5), the per-trajectory MEAN between 0.4e-3 and 2.7e-3
    (profiles/r05_bf16_trajectory_seeds.txt, tools/traj_seeds.py).  So the bf16 assertions are on the first step, on the
    mean (4e-3) and on the envelope (3e-2) - a single seed's maximum against a tight bound (6e-3, then 8e-3 in round 4) only
    tested which way the noise fell."""
    from vault_amd.spec import LMSpec, ViltSpec
    spec = VaultSpec(vilt=ViltSpec(), lm=LMSpec.bertweet_base(), n_classes=3)
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    nsteps, B = 20, 4
    state = build_state(spec, 0)
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half=half)
    step = TrainStep(eng, learning_rate=2e-5, warmup_ratio=0.1, total_steps=nsteps)
    batches = [synthetic_batch(spec, B, seed=300 + i, n_classes=3) for i in range(4)]
    losses = []
    for i in range(nsteps):
        bn = batches[i % 4]
        db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
        losses.append(float(step(db, torch.from_numpy(bn["labels"]).cuda())))
    del step, eng
    torch.cuda.empty_cache()
    ref = _fp32_oracle_trajectory(spec, state, batches, nsteps)
    diffs = [abs(a - b) for a, b in zip(losses, ref)]
    print(f"{half}: 20-step trajectory: max |dloss| {max(diffs):.2e}, final |dloss| {diffs[-1]:.2e}, loss {ref[0]:.4f} -> {ref[-1]:.4f}")
    if half == "fp16":
        assert max(diffs) < 1e-3, (max(diffs), losses, ref)
    else:
        # (the tight bound sits on what is deterministic - the first TWO losses are forwards on the initial weights,
        #  the schedule's first learning rate is 0 - the chaotic part of the trajectory keeps the statistical bounds)
        assert max(diffs[:2]) < 3e-3 and sum(diffs) / len(diffs) < 4e-3 and max(diffs) < 3e-2, (diffs, losses, ref)
    assert ref[-1] < ref[0] and losses[-1] < losses[0]                 # both trajectories descend
    # the drop over the run agrees within 15 %
    assert abs((losses[0] - losses[-1]) - (ref[0] - ref[-1])) < 0.15 * abs(ref[0] - ref[-1]) + 2e-3


def test_8bit_gelu_prime_follows_the_kernel_choice():
    """The 8-bit tile-native gelu' needs the 8-wave kernel on both FFN GEMMs: on with the static and with the data-parallel
    GEMM scheduling (the 8-wave kernel runs in both), off with the ring kernel forced (VAULT_GEMM8W=0 is read once per
    process: emulated here by GELU8 = False) and for small batches (128x128 tiles / stage-level layer calls) - with the same
    loss and bf16-level gradient agreement either way."""
    from vault_amd.spec import LMSpec, ViltSpec
    spec = VaultSpec(vilt=ViltSpec(num_hidden_layers=1), lm=LMSpec.bertweet_base(), n_classes=3)
    spec.lm.num_hidden_layers = 1
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    state = build_state(spec, 1)
    bn = synthetic_batch(spec, 48, seed=9, n_classes=3)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    losses, cfgs, grads = [], [], []
    for sched, g8 in ((0, True), (3, True), (0, False)):
        old = ops.GEMM_SCHED
        ops.GEMM_SCHED = sched
        try:
            eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
            eng.GELU8 = g8
            out = eng.forward(db, train=True, labels=labels, need_hidden=False)
            eng.zero_grad(); eng.backward()
            torch.cuda.synchronize()
        finally:
            ops.GEMM_SCHED = old
        losses.append(float(out["loss"])); cfgs.append(eng.last.get("gelu8_cfg"))
        grads.append(eng.params.gr("encoder.layer.0.intermediate.dense.weight").clone())
    assert cfgs[0] in (5, 6) and cfgs[1] == cfgs[0] and cfgs[2] is None, cfgs
    assert max(abs(losses[0] - l) for l in losses) < 1e-4
    for g in grads[:2]:
        rel = float((g - grads[2]).norm() / grads[2].norm())
        assert rel < 2e-2, rel                   # 8-bit against bf16 gelu': bf16-level agreement
    small = VaultEngine(VaultSpec.tiny(3, "bert"), "cuda:0", state=build_state(VaultSpec.tiny(3, "bert"), 0))
    sb = synthetic_batch(VaultSpec.tiny(3, "bert"), 4, seed=1)
    small.forward({k: torch.from_numpy(v).cuda() for k, v in sb.items() if k != "labels"}, train=True,
                  labels=torch.from_numpy(sb["labels"]).cuda())
    assert small.last.get("gelu8_cfg") is None


def test_bench_refuses_two_ranks_on_the_one_gpu_box():
    """On the 1-GPU box: `python bench.py --gpus 2` exits 2 with the refusal text (devices counted from sysfs in the launcher)."""
    from .test_host import _bench_refusal
    have = _bench_refusal()
    assert have >= 1


def test_bench_line_carries_the_contract_fields_and_the_block_fractions():
    """`python bench.py` as the driver runs it (a short loop at per-GPU batch 32 here): ONE JSON line on stdout with the
    contract's fields, the `roofline` / `cpu_baseline`-shaped objects, and round 5's driver-visible north-star numbers -
    `vilt_block_frac` / `lm_block_frac` with the four block times behind them - consistent with each other; `--scaling strong`
    on one rank is the same workload labelled "strong"."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--batch", "256", "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
           "--no-other-configs", "--no-h2d", "--no-parity", "--scaling", "strong"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "step_mfma_frac", "vilt_block_frac", "lm_block_frac", "blocks"):
        assert k in d, k
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["dtype"] == "bf16" and d["config"]["global_batch"] == 256
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(d["value"] - 256 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]
    bv, bl = d["blocks"]["vilt"], d["blocks"]["lm"]
    assert 0.2 < d["vilt_block_frac"] < 0.6 and 0.1 < d["lm_block_frac"] < 0.6
    assert abs(d["vilt_block_frac"] - 256 * 98.06e9 / ((bv["ms_forward"] + bv["ms_backward"]) * 1e-3) / 2.5e15) < 2e-3
    assert abs(d["lm_block_frac"] - 256 * 20.56e9 / ((bl["ms_forward"] + bl["ms_backward"]) * 1e-3) / 2.5e15) < 2e-3
    # the two stacks' blocks are most of the step, never more than it
    blocks_ms = bv["ms_forward"] + bv["ms_backward"] + bl["ms_forward"] + bl["ms_backward"]
    assert 0.85 * d["ms_per_step"] < blocks_ms < d["ms_per_step"]


@pytest.mark.parametrize("wire,half", [("fp32", "bf16"), ("bf16", "bf16"), ("fp32", "fp16")])
def test_data_parallel_full_width_default_buckets(tmp_path, wire, half):
    """The data-parallel step at FULL WIDTH with the DEFAULT bucket size: hidden 768, FFN 3072, the 64,001 x 768 word-embedding
    table exchanged row-sparse, 2 + 2 layers (80 M gradient elements = five 64 MB buckets launched from inside backward, the
    split optimizer pass over the already reduced upper range), two ranks on the one GPU over gloo, both wire formats and the
    fp16 operand build (scaled gradients on the wire, divided out in the optimizer): replicas bit-identical, parameters where
    ONE process stepping the global batch lands."""
    import torch.multiprocessing as mp
    nsteps, world = 2, 2
    out = str(tmp_path / "dpfw")
    port = 29700 + (1 if wire == "bf16" else 0) + (2 if half == "fp16" else 0)
    # (the first step also runs the debug comparison of the row-sparse exchange with a dense all-reduce of the table)
    os.environ["VAULT_DP_CHECK_SPARSE"] = "1"
    try:
        mp.spawn(_dp_worker, args=(world, port, out, "full-width", nsteps, True, wire, True, None, half), nprocs=world, join=True)
    finally:
        del os.environ["VAULT_DP_CHECK_SPARSE"]
    rs = [torch.load(out + f".{r}") for r in range(world)]
    assert rs[0]["sparse_checks"] == 1 and rs[1]["sparse_checks"] == 1
    assert torch.equal(rs[0]["p"], rs[1]["p"])                                   # replicas stay bit-identical
    assert rs[0]["bucket_elems"] == 64 * 1024 * 1024 // 4                        # the default bucket size was in force
    dense_fp32 = 2 * 4 * rs[0]["n_train"] * (world - 1) // world
    print(f"full width, {wire} wire, {half} operands: bytes per step and rank {rs[0]['wire_bytes']} (dense f32 all-reduce: "
          f"{dense_fp32}); host waits for the row union: {rs[0]['union_waits']}")
    # the word-embedding table is 61 % of this model's gradient; 2 ranks x 4 x 40 token ids touch <= 320 of its 64,001 rows
    assert max(rs[0]["wire_bytes"]) < (0.45 if wire == "fp32" else 0.25) * dense_fp32
    spec = _dp_spec("full-width")
    eng = VaultEngine(spec, "cuda:0", state=build_state(spec, 0), classifier_dropout=0.0, half=half)
    step = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10, use_tape=False)
    ref_losses = []
    for i in range(nsteps):
        bn = synthetic_batch(spec, 8, seed=90 + i, n_classes=3)
        db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
        ref_losses.append(float(step(db, torch.from_numpy(bn["labels"]).cuda())))
    torch.cuda.synchronize()
    for i, c in enumerate(ref_losses):
        mean_local = sum(r["losses"][i] for r in rs) / world
        assert abs(mean_local - c) < 5e-4, (mean_local, c)
    d = (rs[0]["p"] - eng.params.p.cpu()).abs()
    print(f"  against one process on the global batch: mean |dp| {float(d.mean()):.2e}, share above 2e-5: {float((d > 2e-5).float().mean()):.4f}")
    if wire == "fp32":
        assert float(d.mean()) < 2e-6 and float((d > 1e-5).float().mean()) < 0.03
    else:
        assert float(d.mean()) < 4e-6 and float((d > 2e-5).float().mean()) < 0.03


def test_train_step_after_an_api_backward_starts_from_zero_gradients():
    """``TrainStep`` stores its un-split weight-gradient tiles (it counts on a zero gradient buffer: the fused AdamW leaves one).
    Gradients an API-level ``engine.backward()`` left behind are cleared on entry instead of being half overwritten, half
    accumulated: the step lands where a fresh engine's step lands."""
    spec = VaultSpec.tiny(3, "roberta")
    spec.lm.hidden_dropout_prob = 0.0
    spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, 4, seed=12, n_classes=3)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    state = build_state(spec, 0)
    a = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    a.forward(dict(db), train=True, labels=labels)
    a.backward()                                    # leaves gradients in the flat buffer
    assert a._g_dirty
    sa = TrainStep(a, learning_rate=1e-4, warmup_ratio=0.0, total_steps=10, constant_lr=True)
    la = float(sa(db, labels))
    b = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    sb = TrainStep(b, learning_rate=1e-4, warmup_ratio=0.0, total_steps=10, constant_lr=True)
    lb = float(sb(db, labels))
    torch.cuda.synchronize()
    assert la == lb and not a._g_dirty
    d = (a.params.p - b.params.p).abs()
    assert float(d.mean()) < 2e-6 and float((d > 2e-5).float().mean()) < 0.03      # float-atomic summation order only


@pytest.mark.parametrize("half", ["bf16", "fp16"])
def test_optimizer_leaves_stored_weight_gradient_ranges_unzeroed_and_nobody_builds_on_them(half, monkeypatch):
    """The fused AdamW skips the zeroing of the weight-gradient matrices the next step's un-split launches STORE into
    (vault_adamw_step zero_mask, 4 of its 34 B/param): full width 2 + 2, B = 24 - (1) the mask covers most of the encoder
    matrices and the trajectory equals the one with the mask off (VAULT_ADAMW_ZERO_MASK=0) up to float-atomic summation order;
    (2) the un-zeroed ranges really hold old gradients after a step, everything else is zero; (3) a step of ANOTHER shape
    (other tape: its launches may accumulate where these stored) and an API-level backward both start from a cleared buffer."""
    spec = _dp_spec("full-width")
    state = build_state(spec, 1)
    batches = [synthetic_batch(spec, 24, seed=900 + i, n_classes=3) for i in range(3)]
    small = synthetic_batch(spec, 8, seed=950, n_classes=3)

    def run(mask_on):
        monkeypatch.setenv("VAULT_ADAMW_ZERO_MASK", "1" if mask_on else "0")
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half=half)
        st = TrainStep(eng, learning_rate=5e-5, warmup_ratio=0.0, total_steps=10)
        losses = []
        for bn in batches + [small, batches[0]]:
            db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
            losses.append(float(st(db, torch.from_numpy(bn["labels"]).cuda())))
        torch.cuda.synchronize()
        return eng, st, losses

    ea, sa, la = run(True)
    assert sa._zero_mask is not None
    zm = sa._zero_mask.cpu().numpy()
    share = 1.0 - float(zm.mean())
    P = ea.params
    assert share > 0.25, share                      # (80 M parameters of which 49 M are the embedding table: the encoder matrices)
    g = P.g[:P.n_train].cpu().numpy()
    keep = np.repeat(zm == 0, 64)
    assert float(np.abs(g[~keep]).max()) == 0.0 and float(np.abs(g[keep]).max()) > 0.0 and ea._g_stale_key is not None
    eb, sb, lb = run(False)
    assert sb._zero_mask is None and float(eb.params.g[:P.n_train].abs().max()) == 0.0
    print(f"{half}: zero mask skips {share:.1%} of the gradient buffer; losses {la} / {lb}")
    # (two runs differ in the order of their float atomics; AdamW's sign-like early steps amplify that along the trajectory -
    #  see test_tape_replay_matches_eager_steps - bf16 operands faster than fp16: measured 2.5e-4 / 1.3e-5 at the fifth step)
    assert abs(la[0] - lb[0]) < 1e-6 and max(abs(a - b) for a, b in zip(la, lb)) < (1e-3 if half == "bf16" else 1e-4)
    d = (ea.params.p - eb.params.p).abs()
    assert float(d.mean()) < 2e-6 and float((d > 2e-5).float().mean()) < 0.03
    # an API-level backward after the fused steps accumulates onto a CLEARED buffer: equal to the same backward on the mask-off engine
    bn = batches[1]
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items()}
    for e in (ea, eb):
        e.params.load_numpy(state)
        e.forward(db, train=True, labels=db["labels"], need_hidden=False)
        e.backward()
    torch.cuda.synchronize()
    ga, gb = ea.params.g[:P.n_train], eb.params.g[:P.n_train]
    assert ea._g_stale_key is None and float((ga - gb).norm() / gb.norm()) < 1e-5
    # a fused step, then an API-level backward and a MANUAL optimizer_step(): the public call takes no zero mask, so
    # every gradient it read is cleared - the next API-level backward does not build on the previous one's weight gradients
    db0 = {k: torch.from_numpy(v).cuda() for k, v in batches[0].items() if k != "labels"}
    sa(db0, torch.from_numpy(batches[0]["labels"]).cuda())
    assert ea._g_stale_key is not None
    ea.forward(db, train=True, labels=db["labels"], need_hidden=False)
    ea.backward()
    sa.optimizer_step()
    torch.cuda.synchronize()
    assert float(ea.params.g[:P.n_train].abs().max()) == 0.0


def test_adopted_pixel_patches_feed_the_recorded_step_without_a_copy():
    """``VaultEngine.adopt_pixel_patches``: the caller's ``pixel_patches`` tensor IS the patch-embedding GEMM's operand (forward
    and weight gradient) - a loader that alternates two tensors saves the device-to-device restage.  The recorded step re-points
    its two launches at each step's tensor (ops.Tape.rebind): same trajectory as the copying path, on different batches; the
    engine's own buffer is not written; a tensor that does not qualify (row count not a whole padded count) is copied."""
    import torch.nn.functional as F
    spec = VaultSpec.tiny(3, "roberta")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    state = build_state(spec, 0)
    B, ps = 64, spec.vilt.patch_size                    # 64 x 36 patches: a whole number of 256-row panels
    batches = [synthetic_batch(spec, B, seed=70 + i, n_classes=3) for i in range(4)]
    res = {}
    for adopt in (False, True):
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
        eng.adopt_pixel_patches = adopt
        step = TrainStep(eng, learning_rate=1e-4, warmup_ratio=0.0, total_steps=10, assume_full_pixel_mask=True)
        two = [torch.empty(B * spec.vilt.num_patches, 3 * ps * ps, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
        losses = []
        for i, bn in enumerate(batches):
            pv = torch.from_numpy(bn["pixel_values"]).cuda()
            two[i % 2].copy_(F.unfold(pv, kernel_size=ps, stride=ps).transpose(1, 2).reshape(two[0].shape))
            db = {"input_ids": torch.from_numpy(bn["input_ids"]).cuda(), "attention_mask": torch.from_numpy(bn["attention_mask"]).cuda(),
                  "pixel_patches": two[i % 2]}
            losses.append(float(step(db, torch.from_numpy(bn["labels"]).cuda())))
        if adopt:
            assert len(step._tape.rebinds) == 2
            own = eng.input_buffers(B, batches[0]["input_ids"].shape[1])["pixel_patches"]
            assert float(own.float().abs().sum()) == 0.0                  # never written: the GEMMs read the caller's tensors
        res[adopt] = (losses, eng.params.p.clone())
    torch.cuda.synchronize()
    la, lb = res[False][0], res[True][0]
    assert abs(la[0] - lb[0]) < 1e-6 and max(abs(a - b) for a, b in zip(la, lb)) < 5e-4, (la, lb)
    assert len(set(round(x, 4) for x in lb)) > 1
    d = (res[False][1] - res[True][1]).abs()
    assert float(d.mean()) < 3e-6 and float((d > 2e-5).float().mean()) < 0.03
    # 4 x 36 rows are not a whole padded panel count: copied, as without the flag
    eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
    eng.adopt_pixel_patches = True
    bn = synthetic_batch(spec, 4, seed=2, n_classes=3)
    po = F.unfold(torch.from_numpy(bn["pixel_values"]).cuda(), kernel_size=ps, stride=ps).transpose(1, 2).reshape(4 * spec.vilt.num_patches, -1).bfloat16()
    ws = eng.stage_inputs({"input_ids": torch.from_numpy(bn["input_ids"]).cuda(), "pixel_patches": po}, True, None)
    assert not ws["patch_adopted"] and ws["apatch_in"].data_ptr() != po.data_ptr()


def test_vilt_weight_gradients_beside_the_lm_backward_give_the_same_step():
    """``VaultEngine.WGRAD_BESIDE_LM_ITEMS`` (off by default: profiles/r06_dev_wgrads_beside_lm.txt): the ViLT stack's grouped
    weight-gradient launches on the second stream beside the LM backward, joined before the optimizer - same trajectory."""
    spec = VaultSpec.tiny(3, "roberta")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    state = build_state(spec, 0)
    B = 112                                             # (> 16,384 token rows: the small-batch rule of the second stream is off)
    bn = synthetic_batch(spec, B, seed=5, n_classes=3)
    db = {k: torch.from_numpy(v).cuda() for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).cuda()
    res = {}
    for items in (0, 192):
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
        eng.WGRAD_BESIDE_LM_ITEMS = items
        step = TrainStep(eng, learning_rate=1e-4, warmup_ratio=0.0, total_steps=10, assume_full_pixel_mask=True)
        losses = [float(step(db, labels)) for _ in range(3)]
        assert (eng._wgrad_stream is not None) == (items > 0)
        res[items] = (losses, eng.params.p.clone())
    torch.cuda.synchronize()
    la, lb = res[0][0], res[192][0]
    assert abs(la[0] - lb[0]) < 1e-6 and max(abs(a - b) for a, b in zip(la, lb)) < 5e-4, (la, lb)
    d = (res[0][1] - res[192][1]).abs()
    assert float(d.mean()) < 3e-6 and float((d > 2e-5).float().mean()) < 0.03
