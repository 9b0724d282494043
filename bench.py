#!/usr/bin/env python
"""Headline benchmark: train samples/sec (image+text pairs) of the ViLT-B32 + BERTweet-base
fine-tune step (forward + backward + fused AdamW, bf16 MFMA compute / fp32 master weights), per-GPU
batch 256, synthetic 384x384 images + 40-token captions, on N GPUs of one node (data parallel,
RCCL all-reduce of gradients).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").  `roofline` is the dominant kernel of the
step by GPU time (rocprofv3 --stats, profiles/): the weight-gradient instantiation of the bf16 MFMA ring
GEMM, gemm256_kernel<1,1,EPI_F32_ATOMIC,4>, timed live with events on the launch stream around all of its
launches (ViLT layers + patch projection) in every 4th timed step; `roofline_ffn1` is the same for the FFN-in forward instantiation (the
largest single GEMM call site); `step_mfma_frac` is the whole step against the 2.5 PFLOP/s dense bf16 peak
with BASELINE.md's 120.67 GFLOP/sample.  `cpu_baseline` times the CPU oracle (plain fp32 torch
restatement of the reference path) on the host cores, on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vault_amd.engine import VaultEngine  # noqa: E402
from vault_amd.spec import LMSpec, VaultSpec, ViltSpec, build_state, synthetic_batch  # noqa: E402
from vault_amd.train import TrainStep  # noqa: E402

FLOP_PER_SAMPLE_TRAIN = 120.67e9      # BASELINE.md §2 (fwd 40.22 GF x 3)
PEAK_BF16 = 2.5e15                    # MI355X dense bf16 MFMA, MI355X_MICROARCH.md


def cpu_baseline(spec, seconds_budget: float = 30.0, batch: int = 8):
    """Oracle (fp32 torch restatement of the reference path) fwd + bwd + HF-AdamW on the host cores.

    Bounded sample: at most `seconds_budget` of timed CPU work at batch 8 (one un-timed step first:
    allocator / thread-pool warm-up).  32 threads at most: eager torch CPU ops on this model stop scaling
    (and collapse from oversubscription) far below the 256 hardware threads of the GPU box."""
    from oracle import vault_oracle as O
    cores = max(1, min(32, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    state = build_state(spec, 0)
    P = O.to_torch_state(state, requires_grad=True)
    names = [k for k in P]
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v2 = {k: torch.zeros_like(v) for k, v in P.items()}
    bn = synthetic_batch(spec, batch, seed=99, n_classes=3)
    tb = O.torch_batch(bn)

    def step(t):
        for p in P.values():
            p.grad = None
        loss, _ = O.vault_loss(P, spec, tb)
        loss.backward()
        with torch.no_grad():
            for k in names:
                g = P[k].grad
                if g is None:
                    continue
                O.hf_adamw_step(P[k], g, m[k], v2[k], 2e-5, t)
        return float(loss.detach())

    t0 = time.time()
    step(1)  # warm-up (allocator, thread pools), also a guard: a pathological host aborts the baseline
    warm = time.time() - t0
    if warm > 4 * seconds_budget:
        return {"value": round(batch / warm, 4), "unit": "samples/s", "cores": cores, "kind": "port",
                "sample": f"warm-up step only (fwd+bwd+AdamW fp32, batch {batch}): {warm:.1f} s - host too slow to time more"}
    t0 = time.time()
    n = 0
    while True:
        step(n + 2)
        n += 1
        if time.time() - t0 > seconds_budget * 0.5 or n >= 16:
            break
    dt = time.time() - t0
    return {"value": round(batch * n / dt, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} full fine-tune steps (fwd+bwd+AdamW, fp32) of the CPU oracle at batch {batch}, "
                      f"same model shape and synthetic inputs; {dt:.1f} s on {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--freeze-lm", action="store_true")
    ap.add_argument("--fp8-forward", action="store_true",
                    help="BASELINE config 5: MXFP8 forward Linear GEMMs (block-scaled fp8 MFMA), bf16 backward")
    ap.add_argument("--lm", default="bertweet", choices=["bertweet", "bert-base-uncased"])
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    pg = None
    if world > 1 or os.environ.get("VAULT_FORCE_DP") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    lm = LMSpec.bertweet_base() if args.lm == "bertweet" else LMSpec.bert_base_uncased()
    spec = VaultSpec(vilt=ViltSpec(), lm=lm, n_classes=3)
    eng = VaultEngine(spec, dev, seed=0, freeze_lm=args.freeze_lm, classifier_dropout=0.1, fp8_forward=args.fp8_forward)
    total = args.steps + args.warmup
    stepper = TrainStep(eng, learning_rate=2e-5, warmup_ratio=0.1, total_steps=max(total, 10), process_group=pg,
                       assume_full_pixel_mask=True)   # synthetic 384x384 images, all-ones masks: no per-step mask check

    B = args.batch
    bn = synthetic_batch(spec, B, seed=1234 + rank, n_classes=3)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in bn.items() if k != "labels"}
    labels = torch.from_numpy(bn["labels"]).to(dev)
    # inputs are resident in HBM before the timed region: put them where a data loader's host->device copy would
    # land, the engine's own staging buffers (no device-to-device re-copy of the pixels per step)
    stage = eng.input_buffers(B, batch["input_ids"].shape[1], True)
    for k in ("input_ids", "pixel_values"):
        stage[k].copy_(batch[k])
        batch[k] = stage[k]
    stage["labels"].copy_(labels)
    labels = stage["labels"]

    # live timing of the dominant kernels: event pairs on the launch stream around their launches, in every 4th
    # timed step (an event pair costs ~1-2 us of stream time: sampling keeps `value` undisturbed)
    evs = {"wgrad": [], "ffn1": []}
    eng.profile_events = None

    def sync_all():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        stepper(batch, labels)
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        eng.profile_events = evs if (k % 4 == 0) else None
        stepper(batch, labels)
    sync_all()
    dt = time.perf_counter() - t0
    eng.profile_events = None
    loss = float(stepper.loss.item())

    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        sps = B * world * args.steps / dt
        v = spec.vilt
        M = B * (40 + 1 + v.num_patches)
        flop_per_sample = FLOP_PER_SAMPLE_TRAIN if not args.freeze_lm else 106.96e9

        def roof(site, kernel, pmc_file):
            """achieved = algorithmic FLOPs of all timed launches / the sum of their durations"""
            ms = [s_.elapsed_time(e_) for s_, e_, _ in evs[site]]
            fl = [f for _, _, f in evs[site]]
            if not ms:
                return {"bound": "mfma", "kernel": kernel, "achieved": None, "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": None, "traffic": None, "launches_timed": 0}
            achieved = sum(fl) / (sum(ms) * 1e-3) / 1e12
            # HBM/fabric bytes of that kernel from PMC counters (separate rocprofv3 --pmc passes, committed under
            # profiles/: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE), average per launch; only for the profiled batch
            traffic = None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
                if pm.get("batch") == B and pm.get("lm") == args.lm and not args.freeze_lm:
                    traffic = pm["traffic_bytes_per_launch_avg"]
            except Exception:
                traffic = None
            return {"bound": "mfma", "kernel": kernel, "flop_per_launch_avg": round(float(np.mean(fl)) / 1e9, 2),
                    "achieved": round(achieved, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(achieved / 2500.0, 4),
                    "traffic": traffic, "launches_timed": len(ms), "avg_launch_ms": round(float(np.mean(ms)), 4)}

        r_wgrad = roof("wgrad", "gemm256_kernel<1,1,5,4> (A[K][M]^T B[K][N], EPI_F32_ATOMIC): the weight-gradient GEMMs, "
                                f"dW[N x K] += dY[tokens][N]^T X[tokens][K] - batched launches of 6 layers each ({M} ViLT tokens / "
                                f"{B * 40} LM tokens per layer; FFN-out, FFN-in, attention-out, QKV) and the patch projection",
                       "r01_pmc_gemm_wgrad.json")
        r_ffn1 = roof("ffn1", "gemm256_kernel<0,0,1,4> (EPI_BF16_GELU): FFN-in forward, ViLT "
                              f"[{M}x{v.intermediate_size}x{v.hidden_size}] + LM [{B * 40}x{v.intermediate_size}x{v.hidden_size}]",
                      "r01_pmc_gemm_ffn1.json")
        out = {
            "metric": "train samples/sec (img+text pairs) ViLT-B32+BERTweet, bs256, 1/2/4/8 MI355X",
            "value": round(sps, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "mxfp8 forward GEMMs / bf16 backward" if args.fp8_forward else "bf16",
            "data": "synthetic",
            "config": {"workload": f"ViLT-B32 + {args.lm} fine-tune step (fwd+bwd+AdamW), per-GPU batch {B}, "
                                   f"40 text tokens + 384x384 image (185-token fused sequence), "
                                   f"{'frozen LM' if args.freeze_lm else 'all weights trained'}"
                                   f"{', MXFP8 forward Linears' if args.fp8_forward else ''}",
                       "global_batch": B * world, "seq_len": 185, "parallelism": f"dp{world}"},
            "roofline": r_wgrad, "roofline_ffn1": r_ffn1,
            "step_mfma_frac": round(sps / world * flop_per_sample / PEAK_BF16, 4),
            "final_loss": round(loss, 5),
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(spec)
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"error": repr(e)}
        # librccl prints its version banner through C stdio, which a redirected stdout only flushes at exit - after
        # this line.  Flush it first: the JSON line must be the last thing on stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # pragma: no cover
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
