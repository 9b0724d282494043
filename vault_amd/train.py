"""Data-parallel fine-tune step around the HIP engine.

Restates the per-step body of the reference's loop, ref: vault/tmsc_utils/trainer.py:353-369
(forward -> CrossEntropy -> zero_grad/backward -> AdamW step -> scheduler step) with
  * the optimizer of trainer.py:244-254 (transformers-4.48 ``AdamW``, ``correct_bias=False`` by
    default, eps 1e-8, decoupled weight decay on all parameters) as ONE fused HIP kernel over the
    flat parameter buffer,
  * the schedule of trainer.py:256-280 (linear warm-up over ``warmup_ratio`` of the steps, then
    linear decay to 0),
  * and - absent from the reference, which is single-device - data parallelism: one process per
    GPU, each rank runs the step on its shard of the batch, gradients are summed with RCCL
    all-reduce (``torch.distributed`` backend "nccl" on ROCm) over xGMI and averaged inside the
    optimizer kernel.  The flat gradient buffer is reduced in contiguous buckets that are launched
    from the backward pass as soon as the stages that write them have been enqueued, on a side
    stream, so the exchange overlaps the remaining backward kernels.

No loss.item() per step: the loss stays on the device (the reference syncs every step at
trainer.py:369); read ``TrainStep.loss`` when needed.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .engine import VaultEngine


def linear_schedule(base_lr: float, step: int, warmup_steps: int, total_steps: int) -> float:
    """lr used by optimizer step number ``step`` (0-based) under get_linear_schedule_with_warmup."""
    if step < warmup_steps:
        return base_lr * float(step) / float(max(1, warmup_steps))
    return base_lr * max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


class GradBuckets:
    """Contiguous ranges of the flat gradient buffer in the order backward finalises them.

    The flat layout is [LM embeddings, LM layers 0.., ViLT embeddings, ViLT layers 0.., head], and
    backward runs head -> ViLT layers (top down) -> ViLT embeddings -> LM layers (top down) -> LM
    embeddings, i.e. strictly descending addresses: after stage ``tag`` every gradient at or above
    ``stage_lo[tag]`` is final.
    """

    def __init__(self, engine: VaultEngine, bucket_mb: float = 64.0):
        P = engine.params
        self.engine = engine
        self.bucket_elems = int(bucket_mb * 1024 * 1024 / 4)
        lo: Dict[str, int] = {}

        def first(prefix_or_names) -> int:
            offs = [P.offsets[n][0] for n in P.trainable
                    if (n.startswith(prefix_or_names) if isinstance(prefix_or_names, str) else n in prefix_or_names)]
            return min(offs) if offs else P.n_train

        spec = engine.spec
        lo["head"] = first(("layernorm.weight", "layernorm.bias", "pooler.dense.weight", "pooler.dense.bias",
                            "classifier.1.weight", "classifier.1.bias"))
        for i in range(spec.vilt.num_hidden_layers):
            lo[f"vilt{i}"] = first(f"encoder.layer.{i}.")
        lo["vilt_embed"] = first("embeddings.")
        if spec.lm is not None and not engine.freeze_lm:
            for i in range(spec.lm.num_hidden_layers):
                lo[f"lm{i}"] = first(f"bert.encoder.layer.{i}.")
            lo["lm_embed"] = 0
        self.stage_lo = lo
        self.last_tag = "lm_embed" if (spec.lm is not None and not engine.freeze_lm) else "vilt_embed"
        self.n = P.n_train


class SparseTable:
    """An embedding table inside the flat gradient buffer whose gradient is exchanged row-sparse: rows x H floats at
    element offset ``lo``; only the rows named by some rank's token ids of the step are non-zero."""

    def __init__(self, lo: int, rows: int, H: int):
        self.lo, self.rows, self.H = int(lo), int(rows), int(H)
        self.hi = self.lo + self.rows * self.H


def _on_bf16_library(fn):
    """The wire format of the exchange is bf16 whatever the engine's operand format (f32 range: no gradient scale to
    respect): these kernels always run on libvault_hip.so."""
    def run(*a):
        with ops.operand_format("bf16"):
            return fn(*a)
    return staticmethod(run)


class ExchangeKernels:
    """Pack / unpack kernels of the exchange on DEVICE tensors (csrc/exchange.hip).  The reducer takes them as an object so
    that its bucket logic can be driven with host tensors over gloo in the CPU tests (which bring their own stand-ins)."""
    narrow = _on_bf16_library(ops.cast_bf16)              # (src_f32, dst_bf16, n)
    widen = _on_bf16_library(ops.widen_bf16)              # (src_bf16, dst_f32, n)
    sum_chunks = _on_bf16_library(ops.sum_chunks_bf16)    # (src_bf16, n_src, chunk, out_bf16)
    rows_union = staticmethod(ops.rows_union)         # (keys, n_keys, V, flags, uniq, count)
    rows_gather = staticmethod(ops.rows_gather)       # (table, idx, n_rows, H, out)
    rows_scatter = staticmethod(ops.rows_scatter)     # (src, idx, n_rows, H, table)


class BucketReducer:
    """Sums each contiguous gradient range over the ranks as soon as it is final.

    ``on_stage(tag)`` is the engine's ``after_layer`` callback, ``finish()`` waits for everything.  Two things keep the
    bytes on the wire down:
      * ``sparse``: the word-embedding table (a fifth of the gradient elements, of which a step touches at most
        world x B x T rows) is exchanged as the union of the touched rows: all-gather of the token ids at the START of the
        step (``begin_step``), sorted union on every rank, gather -> all-reduce of the compact [U, H] rows -> scatter.
        Exact: untouched rows are zero on every rank.
      * ``wire="bf16"``: reduce-scatter + all-gather with bf16 on the wire and f32 accumulation - all-to-all of the
        bf16 chunks (every peer link carries its own chunk: point-to-point xGMI, no ring), f32 sum in rank order,
        rounded to bf16 once, all-gather, widened back into the f32 gradient.  Half the bytes of the f32 all-reduce;
        every rank ends with the same bits.  ``wire="fp32"`` is a plain sum all-reduce.
    Device-agnostic on purpose (CPU tensors + gloo in the tests, HIP tensors + RCCL in production).
    """

    WIRE_PIECE = 1 << 25      # bf16 wire: elements per reduce-scatter / all-gather round (64 MiB of bf16 scratch x 2)
    # One rank (VAULT_FORCE_DP=1 on a single GPU): run the transport's collectives anyway - the tests that exercise the RCCL calls
    # with one process set this; by default the identity sum is skipped, so that the one-rank line of bench.py shows what the
    # data-parallel CODE PATH costs (buckets, events, stage notes, split optimizer pass), not RCCL's degenerate copy kernels
    SINGLE_RANK_COLLECTIVES = False

    def __init__(self, flat_grad: torch.Tensor, stage_lo: Dict[str, int], last_tag: str, bucket_elems: int,
                 dist, group=None, comm_stream=None, compute_device=None, wire: str = "fp32",
                 sparse: Optional[SparseTable] = None, kernels=ExchangeKernels):
        if wire not in ("fp32", "bf16"):
            raise ValueError("wire must be 'fp32' or 'bf16'")
        self.g, self.stage_lo, self.last_tag, self.bucket_elems = flat_grad, stage_lo, last_tag, bucket_elems
        self.dist, self.group, self.comm_stream, self.device = dist, group, comm_stream, compute_device
        self.wire, self.sparse, self.k = wire, sparse, kernels
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.native_a2a = dist.get_backend(group) != "gloo"      # gloo has no all-to-all: emulated with all-gather (tests)
        self.n = flat_grad.numel()
        self.hi = self.n          # everything at or above is launched (top-down frontier)
        self.bottom = 0           # everything below is launched (the lowest stage may finish ahead of the one above it)
        above = [v for t, v in stage_lo.items() if t != last_tag and v > stage_lo.get(last_tag, 0)]
        self.above_last = min(above) if above else self.n      # where the stage above the lowest one starts
        self.min_seen = self.n    # lowest stage_lo among the stages noted in this step
        self.done: List = []      # per launch: event on the communication stream (None on the host)
        self.launched: List[Tuple[int, int]] = []
        self.wire_bytes = 0       # bytes this rank put on the wire in the current step (diagnostics)
        self._scratch: Dict[str, torch.Tensor] = {}
        self._union_ready = None
        self._n_keys = 0
        self._key_capacity = None   # per-rank key count agreed on in the first step (begin_step)
        self._keys_given = None
        self.union_waits = 0        # steps in which the host had to WAIT for the union (diagnostics: expected 0)
        # VAULT_DP_CHECK_SPARSE=N (debug): in the first N steps the row-sparse result is compared with a dense f32 all-reduce of
        # the whole table (a host sync per step): a gradient row that some rank holds outside the union of the step's token
        # ids - a second source of gradient on the table - would otherwise stay un-reduced without an error
        self._check_sparse_left = int(os.environ.get("VAULT_DP_CHECK_SPARSE", "0") or 0)
        self.sparse_checks = 0

    # ---- plumbing -------------------------------------------------------------------------------------------
    def _buf(self, name: str, n: int, dtype) -> torch.Tensor:
        t = self._scratch.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.zeros(n, dtype=dtype, device=self.g.device)
            self._scratch[name] = t
        return t[:n]

    def _on_comm_stream(self, fn):
        """Run ``fn`` with the communication stream current, ordered behind everything enqueued on the compute stream so
        far; returns the event that marks its end (None on the host).  Collectives inside ``fn`` are waited for on the
        stream (``Work.wait()`` blocks the stream, not the host, with RCCL; it blocks the host with gloo)."""
        if self.comm_stream is None:
            fn()
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            fn()
            end = torch.cuda.Event()
            end.record(self.comm_stream)
        return end

    def _all_to_all(self, recv: torch.Tensor, send: torch.Tensor, chunk: int):
        if self.native_a2a:
            self.dist.all_to_all_single(recv, send, group=self.group, async_op=True).wait()
            return
        parts = [torch.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(parts, send, group=self.group)
        for k in range(self.world):
            recv[k * chunk:(k + 1) * chunk].copy_(parts[k][self.rank * chunk:(self.rank + 1) * chunk])

    def _all_gather(self, full: torch.Tensor, part: torch.Tensor):
        if self.native_a2a:
            self.dist.all_gather_into_tensor(full, part, group=self.group, async_op=True).wait()
            return
        parts = [torch.empty_like(part) for _ in range(self.world)]
        self.dist.all_gather(parts, part, group=self.group)
        n = part.numel()
        for k in range(self.world):
            full[k * n:(k + 1) * n].copy_(parts[k])

    def _sum_over_ranks(self, view: torch.Tensor):
        """view (contiguous f32, 4-element aligned) <- its sum over the ranks, in the configured wire format."""
        n = view.numel()
        if n == 0 or (self.world == 1 and not self.SINGLE_RANK_COLLECTIVES):
            return      # (one rank - VAULT_FORCE_DP on a single GPU: the sum is the identity, nothing goes on a wire)
        if self.wire == "fp32":
            self.dist.all_reduce(view, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
            self.wire_bytes += 2 * 4 * n * (self.world - 1) // self.world
            return
        W = self.world
        for p0 in range(0, n, self.WIRE_PIECE):
            m = min(self.WIRE_PIECE, n - p0)
            chunk = -(-m // (8 * W)) * 8                      # elements per rank, 16-byte multiples
            send = self._buf("send", chunk * W, torch.bfloat16)
            recv = self._buf("recv", chunk * W, torch.bfloat16)
            red = self._buf("red", chunk, torch.bfloat16)
            if chunk * W > m:
                send[m - (m % 4):].zero_()                    # the padding behind the range (and a ragged tail) adds zeros
            m4 = m - (m % 4)
            if m4:
                self.k.narrow(view[p0:p0 + m4], send, m4)
            if m % 4:                                         # (never for the engine's 64-aligned ranges)
                send[m4:m].copy_(view[p0 + m4:p0 + m])
            self._all_to_all(recv, send, chunk)               # recv[k] = rank k's bf16 image of MY chunk
            self.k.sum_chunks(recv, W, chunk, red)            # f32 accumulation in rank order, one rounding
            self._all_gather(send, red)                       # (the send buffer is free again: it takes the result)
            if m4:
                self.k.widen(send, view[p0:p0 + m4], m4)
            if m % 4:
                view[p0 + m4:p0 + m].copy_(send[m4:m])
            self.wire_bytes += 2 * 2 * chunk * (W - 1)

    # ---- row-sparse table -----------------------------------------------------------------------------------
    def begin_step(self, keys: Optional[torch.Tensor]):
        """Start of a step: ``keys`` = this rank's token ids (int64, any shape; None: this step touches no row of the table,
        on EVERY rank alike).  Their all-gather and the union run on the communication stream at once; the host reads the
        union's size only when the table's gradient is final (``_exchange_table``).

        The all-gather needs the SAME key count on every rank and every rank to agree on ids-or-none: the first step
        exchanges (count, has-keys) between the ranks and raises on a mismatch (one host synchronisation, once).  The count
        of that step - the first step's B x T - is the CAPACITY afterwards: a later, smaller batch is padded with -1 (ignored
        by the union).  Every later step carries a two-word header (token count, ids given) per rank in the same all-gather:
        a rank with more ids than the capacity, or one that switched between ids and ``inputs_embeds``, makes EVERY rank raise
        when the table is exchanged (``_check_headers``) - a decision taken on one rank only would leave the others waiting
        in their next collective until it times out.

        The bucket frontier starts over here as well: a step that raised from inside the exchange (the header / id checks run
        under ``_launch``) never reached ``finish()``, and a frontier left at its position would make ``on_stage`` skip every
        upper range of the next step without an error (the replicas would diverge silently)."""
        self.reset(wait=True)
        self.wire_bytes = 0
        self._union_ready, self._n_keys = None, 0
        if self.sparse is None:
            return
        n_local = 0 if keys is None else int(keys.numel())
        if self._key_capacity is None:
            hdr = torch.tensor([n_local, 0 if keys is None else 1], dtype=torch.int64, device=self.g.device)
            allh = [torch.zeros_like(hdr) for _ in range(self.world)]
            self.dist.all_gather(allh, hdr, group=self.group)
            rows = [tuple(int(v) for v in h.cpu()) for h in allh]
            if any(r != rows[0] for r in rows):
                raise RuntimeError(f"row-sparse embedding exchange: the ranks disagree on (token count, ids given): {rows} - "
                                   "every rank must step the same per-rank batch shape (or pass sparse_embedding=False)")
            self._key_capacity, self._keys_given = n_local, keys is not None
        n = self._key_capacity
        W, V = self.world, self.sparse.rows
        # this rank's slice of the all-gather: [token count, ids given, keys (capacity n; short batches padded with -1, a
        # batch beyond the capacity truncated - its header makes every rank raise)]
        mine = self._buf("keys_local", n + 2, torch.int64)
        mine[0], mine[1] = n_local, (0 if keys is None else 1)
        if n:
            m = min(n_local, n)
            if m:
                mine[2:2 + m].copy_(keys.reshape(-1)[:m])
            if m < n:
                mine[2 + m:].fill_(-1)
        allk = self._buf("keys", (n + 2) * W, torch.int64)
        hdrs = self._buf("headers", 2 * W, torch.int64)
        host_h = self._scratch.get("headers_host")
        if host_h is None:
            host_h = torch.zeros(2 * W, dtype=torch.int64)
            if self.comm_stream is not None:
                host_h = host_h.pin_memory()
            self._scratch["headers_host"] = host_h
        if n:
            flags = self._buf("flags", V, torch.int32)
            uniq = self._buf("uniq", min(V, n * W), torch.int64)
            cnt = self._buf("count", 2, torch.int32)               # [distinct rows, keys outside the table]
            host = self._scratch.get("count_host")
            if host is None:
                host = torch.zeros(2, dtype=torch.int32)
                if self.comm_stream is not None:
                    host = host.pin_memory()
                self._scratch["count_host"] = host

        def go():
            self._all_gather(allk, mine)
            slots = allk.view(W, n + 2)[:, :2]
            hdrs.view(W, 2).copy_(slots)
            host_h.copy_(hdrs, non_blocking=True)
            if n:
                slots.fill_(-1)                                   # (the header words are no keys: padding for the union)
                self.k.rows_union(allk, (n + 2) * W, V, flags, uniq, cnt)
                host.copy_(cnt, non_blocking=True)

        self._union_ready = self._on_comm_stream(go)
        self._n_keys = n * W if (keys is not None and self._keys_given) else 0
        self.wire_bytes += 8 * (n + 2) * (W - 1)

    def _check_headers(self):
        """Every rank reads the same (token count, ids given) words of all ranks and takes the same decision."""
        if self._union_ready is not None and not self._union_ready.query():
            # (enqueued at the start of the step: normally long done - the host only blocks, and counts it, when the
            #  first collective of the step was slow)
            self.union_waits += 1
            self._union_ready.synchronize()
        h = self._scratch["headers_host"].view(self.world, 2).tolist()
        over = [(r, int(c)) for r, (c, _) in enumerate(h) if c > self._key_capacity]
        if over:
            raise RuntimeError(f"row-sparse embedding exchange: rank(s) {over} stepped more token ids than the "
                               f"{self._key_capacity} per rank agreed on in the first step (the capacity is the first step's "
                               "B x T: start with the largest batch, or pass sparse_embedding=False)")
        switched = [r for r, (_, g_) in enumerate(h) if bool(g_) != self._keys_given]
        if switched:
            raise RuntimeError(f"row-sparse embedding exchange: rank(s) {switched} switched between token ids and inputs_embeds "
                               "(construct the TrainStep with sparse_embedding=False)")

    def _exchange_table(self):
        sp = self.sparse
        self._check_headers()
        U = 0
        if self._n_keys:
            U, bad = int(self._scratch["count_host"][0]), int(self._scratch["count_host"][1])
            if bad:
                raise RuntimeError(f"row-sparse embedding exchange: {bad} token ids lie outside the table's {sp.rows} rows - their "
                                   "gradient rows would be left un-reduced (replicas would diverge)")
        table = self.g[sp.lo:sp.hi]
        dense = None
        if self._check_sparse_left > 0:
            # (debug; VAULT_DP_CHECK_SPARSE must be set identically on every rank - the comparison is a collective.  It runs in
            #  the steps without any touched row too - inputs_embeds steps, an empty union: exactly the steps in which a second
            #  source of gradient on the table would go un-reduced altogether)
            dense = table.clone()
            self.dist.all_reduce(dense, op=self.dist.ReduceOp.SUM, group=self.group)
        if U > 0:
            uniq = self._scratch["uniq"]
            compact = self._buf("rows", min(sp.rows, self._n_keys) * sp.H, torch.float32)[:U * sp.H]
            self.k.rows_gather(table, uniq, U, sp.H, compact)
            self._sum_over_ranks(compact)
            self.k.rows_scatter(compact, uniq, U, sp.H, table)
        if dense is not None:
            self._check_sparse_left -= 1
            self.sparse_checks += 1
            both = torch.stack([(table - dense).abs().max(), dense.abs().max()])
            self.dist.all_reduce(both, op=self.dist.ReduceOp.MAX, group=self.group)   # (every rank takes the same decision)
            err, ref = float(both[0]), float(both[1])
            tol = (1e-2 if self.wire == "bf16" else 1e-5) * ref + 1e-30
            if not err <= tol:
                raise RuntimeError(f"row-sparse embedding exchange differs from the dense all-reduce of the table: max |diff| "
                                   f"{err:.3e} against max |gradient| {ref:.3e} (VAULT_DP_CHECK_SPARSE) - some rank holds "
                                   "gradient rows outside the union of this step's token ids")

    # ---- bucket logic ---------------------------------------------------------------------------------------
    def reset(self, wait: bool = False):
        """Forget the ranges launched so far: top-down frontier at the end of the buffer, nothing launched, no events.
        ``wait``: the compute stream first waits for exchanges still in flight (a step that raised left them behind: whatever
        the caller does to the gradient buffer next must not race with them)."""
        if wait:
            self._wait(self.done)
        self.hi, self.bottom, self.min_seen = self.n, 0, self.n
        self.done.clear()
        self.launched.clear()

    def on_stage(self, tag: str):
        lo = self.stage_lo.get(tag)
        if lo is None:
            return
        early = tag == self.last_tag and self.min_seen > self.above_last and self.bottom == 0 and self.above_last < self.hi
        self.min_seen = min(self.min_seen, lo)
        if early:
            # the lowest stage (the embedding tables) became final BEFORE the stage above it was noted - the engine runs the
            # embedding backward ahead of the deferred weight gradients of the last group of layers in data-parallel
            # steps, so that this exchange runs under them: reduce [0, above_last) now
            self._launch(0, self.above_last)
            self.bottom = self.above_last
            return
        final = lo <= self.bottom
        if final:
            lo = self.bottom
        if self.hi <= lo or (not final and (self.hi - lo) < self.bucket_elems):
            return
        self._launch(lo, self.hi)
        self.hi = lo

    def _launch(self, lo: int, hi: int):
        sp = self.sparse

        def go():
            if sp is not None and lo < sp.hi and hi > sp.lo:
                self._sum_over_ranks(self.g[lo:max(lo, sp.lo)])
                self._exchange_table()
                self._sum_over_ranks(self.g[min(hi, sp.hi):hi])
            else:
                self._sum_over_ranks(self.g[lo:hi])

        self.done.append(self._on_comm_stream(go))
        self.launched.append((lo, hi))

    def _wait(self, events):
        for ev in events:
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)

    def finish_upper(self) -> int:
        """Make the compute stream wait for every launched range except the LAST one (the lowest addresses: launched when
        backward ends, nothing is left to hide it behind) and return its upper bound x: gradients in [x, n) are final
        and reduced, so the optimizer can already run on them while [0, x) is still on the wire."""
        if len(self.done) < 2:
            return self.n if not self.done else self.launched[-1][1]
        self._wait(self.done[:-1])
        del self.done[:-1]
        return self.launched[-1][1]

    def finish(self):
        self._wait(self.done)
        missing = (self.bottom, self.hi) if self.hi != self.bottom else None
        self.reset()      # (first: the reducer stays usable after the error)
        if missing is not None:
            raise RuntimeError("gradient range [%d, %d) was never reduced" % missing)


class TrainStep:
    def __init__(self, engine: VaultEngine, learning_rate: float = 2e-5, adam_beta1: float = 0.9,
                 adam_beta2: float = 0.999, adam_epsilon: float = 1e-8, weight_decay: float = 0.0,
                 correct_bias: bool = False, warmup_ratio: float = 0.1, total_steps: int = 1000,
                 process_group=None, bucket_mb: float = 64.0, constant_lr: bool = False, use_tape: bool = True,
                 assume_full_pixel_mask: bool = False, wire: Optional[str] = None, sparse_embedding: Optional[bool] = None,
                 precise_forward: bool = False):
        self.engine = engine
        self.lr, self.b1, self.b2, self.eps, self.wd = learning_rate, adam_beta1, adam_beta2, adam_epsilon, weight_decay
        self.correct_bias = correct_bias
        self.total_steps = int(total_steps)
        self.warmup_steps = int(warmup_ratio * self.total_steps)
        self.constant_lr = constant_lr
        self.step_idx = 0
        self.loss: Optional[torch.Tensor] = None
        self.use_tape = use_tape
        # split-bf16 ("bf16x3") forward GEMMs: logits / loss of the training step at fp32 class (inside the 1e-3 of the reference),
        # bf16 backward as in the fast mode; ~3x the forward GEMM time
        self.precise_forward = bool(precise_forward)
        self._tape = None
        self._tape_key = None
        self._tape_ws = None
        self._zero_mask = None      # see _build_zero_mask
        # True: the caller vouches that every pixel_mask is all ones on the square pre-training canvas (no device
        # sync per step); False: the mask is checked each step and padded batches take the general image path
        self.assume_full_pixel_mask = assume_full_pixel_mask
        self._loss_buf = None
        self.world = 1
        self.wire = None
        self.reducer: Optional[BucketReducer] = None
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.world = dist.get_world_size(process_group)
        engine.dp_world = self.world
        # VAULT_FORCE_DP=1 exercises the bucketed all-reduce path even with a single rank (debug/testing)
        if self.world > 1:
            # every rank draws its own dropout masks: the keep/drop hash takes (seed, stream, LOCAL element index), so
            # equal seeds would give sample j of every rank the same pattern
            engine.drop_seed = (engine.drop_seed + dist.get_rank(process_group) * 0x9E3779B1) & 0xFFFFFFFF
            # the gradient all-reduce (RCCL kernels on a side stream) shares the CUs with the backward GEMMs: hand the
            # GEMM tiles out dynamically so that a CU held by the collective does not stall a static tile walk.
            # (The tape records the argument structs: this must be set before the first step is recorded.)
            ops.GEMM_SCHED = 3
            # (the deferred, batched weight gradients already come in groups of 6 layers - engine.LM_WGRAD_GROUP - so the
            # upper group's gradient range is all-reduced under the backward of the layers below it)
        if self.world > 1 or (os.environ.get("VAULT_FORCE_DP") == "1" and dist.is_available() and dist.is_initialized()):
            b = GradBuckets(engine, bucket_mb)
            # wire format of the dense ranges ("fp32": sum all-reduce; "bf16": reduce-scatter + all-gather with bf16 on the
            # wire and f32 accumulation) and the row-sparse exchange of the word-embedding table: BucketReducer
            self.wire = wire or "fp32"
            if sparse_embedding is None:
                sparse_embedding = True
            self._sparse_name = None
            sparse = None
            if sparse_embedding:
                P = engine.params
                name = ("bert.embeddings.word_embeddings.weight" if engine.spec.lm is not None
                        else "embeddings.text_embeddings.word_embeddings.weight")
                # (a tied MLM decoder puts a DENSE gradient on ViLT's own table)
                if P.has_grad(name) and not (engine.spec.lm is None and engine.spec.head == "mlm"):
                    o, shp = P.offsets[name]
                    sparse = SparseTable(o, shp[0], shp[1])
                    self._sparse_name = name
            self.reducer = BucketReducer(engine.params.g[:engine.params.n_train], b.stage_lo, b.last_tag, b.bucket_elems, dist,
                                         process_group, torch.cuda.Stream(device=engine.device), engine.device,
                                         wire=self.wire, sparse=sparse)

    def current_lr(self) -> float:
        if self.constant_lr:
            return self.lr
        return linear_schedule(self.lr, self.step_idx, self.warmup_steps, self.total_steps)

    def __call__(self, batch: Dict[str, torch.Tensor], labels: torch.Tensor) -> torch.Tensor:
        """One optimisation step.  The first call for a (B, T) shape runs eagerly and records the call
        tape (ops.Tape); later calls copy the batch into the persistent input buffers and replay it."""
        eng = self.engine
        B, T = batch["input_ids"].shape
        if self.reducer:
            # (a step that raised inside the exchange never reached finish(): its frontier and its events are dropped here,
            #  before anything below touches the gradient buffer)
            self.reducer.reset(wait=True)
        if eng._g_dirty:
            # gradients of an API-level backward (loss.backward() through the module) are still in the flat buffer: the fused step
            # starts from zeros (it STORES its un-split weight-gradient tiles and lets AdamW clear the buffer afterwards)
            eng.zero_grad()
        with torch.cuda.device(eng.device), ops.operand_format(eng.half):
            # staging decides the image geometry (square all-valid canvas, or a padded batch of differently sized
            # images: host-side patch selection); every geometry has its own workspace, hence its own tape
            ws = eng.stage_inputs(batch, True, labels, validate=not self.assume_full_pixel_mask)
            # what the recorded launches bake in besides the buffers of the (B, T, geometry) workspace: whether token
            # types were given, the launch stream, the GEMM scheduling mode and the forward number format
            key = ws["key"] + (ws["tt"] is None, torch.cuda.current_stream().cuda_stream, ops.GEMM_SCHED,
                               bool(eng.fp8_forward), labels.dtype.is_floating_point, bool(ws.get("patches_in")), self.precise_forward,
                               eng.half, eng.grad_scale, bool(ws.get("patches_in") and ws.get("patch_adopted")))
            # from here to the enqueued AdamW (which clears it) the flat gradient buffer holds this step's partial gradients: an
            # exception in between (the reducer's checks raise and stay usable) must not leave them for the next step, which
            # STORES its un-split weight-gradient tiles but atomically ADDS bias / LayerNorm / embedding / split-K gradients
            if eng._g_stale_key is not None and eng._g_stale_key != key:
                # the previous fused step (another shape / mode: another tape) left the ranges ITS successor would have stored
                # un-zeroed; this step may accumulate there
                eng.zero_grad()
            eng._g_dirty = True        # (after the zeroing above, which clears the flag)
            if self.reducer:
                # the token ids of every rank name the rows of the word-embedding table this step touches: their
                # all-gather + union start now on the communication stream (inputs_embeds: no row is touched)
                self.reducer.begin_step(ws["ids"] if ws.get("txt_embeds") is None else None)
            if self.use_tape and self._tape is not None and self._tape_key == key and self._tape_ws is ws:
                eng.drop_seed = (eng.drop_seed + 1) & 0xFFFFFFFF
                if self._tape.rebinds:       # (the adopted pixel_patches tensor of THIS step: VaultEngine.adopt_pixel_patches)
                    self._tape.rebind(ws["apatch_in"].data_ptr())
                self._tape.replay(seed=eng.drop_seed)
            else:
                eng.drop_seed = (eng.drop_seed + 1) & 0xFFFFFFFF
                tape = ops.start_tape() if self.use_tape else None
                adopted = bool(ws.get("patches_in") and ws.get("patch_adopted"))
                if tape is not None and adopted:
                    ap_in = ws["apatch_in"]
                    tape.rebind_range = (ap_in.data_ptr(), ap_in.data_ptr() + ap_in.numel() * ap_in.element_size())
                try:
                    out = eng.forward_staged(ws, need_hidden=False, loss_scale=1.0 / B, precise=self.precise_forward)
                    # gradients are zero here: the fused optimizer clears them after use (they start at 0)
                    eng._backward(1.0 / B, None, None, None, self.reducer.on_stage if self.reducer else None, grads_zero=True)
                finally:
                    if self.use_tape:
                        ops.stop_tape()
                if tape is not None and adopted and len(tape.rebinds) < 2:
                    raise RuntimeError("adopted pixel_patches: the recorded step does not show the operand's two uses "
                                       "(patch-embedding GEMM, its weight gradient)")
                self._tape, self._tape_key, self._tape_ws, self._loss_buf = tape, key, ws, out["loss"]
                self._zero_mask = self._build_zero_mask(eng._stored_ranges)
            if self.reducer:
                # optimizer on the already reduced upper part of the flat buffer while the last bucket (LM / ViLT
                # embeddings: the lowest addresses) is still being all-reduced, then on the rest
                x = self.reducer.finish_upper()
                if 0 < x < eng.params.n_train:
                    self.optimizer_step(lo=x, hi=eng.params.n_train, advance=False, zero_mask=self._zero_mask)
                    self.reducer.finish()
                    self.optimizer_step(lo=0, hi=x, zero_mask=self._zero_mask)
                else:
                    self.reducer.finish()
                    self.optimizer_step(zero_mask=self._zero_mask)
            else:
                self.optimizer_step(zero_mask=self._zero_mask)
            eng._g_dirty = False       # (the optimizer pass over [0, n_train) is enqueued: it zeroes what it reads ...)
            eng._g_stale_key = key if self._zero_mask is not None else None     # (... but for the ranges the next step stores)
        self.loss = self._loss_buf
        return self.loss

    def _build_zero_mask(self, stored):
        """One byte per 64 gradient elements: 0 over the weight-gradient matrices the recorded backward writes with stores only
        (engine._stored_ranges: every tensor is 64-element aligned) - the optimizer skips their zeroing (4 of its 34 B/param);
        None when nothing is stored (small batches: split-K launches accumulate) or with VAULT_ADAMW_ZERO_MASK=0."""
        if not stored or os.environ.get("VAULT_ADAMW_ZERO_MASK", "1") == "0":
            return None
        key = tuple(sorted(stored))
        if getattr(self, "_zero_mask_key", None) == key:
            return self._zero_mask
        n64 = self.engine.params.n_train // 64
        m = np.ones(n64, np.uint8)
        for o, n in key:
            assert o % 64 == 0 and n % 64 == 0
            m[o // 64:(o + n) // 64] = 0
        self._zero_mask_key = key
        return torch.from_numpy(m).to(self.engine.device)

    def optimizer_step(self, lo: int = 0, hi: Optional[int] = None, advance: bool = True,
                       zero_mask: Optional[torch.Tensor] = None):
        """Fused HF-AdamW over elements [lo, hi) of the flat parameter buffer (default: all trainable ones);
        ``advance=False`` leaves the step counter alone (first part of a split update).  ``zero_mask`` (one byte per 64
        elements of the WHOLE buffer, 0 = leave the gradient un-zeroed) is the fused step's own: a caller that steps the
        optimizer by hand (gradient accumulation, a manual loop) gets every gradient it read cleared."""
        eng = self.engine
        P = eng.params
        hi = P.n_train if hi is None else hi
        t = self.step_idx + 1
        bc = 1.0
        if self.correct_bias:
            bc = math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        with torch.cuda.device(eng.device), ops.operand_format(eng.half):
            if hi > lo:
                # (the gradients carry the operand format's power-of-two scale - fp16: engine.grad_scale - and the sum
                #  over the ranks: both are divided out here)
                zm = zero_mask
                if zm is not None and (lo % 64 or (hi - lo) % 64):
                    zm = None          # (never for the engine's 64-aligned stage boundaries)
                ops.adamw_step(P.p[lo:hi], P.g[lo:hi], P.m[lo:hi], P.v[lo:hi], P.pb[lo:hi], hi - lo, self.current_lr(),
                               self.b1, self.b2, self.eps, self.wd, bias_corr_factor=bc,
                               grad_scale=1.0 / (self.world * eng.grad_scale), zero_grad=True,
                               zero_mask=None if zm is None else zm[lo // 64:hi // 64])
        if advance:
            with torch.cuda.device(eng.device):
                P.refresh_transposed()   # W^T shadow of the data-gradient GEMMs, from the bf16 shadow just written
            P._pb3_fresh = False   # the split-bf16 (precise inference) shadow is stale now
            self.step_idx += 1


def evaluation_metrics(eval_true, eval_preds) -> Dict[str, float]:
    """``eval_accuracy`` and ``macro_f1_score`` as the reference computes them (ref: vault/tmsc_utils/trainer.py:513-549:
    sklearn ``precision_recall_fscore_support(average="macro", zero_division=0)`` over the union of the labels that
    occur in truth or predictions, and the plain match rate)."""
    t = np.asarray(list(eval_true)).reshape(-1)
    p = np.asarray(list(eval_preds)).reshape(-1)
    f1s = []
    for c in np.union1d(t, p):
        tp = float(np.sum((p == c) & (t == c)))
        fp = float(np.sum((p == c) & (t != c)))
        fn = float(np.sum((p != c) & (t == c)))
        prec = tp / (tp + fp) if tp + fp > 0 else 0.0
        rec = tp / (tp + fn) if tp + fn > 0 else 0.0
        f1s.append(2 * prec * rec / (prec + rec) if prec + rec > 0 else 0.0)
    return dict(eval_accuracy=float(np.mean(p == t)) if t.size else 0.0,
                macro_f1_score=float(np.mean(f1s)) if f1s else 0.0)


def evaluate(engine: VaultEngine, batches) -> Dict[str, float]:
    """The reference's evaluation pass (ref: vault/tmsc_utils/trainer.py:429-484): eval-mode forward per batch (all
    dropouts off), mean cross-entropy weighted by batch size, argmax predictions, accuracy and macro-F1.
    ``batches`` yields ``(batch_dict_on_device, labels_on_device)``."""
    preds, true = [], []
    loss_sum, n = 0.0, 0
    for batch, labels in batches:
        out = engine.forward(batch, train=False, labels=labels, need_hidden=False)
        B = int(labels.shape[0])
        loss_sum += float(out["loss"].item()) * B
        n += B
        preds.extend(out["logits"].view(B, -1).argmax(dim=-1).tolist())
        true.extend(labels.view(-1).tolist())
    res = evaluation_metrics(true, preds)
    res["eval_loss"] = loss_sum / max(n, 1)
    return res
