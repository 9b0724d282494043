"""Host-side driver of the stacked BERT -> ViLT path: owns the flat parameter/gradient buffers and
the activation workspace in HBM and enqueues the HIP kernels of libvault_hip.so in order.

PyTorch is used for device memory, streams and (in train.py) torch.distributed only; all arithmetic
of the path runs in the hand-written kernels.  Mirrors ref: vault/models/vault/model.py:151-218
(LM -> inputs_embeds -> ViLT) and, for backward, what autograd does for it.

HBM layout
  parameters   one flat fp32 buffer (master) + bf16 shadow + fp32 grad + Adam m, v; trainable
               tensors first so the optimizer and the gradient all-reduce see one contiguous range;
               q/k/v weights of a layer are adjacent, i.e. one [3H, H] matrix for the fused QKV GEMM.
  activations  token-major [rows, features], rows padded to a multiple of 256 with zero rows
               (GEMM tiles read them, epilogues never write them); residual stream fp32, GEMM
               operands bf16.  The fused [text | patch] sequence of sample b is rows b*S .. b*S+S-1.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import os

import numpy as np
import torch

from . import ops
from .ops import Drop, NO_DROP
from .spec import select_patches, VaultSpec, build_state, param_entries


def _pad(n: int, m: int = 256) -> int:
    return ((n + m - 1) // m) * m


def _in_format(method):
    """Run a method with its object's 16-bit operand format current (ops.operand_format): every launch inside goes to the
    library built for that format."""
    import functools

    @functools.wraps(method)
    def run(self, *a, **kw):
        with ops.operand_format(self.half):
            return method(self, *a, **kw)
    return run


class ParamStore:
    def __init__(self, spec: VaultSpec, device, state: Optional[Dict[str, np.ndarray]] = None, seed: int = 0,
                 freeze_lm: bool = False, with_grads: bool = True, half: str = "bf16"):
        self.spec, self.device, self.freeze_lm = spec, device, freeze_lm
        self.half, self.hdt = half, ops.HALF_DTYPE[half]      # operand format of the shadow copies (bf16 | fp16)
        entries = {n: s for n, s, _ in param_entries(spec)}
        order = self._flat_order(spec)
        assert set(order) == set(entries), "flat order must cover the parameter inventory"
        no_grad = set(self.no_grad_names(spec, freeze_lm))
        train = [n for n in order if n not in no_grad]
        rest = [n for n in order if n in no_grad]
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for n in train:
            self.offsets[n] = (off, entries[n])
            off += _pad(int(np.prod(entries[n])), 64)
        self.n_train = _pad(off, 1024)
        off = self.n_train
        for n in rest:
            self.offsets[n] = (off, entries[n])
            off += _pad(int(np.prod(entries[n])), 64)
        self.n_total = _pad(off, 1024)
        self.trainable = train
        self.frozen = rest
        # the MLP head's output projection is used as a GEMM operand padded to a multiple of 256 rows: readable (zero)
        # slack behind the buffers, never part of [0, n_train) that the optimizer and the all-reduce walk
        self.slack = (256 * spec.mlp_dims[1] + 1024 if (spec.n_classes > 0 and spec.head == "mlp") else
                      (256 * spec.vilt.hidden_size + 1024 if spec.head == "mlm" else 0))
        host = np.zeros(self.n_total + self.slack, np.float32)
        if state is None:
            state = build_state(spec, seed)
        for n, (o, shp) in self.offsets.items():
            host[o:o + int(np.prod(shp))] = np.asarray(state[n], np.float32).reshape(-1)
        self.p = torch.from_numpy(host).to(device)
        self.pb = torch.zeros(self.n_total + self.slack, dtype=self.hdt, device=device)
        with ops.operand_format(half):
            ops.cast_bf16(self.p, self.pb, self.n_total)
        # transposed bf16 shadow W^T [in][out] of the Linears whose data gradient runs as a forward-form GEMM on the
        # register-direct kernel (attention-out and FFN-out of every trained encoder layer): {weight name: tensor}
        self.pbT: Dict[str, torch.Tensor] = {}
        self._pbT_groups: List[tuple] = []
        self.g = self.m = self.v = None
        if with_grads:
            self.g = torch.zeros(self.n_train + self.slack, device=device)
            self.m = torch.zeros(self.n_train, device=device)
            self.v = torch.zeros(self.n_train, device=device)

    @staticmethod
    def no_grad_names(spec: VaultSpec, freeze_lm: bool) -> List[str]:
        out = []
        if spec.lm is not None:
            if spec.head != "mlm":     # (the MLM decoder is tied to ViLT's word embeddings: used and trained there)
                out.append("embeddings.text_embeddings.word_embeddings.weight")
            if not spec.use_vilt_position_embeddings:
                out.append("embeddings.text_embeddings.position_embeddings.weight")
            if freeze_lm:
                out += [n for n, _, _ in param_entries(spec) if n.startswith("bert.")]
        return out

    @staticmethod
    def _layer_order(prefix: str, style: str) -> List[str]:
        att = "attention.attention" if style == "vilt" else "attention.self"
        o = [f"{prefix}.{att}.{n}.weight" for n in ("query", "key", "value")]
        o += [f"{prefix}.{att}.{n}.bias" for n in ("query", "key", "value")]
        o += [f"{prefix}.attention.output.dense.weight", f"{prefix}.attention.output.dense.bias"]
        if style == "bert":
            o += [f"{prefix}.attention.output.LayerNorm.weight", f"{prefix}.attention.output.LayerNorm.bias"]
        else:
            o += [f"{prefix}.layernorm_before.weight", f"{prefix}.layernorm_before.bias",
                  f"{prefix}.layernorm_after.weight", f"{prefix}.layernorm_after.bias"]
        o += [f"{prefix}.intermediate.dense.weight", f"{prefix}.intermediate.dense.bias",
              f"{prefix}.output.dense.weight", f"{prefix}.output.dense.bias"]
        if style == "bert":
            o += [f"{prefix}.output.LayerNorm.weight", f"{prefix}.output.LayerNorm.bias"]
        return o

    @classmethod
    def _flat_order(cls, spec: VaultSpec) -> List[str]:
        o: List[str] = []
        if spec.lm is not None:
            o += ["bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
                  "bert.embeddings.token_type_embeddings.weight", "bert.embeddings.LayerNorm.weight",
                  "bert.embeddings.LayerNorm.bias"]
            for i in range(spec.lm.num_hidden_layers):
                o += cls._layer_order(f"bert.encoder.layer.{i}", "bert")
        o += ["embeddings.cls_token", "embeddings.position_embeddings",
              "embeddings.text_embeddings.word_embeddings.weight",
              "embeddings.text_embeddings.position_embeddings.weight",
              "embeddings.text_embeddings.token_type_embeddings.weight",
              "embeddings.text_embeddings.LayerNorm.weight", "embeddings.text_embeddings.LayerNorm.bias",
              "embeddings.patch_embeddings.projection.weight", "embeddings.patch_embeddings.projection.bias",
              "embeddings.token_type_embeddings.weight"]
        for i in range(spec.vilt.num_hidden_layers):
            o += cls._layer_order(f"encoder.layer.{i}", "vilt")
        o += ["layernorm.weight", "layernorm.bias"]
        if spec.add_pooling_layer:
            o += ["pooler.dense.weight", "pooler.dense.bias"]
        if spec.head == "mlm":
            # the vocabulary-sized bias last: it is read as a 256-padded GEMM operand (slack behind the buffers)
            o += ["mlm_score.transform.dense.weight", "mlm_score.transform.dense.bias",
                  "mlm_score.transform.LayerNorm.weight", "mlm_score.transform.LayerNorm.bias", "mlm_score.bias"]
        elif spec.n_classes > 0 and spec.head == "mlp":
            # the output projection last: its [n_classes, 2H] matrix is read (never written) as 256-row padded
            o += ["classifier.0.weight", "classifier.0.bias", "classifier.1.weight", "classifier.1.bias",
                  "classifier.3.weight", "classifier.3.bias"]
        elif spec.n_classes > 0:
            o += ["classifier.1.weight", "classifier.1.bias"]
        return o

    # ---- transposed weight shadow -----------------------------------------------------------
    @_in_format
    def enable_transposed(self, groups):
        """``groups``: lists of weight names of identical shape lying at a uniform stride in the flat buffer (the layers
        of a stack); one stacked [L, in, out] bf16 tensor per group, refreshed by :meth:`refresh_transposed`."""
        for names in groups:
            offs = [self.offsets[n][0] for n in names]
            shp = self.offsets[names[0]][1]
            rows, cols = int(shp[0]), int(np.prod(shp[1:]))
            stride = (offs[1] - offs[0]) if len(offs) > 1 else 0
            if rows % 64 or cols % 64 or stride % 8 or any(offs[k + 1] - offs[k] != stride for k in range(len(offs) - 1)):
                continue
            t = torch.zeros((len(names), cols, rows), dtype=self.hdt, device=self.device)
            for k, n in enumerate(names):
                self.pbT[n] = t[k]
            self._pbT_groups.append((offs[0], rows, cols, len(names), stride, t))
        self.refresh_transposed()

    @_in_format
    def refresh_transposed(self):
        """Re-derive the transposed shadow from the bf16 shadow (after every change of the parameters)."""
        for o, rows, cols, L, stride, t in self._pbT_groups:
            ops.transpose_bf16(self.pb[o:], t, rows, cols, L, stride, rows * cols)

    @_in_format
    def refresh_shadows(self):
        """fp32 master -> bf16 shadow -> transposed shadow (after the master changed outside the fused optimizer)."""
        ops.cast_bf16(self.p, self.pb, self.n_total)
        self.refresh_transposed()
        self._pb3_fresh = False

    # ---- views ------------------------------------------------------------------------------
    def _view(self, buf, name, n_elems=None, shape=None):
        o, shp = self.offsets[name]
        n = int(np.prod(shp)) if n_elems is None else n_elems
        return buf[o:o + n].view(*(shape if shape is not None else shp))

    def w(self, name, **kw):
        return self._view(self.p, name, **kw)

    def wb(self, name, **kw):
        return self._view(self.pb, name, **kw)

    def gr(self, name, **kw):
        if self.g is None or self.offsets[name][0] >= self.n_train:
            return None
        return self._view(self.g, name, **kw)

    # ---- split-bf16 (precise inference) weight shadow: [N][hi | hi | lo] per 2-D weight --------
    @_in_format
    def ensure_split3(self):
        if getattr(self, "pb3", None) is None:
            self.pb3 = torch.zeros(3 * self.n_total, dtype=self.hdt, device=self.device)
            self._pb3_fresh = False
        if self._pb3_fresh:
            return
        for n, (o, shp) in self.offsets.items():
            if len(shp) < 2 or not n.endswith("weight") or "embeddings.word" in n or "position_embeddings" in n \
                    or "token_type_embeddings" in n:
                continue
            N = shp[0]
            K = int(np.prod(shp[1:]))
            if K % 4:
                continue
            ops.split3_bf16(self.p[o:o + N * K], self.pb3[3 * o:3 * o + 3 * N * K], N, K, 1)
        self._pb3_fresh = True

    def wb3(self, name, N, K):
        o, _ = self.offsets[name]
        return self.pb3[3 * o:3 * o + 3 * N * K].view(N, 3 * K)

    def has_grad(self, name) -> bool:
        return self.g is not None and self.offsets[name][0] < self.n_train

    def state_dict_numpy(self) -> Dict[str, np.ndarray]:
        host = self.p.detach().cpu().numpy()
        return {n: host[o:o + int(np.prod(s))].reshape(s).copy() for n, (o, s) in self.offsets.items()}

    @_in_format
    def load_numpy(self, state: Dict[str, np.ndarray]):
        host = self.p.detach().cpu().numpy().copy()
        for n, v in state.items():
            o, shp = self.offsets[n]
            host[o:o + int(np.prod(shp))] = np.asarray(v, np.float32).reshape(-1)
        self.p.copy_(torch.from_numpy(host))
        self.refresh_shadows()


class _LayerNames:
    def __init__(self, prefix: str, style: str):
        att = "attention.attention" if style == "vilt" else "attention.self"
        self.qw, self.qb = f"{prefix}.{att}.query.weight", f"{prefix}.{att}.query.bias"
        self.ow, self.ob = f"{prefix}.attention.output.dense.weight", f"{prefix}.attention.output.dense.bias"
        self.iw, self.ib = f"{prefix}.intermediate.dense.weight", f"{prefix}.intermediate.dense.bias"
        self.fw, self.fb = f"{prefix}.output.dense.weight", f"{prefix}.output.dense.bias"
        if style == "vilt":
            self.ln1w, self.ln1b = f"{prefix}.layernorm_before.weight", f"{prefix}.layernorm_before.bias"
            self.ln2w, self.ln2b = f"{prefix}.layernorm_after.weight", f"{prefix}.layernorm_after.bias"
        else:
            self.ln1w, self.ln1b = (f"{prefix}.attention.output.LayerNorm.weight",
                                    f"{prefix}.attention.output.LayerNorm.bias")
            self.ln2w, self.ln2b = f"{prefix}.output.LayerNorm.weight", f"{prefix}.output.LayerNorm.bias"


class VaultEngine:
    """Forward / backward of VaultModel / VaultForTMSC over one batch resident in HBM."""

    DEFAULT_HALF = "fp16"          # operand format of an engine constructed without `half` (see __init__)
    WGRAD_TARGET_WGS = 768
    # Weight gradients of the encoder layers are deferred and contracted `LM_WGRAD_GROUP` layers per launch (vault_gemm
    # batch, ABI 3): one layer alone fills the GPU only with split-K partial sums through float atomics (and, for the
    # LM's 40-token sequences or small batches, 25-50 k-step blocks); a group of layers gives 100-400 tiles of the full
    # contraction.  Groups of 6 of the 12 layers measured best or equal at every batch size (B = 256: 42.7 ms/step,
    # one group of 12: 43.1, groups of 4: 43.6, per-layer launches: 45.6) and let a data-parallel step all-reduce the
    # upper group's gradients under the backward of the lower layers.  0 = all layers of a stack in one group.
    LM_WGRAD_BATCHED = True
    LM_WGRAD_GROUP = 6
    WGRAD_BATCH_RING = True        # batched launches on the 256x256 ring kernel where the shapes allow (else 128x128)
    # Encoder layers through the stage-level C ABI (vault_{vilt,lm}_layer_{fwd,bwd}: one C call per layer and direction,
    # the kernel order lives in csrc/stage.hip) when the step is host-launch-bound: up to this many (padded) fused token
    # rows; larger batches keep the per-kernel calls below (same kernels, same order) so that bench.py can bracket single
    # GEMM call sites with events.  VAULT_STAGE_ABI=0 / 1 forces either.
    STAGE_MAX_ROWS = 8192
    QKV_BIAS_SHORTCUT = True       # ViLT QKV bias gradient: value part from the dctx GEMM's epilogue, key part zero (see _backward)
    GELU8 = True                   # ViLT FFN: gelu' kept for backward as the 8-wave kernel's 8-bit tile-native image (vault_gemm aux_u8)
    GRAD_STREAM_BF16 = True        # ViLT residual-gradient stream in bf16 only (what bf16 autocast training carries): see _backward
    WGRAD_STREAM_MAX_ROWS = 16384  # deferred weight gradients run on a second stream up to this many ViLT token rows (B <= 88)
    WGRAD_BATCH_MAX_ROWS = 131072  # the ViLT layers take the same route up to this many (padded) token rows (B <= 708:
                                   # 22 GB of per-layer dY operands at that size; 7.9 GB at B = 256)

    def __init__(self, spec: VaultSpec, device="cuda:0", state=None, seed: int = 0, freeze_lm: bool = False,
                 with_grads: bool = True, classifier_dropout: float = 0.1, fp8_forward: bool = False,
                 half: Optional[str] = None, grad_scale_pow2: Optional[float] = None):
        self.spec, self.device = spec, torch.device(device)
        # 16-bit operand format of every GEMM / attention operand, saved activation and data gradient: "fp16" (DEFAULT_HALF: the
        # format that meets the reference's tolerance) - IEEE half operands (libvault_hip_f16.so), 11 significant bits: logits /
        # loss of the full-size stack inside 1e-3 of the fp32 reference (ref: vault/models/vault/model.py:557-570 runs fp32) -
        # or "bf16" (BASELINE's format, what bench.py times as `value`): the same kernels compiled for bf16 operands
        # (libvault_hip.so), same matrix rate, 8 significant bits (4e-3 on the logits).  fp16's narrow
        # exponent range is handled the classic way: the backward runs on gradients multiplied by the static power of
        # two `grad_scale` (exact in every format; default 2^12: |dlogits| <= 1 becomes 4096, elements down to 1.5e-8
        # stay normal numbers), the flat gradient buffer holds SCALED gradients while a backward runs, the fused
        # optimizer divides the scale out (TrainStep), the autograd bridge un-scales after each backward (backward());
        # conversions saturate at +-65504 (csrc/common.h H16_SATURATE) instead of producing infinities.
        if half is None:      # (the fp8-forward mode quantises bf16 operands)
            half = "bf16" if fp8_forward else self.DEFAULT_HALF
        if half not in ops.HALF_DTYPE:
            raise ValueError("half must be 'bf16' or 'fp16'")
        self.half, self.hdt = half, ops.HALF_DTYPE[half]
        if fp8_forward and half != "bf16":
            raise ValueError("the fp8-forward mode quantises bf16 operands: half must be 'bf16'")
        self.grad_scale = float(grad_scale_pow2) if grad_scale_pow2 is not None else (4096.0 if half == "fp16" else 1.0)
        if self.grad_scale <= 0 or math.frexp(self.grad_scale)[0] != 0.5:
            raise ValueError("grad_scale_pow2 must be a power of two")
        # BASELINE config "fp8 MFMA forward, bf16 backward": the forward Linear layers of both encoder stacks run on
        # MXFP8 operands (QKV and FFN-in: activations quantised in front of the GEMM, weights from the bf16 shadow once
        # per forward);
        # everything saved for backward, and backward itself, stay bf16 (straight-through)
        self.fp8_forward = fp8_forward
        self._w8: Dict[str, tuple] = {}
        self.freeze_lm = freeze_lm and spec.lm is not None
        self.classifier_dropout = classifier_dropout
        if spec.vilt.hidden_size % 256 or (spec.lm and spec.lm.hidden_size != spec.vilt.hidden_size):
            raise ValueError("hidden size must be a multiple of 256 and equal for LM and ViLT")
        if spec.vilt.hidden_size // spec.vilt.num_attention_heads != 64:
            raise ValueError("head dimension must be 64")
        with torch.cuda.device(self.device):
            self.params = ParamStore(spec, self.device, state, seed, self.freeze_lm, with_grads, half=half)
        if os.environ.get("VAULT_LM_WGRAD_BATCHED") == "0":   # development override (same-box A/B)
            self.LM_WGRAD_BATCHED = False
        if os.environ.get("VAULT_WGRAD_BATCH_RING") == "0":   # development override (same-box A/B)
            self.WGRAD_BATCH_RING = False
        if os.environ.get("VAULT_WGRAD_GROUPED") in ("0", "1"):   # development override (same-box A/B)
            self.WGRAD_GROUPED = os.environ["VAULT_WGRAD_GROUPED"] == "1"
        if os.environ.get("VAULT_HEAD_MAJOR") in ("0", "1"):      # development override (same-box A/B)
            self.HEAD_MAJOR = os.environ["VAULT_HEAD_MAJOR"] == "1"
        if os.environ.get("VAULT_WGRAD_BATCH_MAX_ROWS"):
            self.WGRAD_BATCH_MAX_ROWS = int(os.environ["VAULT_WGRAD_BATCH_MAX_ROWS"])
        if os.environ.get("VAULT_GELU8") in ("0", "1"):   # development override (same-box A/B)
            self.GELU8 = os.environ["VAULT_GELU8"] == "1"
        if os.environ.get("VAULT_GRAD_STREAM_BF16") in ("0", "1"):   # development override (same-box A/B)
            self.GRAD_STREAM_BF16 = os.environ["VAULT_GRAD_STREAM_BF16"] == "1"
        if os.environ.get("VAULT_WGRAD_GROUP"):   # layers per batched weight-gradient launch (tuning knob for DP runs)
            self.LM_WGRAD_GROUP = int(os.environ["VAULT_WGRAD_GROUP"])
        self.vl = [_LayerNames(f"encoder.layer.{i}", "vilt") for i in range(spec.vilt.num_hidden_layers)]
        self.ll = ([_LayerNames(f"bert.encoder.layer.{i}", "bert") for i in range(spec.lm.num_hidden_layers)]
                   if spec.lm else [])
        if with_grads and os.environ.get("VAULT_DGRAD_TRANSPOSED", "1") != "0":
            stacks = [self.vl] + ([self.ll] if (self.ll and not self.freeze_lm) else [])
            with torch.cuda.device(self.device):
                self.params.enable_transposed([[getattr(ln, k) for ln in st] for st in stacks for k in ("ow", "fw")])
        self._ws: Dict[tuple, dict] = {}
        self.keep_layer_outputs = False              # eval-mode forward: one residual-stream buffer per layer (output_hidden_states)
        self._sel_cache: Dict[tuple, dict] = {}      # patch bookkeeping of padded image batches, per patch-grid mask
        self.drop_seed = 0
        self.last: Optional[dict] = None
        self._wgrad_stream, self._wgrad_pending = None, False
        self._wgrad_side = False
        self._grads_zero = False
        self._g_dirty = False      # the flat gradient buffer may hold gradients of an API-level backward (not yet consumed / zeroed)
        self._g_stale_key = None   # tape key of the fused step whose stored weight-gradient ranges the optimizer left un-zeroed
        self._stored_ranges: List[Tuple[int, int]] = []
        # optional live kernel timing (bench.py): {site: [(start, end, flops), ...]} of torch.cuda.Event pairs recorded
        # on the launch stream around every launch of a kernel instantiation.  Sites: "wgrad" = the ring kernel's
        # weight-gradient form gemm256_kernel<1,1,EPI_F32_ATOMIC,4> (every _wgrad launch that takes it), "ffn1" =
        # the FFN-in forward GEMM gemm256_kernel<0,0,EPI_BF16_GELU,4> (ViLT and LM layers).
        self.profile_events: Optional[Dict[str, list]] = None
        self._e0: Dict[str, torch.cuda.Event] = {}
        # VAULT_H16_CENSUS=1 (debug; synchronises): after every eager forward / backward, what the 16-bit operand format did to
        # each 16-bit tensor of the workspace - {"forward" | "backward": {tensor name: {"saturated", "nonfinite", "subnormal",
        # "zero", "n"}}} (vault_h16_census: elements at the largest finite magnitude = what a saturating conversion leaves)
        self.census_on = os.environ.get("VAULT_H16_CENSUS") == "1"
        self.census: Dict[str, Dict[str, dict]] = {}

    def _prof_begin(self, site: str, stream=None):
        if self.profile_events is not None and site in self.profile_events:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)          # (the launch stream: the recording closure captures it, replays run elsewhere)
            self._e0[site] = e0

    def _prof_end(self, site: str, flops: float = 0.0, stream=None):
        if self.profile_events is not None and site in self.profile_events:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(stream)
            self.profile_events[site].append((self._e0[site], e1, flops))

    def _run_census(self, ws: dict, phase: str):
        if not self.census_on or ops.taping():
            return
        import re
        skip_u = ws.get("gelu8_active") is not None       # (the ViLT `u` buffers then hold the 8-bit tile-native gelu' image)
        names = [k for k, t in ws.items() if isinstance(k, str) and isinstance(t, torch.Tensor) and t.dtype == self.hdt
                 and not k.endswith("_all") and not (skip_u and re.fullmatch(r"u\d*", k))]
        seen, todo = set(), []
        for k in sorted(names):
            key = (ws[k].data_ptr(), ws[k].numel())
            if key not in seen and ws[k].is_contiguous():
                seen.add(key)
                todo.append(k)
        cnt = torch.zeros((max(1, len(todo)), 4), dtype=torch.int64, device=self.device)
        for j, k in enumerate(todo):
            ops.h16_census(ws[k], cnt[j])
        host = cnt.cpu().tolist()
        self.census[phase] = {k: dict(saturated=host[j][0], nonfinite=host[j][1], subnormal=host[j][2], zero=host[j][3],
                                      n=ws[k].numel()) for j, k in enumerate(todo)}

    # ---- deferred weight gradients on a second stream -------------------------------------------
    def _wgrads_aside(self, launch, after_layer):
        """Run ``launch()`` (the batched weight-gradient GEMMs of a group of layers) on the engine's second stream when the
        backward chain leaves CUs idle (few token rows: a chain GEMM of a small batch is a single partial round of tiles).
        Nothing in the chain reads the weight gradients: only the optimizer, which waits for the stream (_join_wgrads).
        Not in data-parallel steps (``after_layer``: the reducer starts on the main stream's events)."""
        if not self._wgrad_side or after_layer is not None:
            launch()
            return
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(self.device)
        side, main, ev = self._wgrad_stream, torch.cuda.current_stream(), torch.cuda.Event()
        ops.pycall(lambda: ev.record(main))
        ops.pycall(lambda: side.wait_event(ev))
        with torch.cuda.stream(side):
            launch()
        self._wgrad_pending = True

    def _join_wgrads(self):
        if self._wgrad_pending:
            side, main, ev = self._wgrad_stream, torch.cuda.current_stream(), torch.cuda.Event()
            ops.pycall(lambda: ev.record(side))
            ops.pycall(lambda: main.wait_event(ev))
            self._wgrad_pending = False

    # ---- workspace --------------------------------------------------------------------------
    def _buf(self, ws, name, shape, dtype):
        t = ws.get(name)
        if t is None:
            t = torch.zeros(shape, dtype=dtype, device=self.device)
            ws[name] = t
        return t

    def _stack(self, ws, base, n, shape, dtype):
        """`n` equally shaped buffers `base0 .. base{n-1}` as slices of ONE tensor (`base_all`): uniform stride between
        the layers of a stack, as the batched weight-gradient GEMM addresses them."""
        key = base + "_all"
        t = ws.get(key)
        if t is None:
            t = torch.zeros((n,) + tuple(shape), dtype=dtype, device=self.device)
            ws[key] = t
            for i in range(n):
                ws[f"{base}{i}"] = t[i]
        return t

    MAX_RAGGED_WORKSPACES = 2   # padded-image geometries kept alive (each owns every activation buffer of a step)

    def workspace(self, B: int, T: int, train: bool, geom: Tuple[int, int, int] = (0, 0, 0), tag: int = 0) -> dict:
        """Buffers of one (batch, text length, mode, image geometry); geom = (L, HP, WP) for padded batches of
        differently sized images, (0, 0, 0) for the square pre-training canvas with all-valid masks.  ``tag``
        separates the activation sets of several encoder passes that are alive at once (multi-image heads)."""
        key = (B, T, train) + tuple(geom) + (tag,)
        if key not in self._ws:
            if geom != (0, 0, 0):
                old = [k for k in self._ws if len(k) == 7 and k[3:6] != (0, 0, 0)]
                while len(old) >= self.MAX_RAGGED_WORKSPACES:
                    self._ws.pop(old.pop(0))
            self._ws[key] = {"B": B, "T": T, "key": key}
        return self._ws[key]

    # ---- helpers ----------------------------------------------------------------------------
    def _fp8_scratch(self, M, K):
        """MXFP8 image of the A operand of the GEMM about to run (consumed at once: one buffer per shape)."""
        key = ("fp8_a", M, K)
        if key not in self._ws:
            self._ws[key] = (torch.empty(M * K, dtype=torch.uint8, device=self.device),
                             torch.empty(M * (K // 32), dtype=torch.uint8, device=self.device))
        return self._ws[key]

    def _fp8_refresh_weights(self):
        """MXFP8 shadow of the encoder Linear weights, re-quantised from the bf16 shadow (a launch per weight: on
        the tape of a train step, so every step sees the weights the optimizer just wrote)."""
        P = self.params
        for ln in self.ll + self.vl:
            # the two Linears fed by a LayerNorm (K = H): measured at M = 47360 (tools/mx8_bench.py), quantise + MXFP8
            # GEMM 133 / 246 us against 192 / 336 us in bf16; attention-out is bound by its fp32 epilogue either way and
            # FFN-out would pay 85 us to quantise its [M, 4H] operand (until the GELU epilogue emits MXFP8 itself)
            for wname in (ln.qw, ln.iw):
                o, shp = P.offsets[wname]
                N = 3 * shp[0] if wname == ln.qw else shp[0]     # fused QKV: three [H, H] blocks stored back to back
                K = shp[1]
                if N % 256 or K % 128:
                    continue
                if wname not in self._w8:
                    self._w8[wname] = (torch.empty(N * K, dtype=torch.uint8, device=self.device),
                                       torch.empty(N * (K // 32), dtype=torch.uint8, device=self.device))
                wq, wsc = self._w8[wname]
                ops.quant_mxfp8(P.wb(wname, n_elems=N * K, shape=(N, K)), N, K, K, wq, wsc)

    @staticmethod
    def _keep_hi(split3: torch.Tensor, plain: torch.Tensor, K: int):
        """plain[:, :] = the `hi` third of a [rows, 3K] = [hi | lo | hi] split-bf16 operand: the bf16 value the fast mode would
        have stored (a strided device copy, no arithmetic) - what the bf16 backward reads in a precise-forward training step."""
        ops.pycall(lambda: plain.copy_(split3[:, :K]))

    def _linear(self, a_bf16, wname, out, M, N, K, epi, m_valid, bias=None, precise=False, ldo=None, prequant=False,
                **kw):
        """out = epilogue(A . W^T).  ``precise``: A is a [M, 3K] = [hi | lo | hi] split-bf16 operand and the
        weight its [N, 3K] = [hi | hi | lo] counterpart: the same kernel over a 3x longer contraction."""
        P = self.params
        if precise:
            ops.gemm(a_bf16, P.wb3(wname, N, K), out, M, N, 3 * K, 3 * K, 3 * K, N if ldo is None else ldo, 0, 0, epi,
                     m_valid=m_valid, bias=bias, **kw)
        elif self.fp8_forward and wname in self._w8 and M % 256 == 0:
            wq, wsc = self._w8[wname]
            aq, asc = self._fp8_scratch(M, K)
            if not prequant:          # (the LayerNorm in front wrote the MXFP8 image of its bf16 output itself)
                ops.quant_mxfp8(a_bf16, M, K, K, aq, asc)
            for k in ("split3", "cfg", "aux_u8"):   # (precise mode / the bf16 8-wave kernel's 8-bit gelu' only)
                kw.pop(k, None)
            ops.gemm_mxfp8(aq, asc, wq, wsc, out, M, N, K, N, epi, m_valid=m_valid, bias=bias, **kw)
        else:
            ops.gemm(a_bf16, P.wb(wname, n_elems=N * K, shape=(N, K)), out, M, N, K, K, K, N if ldo is None else ldo,
                     0, 0, epi, m_valid=m_valid, bias=bias, **kw)

    def _dgrad(self, dy_bf16, wname, out, M, Kin, Nout, epi, m_valid, **kw):
        # dX[M,Kin] = dY[M,Nout] . W[Nout,Kin]
        P = self.params
        wt = P.pbT.get(wname)
        if wt is not None and M % 256 == 0:
            # forward-form operands on the transposed shadow W^T [Kin][Nout]: the register-direct GEMM
            ops.gemm(dy_bf16, wt, out, M, Kin, Nout, Nout, Nout, Kin, 0, 0, epi, m_valid=m_valid, **kw)
            return
        ops.gemm(dy_bf16, P.wb(wname, n_elems=Nout * Kin, shape=(Nout, Kin)), out, M, Kin, Nout, Nout, Kin, Kin, 0, 1,
                 epi, m_valid=m_valid, **kw)

    def _wgrad(self, dy_bf16, x_bf16, wname, bname, Mtok_pad, Nout, Kin, m_valid, out_rows=0):
        # dW[Nout,Kin] += dY[Mtok,Nout]^T . X[Mtok,Kin] ; db[Nout] += colsum(dY)
        P = self.params
        gw = P.gr(wname, n_elems=Nout * Kin, shape=(Nout, Kin))
        if gw is None:
            return
        nk = Mtok_pad // 64
        if Mtok_pad <= 16384 and Nout % 128 == 0 and Kin % 128 == 0:
            # short contractions (the LM's 40-token sequences: 10240 rows at B = 256): 128x128 tiles with few splits
            # beat the 256x256 ring kernel, whose tiles x splits cannot fill the chip without very short K ranges
            # (tools/wgrad_sweep.py; in-step A/B on one box: +1.0 % samples/s)
            tiles = (Nout // 128) * (Kin // 128)
            splits = 7 if tiles <= 36 else (4 if tiles <= 108 else 3)
            cfg = 0
        elif Nout % 256 == 0 and Kin % 256 == 0:
            # 256x256 ring kernel; split the token contraction so that tiles x splits fills the 256 CUs once
            tiles = (Nout // 256) * (Kin // 256)
            splits = max(1, min(nk // 2, 256 // tiles, 16))   # >16 partial sums per element: float atomics dominate
            cfg = 3
        else:
            tiles = (Nout // 128) * (Kin // 128)
            splits = max(1, min(nk, (self.WGRAD_TARGET_WGS + tiles - 1) // tiles))
            cfg = 0
        if cfg == 3:
            ops.pycall(lambda: self._prof_begin("wgrad"))
        ops.gemm(dy_bf16, x_bf16, gw, Nout, Kin, Mtok_pad, Nout, Kin, Kin, 1, 1, ops.EPI_F32_ATOMIC, cfg=cfg,
                 splits=splits, accumulate=1, m_valid=out_rows)   # out_rows: rows of dW that exist (0 = all Nout)
        if cfg == 3:
            fl = 2.0 * m_valid * Nout * Kin
            ops.pycall(lambda: self._prof_end("wgrad", fl))
        if bname is not None:
            ops.colsum(dy_bf16, Nout, m_valid, Nout, P.gr(bname, n_elems=Nout, shape=(Nout,)))

    def _wgrad_batched(self, dY_all, X_all, wnames, i0, Mtok_pad, Nout, Kin, m_valid):
        """dW_l[Nout,Kin] += dY_l[Mtok,Nout]^T . X_l[Mtok,Kin] for the consecutive layers l = i0 .. i0 + len(wnames) - 1 of a
        stack in ONE launch (vault_gemm `batch`): dY_l / X_l are slices of the stacked operand tensors, the dW_l lie
        at a uniform stride in the flat gradient buffer (identical layer layouts)."""
        P = self.params
        G = len(wnames)
        offs = [P.offsets[w][0] for w in wnames]
        stride_o = (offs[1] - offs[0]) if G > 1 else 0
        if any(offs[k + 1] - offs[k] != stride_o for k in range(G - 1)) or Nout % 128 or Kin % 128:
            raise RuntimeError("batched weight gradients need identically laid out layers and 128-multiples")
        gw = P.gr(wnames[0], n_elems=Nout * Kin, shape=(Nout, Kin))
        nk = Mtok_pad // 64
        if Nout % 256 == 0 and Kin % 256 == 0 and self.WGRAD_BATCH_RING:
            # ring kernel, persistent over (layer, split, tile) items, layer-major: an XCD works on whole layers.  Split
            # count by a cost model of the launch: rounds of 256 blocks x (k-steps at 1.67 us + ~40 us fixed per item)
            cfg, tiles = 3, (Nout // 256) * (Kin // 256) * G
            cost = lambda sp: -(-tiles * sp // 256) * (1.67 * -(-nk // sp) + 40.0)   # noqa: E731
            splits = min((sp for sp in range(1, 9) if nk // sp >= 2), key=cost)
        else:
            cfg, tiles = 0, (Nout // 128) * (Kin // 128) * G
            splits = max(1, min(8, Mtok_pad // 512, int(round(512.0 / tiles))))   # ~two resident 128x128 blocks per CU
        st = torch.cuda.current_stream()
        if cfg == 3:
            ops.pycall(lambda: self._prof_begin("wgrad", st))
        # un-split launches whose caller vouches for zero gradients (the fused train step: AdamW cleared them) STORE the
        # tiles instead of adding them with float atomics (memory-side, ~1.3 TB/s against 6 TB/s for stores: 44 -> 10 us
        # of a 216-tile launch's tail)
        acc = 0 if (self._grads_zero and splits == 1) else 1
        ops.gemm(dY_all[i0], X_all[i0], gw, Nout, Kin, Mtok_pad, Nout, Kin, Kin, 1, 1, ops.EPI_F32_ATOMIC, cfg=cfg,
                 splits=splits, accumulate=acc, batch=G, batch_a=dY_all.stride(0), batch_b=X_all.stride(0),
                 batch_o=stride_o)
        if acc == 0:
            self._stored_ranges += [(o, Nout * Kin) for o in offs]
        if cfg == 3:
            fl = 2.0 * m_valid * Nout * Kin * G
            ops.pycall(lambda: self._prof_end("wgrad", fl, st))

    WGRAD_GROUPED = True           # the four weight-gradient kinds of a group of layers packed into full rounds of 256 tiles
    HEAD_MAJOR = True              # qkv / dqkv of large batches in the head-major layout [3][heads][rows][64] (see _plan_head_major)
    HEAD_MAJOR_MIN_ROWS = int(os.environ.get("VAULT_HEAD_MAJOR_MIN_ROWS", "16384"))   # ... from this many (padded) token rows of a stack
    WGRAD_SIDE_ITEMS = int(os.environ.get("VAULT_WGRAD_SIDE_ITEMS", "224"))   # items per grouped launch on the second stream (B = 64, same box: 256: 13.50 / 13.59 ms, 224: 13.32 / 13.46, 192: 13.23 / 13.47, 160: 13.63 / 13.53)

    def _wgrad_group_size(self, n_layers, after_layer):
        """Layers per deferred weight-gradient group.  A data-parallel step (``after_layer``: the reducer's stage listener)
        keeps groups of LM_WGRAD_GROUP layers - the upper group's gradient range goes on the wire under the backward of the
        layers below it; a single process takes the whole stack (1,296 tiles = five full rounds + 16 tiles, against two
        remainders of 136: B = 256, same box, 40.2 -> 39.8 ms per step; equal at B = 64).  VAULT_WGRAD_GROUP forces a size."""
        if os.environ.get("VAULT_WGRAD_GROUP"):
            g = self.LM_WGRAD_GROUP
        else:
            g = self.LM_WGRAD_GROUP if (after_layer is not None or not self.WGRAD_GROUPED) else 0
        return g if g > 0 else n_layers

    def _wgrad_group(self, kinds, layers, i0, hi, Mtok_pad, m_valid):
        """Weight gradients of layers i0 .. hi - 1 of a stack.  ``kinds``: (dY stack, X stack, weight attribute, Nout, Kin) per
        Linear kind.  Every 256 x 256 tile of every kind costs the same (the contraction runs over the tokens), so the tiles
        of all kinds are packed into launches of exactly 256 items - one per CU, un-split, stored (or added) once - and one
        remainder launch whose split count comes from the cost model (vault_wgrad_grouped); one launch per kind leaves 16 %
        of the CUs idle in the 216-tile FFN launches and splits the attention-out / QKV ones 4 / 3 ways with float atomics.
        Falls back to one batched launch per kind when a shape is not a multiple of 256 (the tiny test models)."""
        P = self.params
        G = hi - i0
        hms = [k[5] if len(k) > 5 else 0 for k in kinds]        # rows per plane of a head-major dY (the QKV kind's dqkv), 0 = row-major
        kinds = [k[:5] for k in kinds]
        ok = self.WGRAD_GROUPED and self.WGRAD_BATCH_RING and all(no % 256 == 0 and ki % 256 == 0 for *_, no, ki in kinds)
        strides = []
        for dY_all, X_all, wsel, Nout, Kin in kinds:
            offs = [P.offsets[getattr(l_, wsel)][0] for l_ in layers[i0:hi]]
            so = (offs[1] - offs[0]) if G > 1 else 0
            ok = ok and all(offs[k + 1] - offs[k] == so for k in range(G - 1))
            strides.append(so)
        if not ok:
            if any(hms):
                raise RuntimeError("head-major dqkv needs the grouped ring weight-gradient launches (_plan_head_major)")
            for dY_all, X_all, wsel, Nout, Kin in kinds:
                self._wgrad_batched(dY_all, X_all, [getattr(l_, wsel) for l_ in layers[i0:hi]], i0, Mtok_pad, Nout, Kin, m_valid)
            return
        nk = Mtok_pad // 64
        # items per launch: one per CU; beside a backward chain on another stream (small batches) fewer, so that the chain's
        # kernels find free CUs while a launch's persistent blocks hold theirs (WGRAD_SIDE_ITEMS)
        CU = self.WGRAD_SIDE_ITEMS if self._wgrad_side else 256
        # items of every kind in list order, cut into launches of CU items (<= 3 segments each)
        remaining = []
        for k, (dY_all, X_all, wsel, Nout, Kin) in enumerate(kinds):
            remaining.append([k, 0, (Nout // 256) * (Kin // 256) * G])      # kind, first item, items left
        launches, cur, room = [], [], CU
        for k, first, left in remaining:
            while left > 0:
                take = min(left, room)
                cur.append((k, first, take))
                first, left, room = first + take, left - take, room - take
                if room == 0 or len(cur) == 3:
                    launches.append(cur)
                    cur, room = [], CU
        if cur:
            launches.append(cur)
        st = torch.cuda.current_stream()
        covered: Dict[tuple, int] = {}      # (kind, layer) -> tiles written by store launches
        for segs in launches:
            count = sum(c for _, _, c in segs)
            if count == CU:
                splits = 1
            else:       # remainder: rounds of 256 pieces x (k-steps at 1.67 us + fixed cost per piece: ~10 us stored, ~50 us with float atomics)
                fixed = lambda sp: 10.0 if (sp == 1 and self._grads_zero) else 50.0   # noqa: E731
                cost = lambda sp: -(-count * sp // CU) * (1.67 * -(-nk // sp) + fixed(sp))   # noqa: E731
                splits = min((sp for sp in range(1, 9) if nk // sp >= 2), key=cost)
            acc = 0 if (self._grads_zero and splits == 1) else 1
            args = []
            for k, first, c in segs:
                dY_all, X_all, wsel, Nout, Kin = kinds[k]
                if acc == 0:
                    tpl = (Nout // 256) * (Kin // 256)
                    for it in range(first, first + c):       # items are (layer-major, tile-minor)
                        covered[(k, it // tpl)] = covered.get((k, it // tpl), 0) + 1
                gw = P.gr(getattr(layers[i0], wsel), n_elems=Nout * Kin, shape=(Nout, Kin))
                args.append(dict(dy=dY_all[i0], x=X_all[i0], dw=gw, n_out=Nout, n_in=Kin, batch=G, first=first, count=c,
                                 batch_dy=dY_all.stride(0), batch_x=X_all.stride(0), batch_dw=strides[k], dy_hm=hms[k]))
            ops.pycall(lambda: self._prof_begin("wgrad", st))
            ops.wgrad_grouped(args, Mtok_pad, splits=splits, accumulate=acc)
            fl = 2.0 * m_valid * 65536.0 * count
            ops.pycall(lambda fl=fl: self._prof_end("wgrad", fl, st))
        for (k, lay), n in covered.items():
            _, _, wsel, Nout, Kin = kinds[k]
            if n == (Nout // 256) * (Kin // 256):
                self._stored_ranges.append((P.offsets[getattr(layers[i0 + lay], wsel)][0], Nout * Kin))

    def _qkv_bias_grads_batched(self, dqkv_all, layers, i0, hi, ld, rows, N, hm=0):
        """QKV bias gradients of layers i0 .. hi - 1 (column sums over the token rows of the first N columns of their dqkv) in
        ONE launch, issued with the group's batched weight gradients: at small batches a single layer's pass is a 4 us read
        behind a 10 us launch + reduction tail, and next to the weight gradients it is off the backward chain."""
        P = self.params
        offs = [P.offsets[l_.qb][0] for l_ in layers[i0:hi]]
        stride_o = (offs[1] - offs[0]) if len(offs) > 1 else 0
        if any(offs[k + 1] - offs[k] != stride_o for k in range(len(offs) - 1)):
            raise RuntimeError("batched bias gradients need identically laid out layers")
        gqb = P.gr(layers[i0].qb, n_elems=ld, shape=(ld,))
        if hm:       # head-major dqkv: plane p = columns 64 p .. 64 p + 63
            ops.colsum_hm(dqkv_all[i0], rows, hm, N // 64, gqb, hi - i0, dqkv_all.stride(0), stride_o)
            return
        ops.colsum_batched(dqkv_all[i0], ld, rows, N, gqb, hi - i0, dqkv_all.stride(0), stride_o)

    def _plan_gelu8(self, ws, n2, act, u, ln, Mp, M):
        """Kernel configuration (5 / 6) on which BOTH the FFN-in forward and the gelu'-product data gradient of this ViLT
        workspace run with the 8-bit tile-native gelu' (vault_gemm aux_u8: an opaque image only the same kernel and shape reads
        back), or None: asked from the library once per workspace (vault_gemm_plan) - the automatic kernel choice must land on
        the 8-wave kernel with equal tile width for both (not in data-parallel steps, small batches, fp8-forward)."""
        mode = (bool(self.fp8_forward), ops.GEMM_SCHED, self.GELU8)
        if ws.get("gelu8_mode") == mode:
            return ws["gelu8_cfg"]
        ws["gelu8_mode"] = mode
        P, H, FF = self.params, ws["H"], ws["FF"]
        cfg, wt = None, P.pbT.get(ln.fw)
        if self.GELU8 and u is not None and wt is not None and not self.fp8_forward and Mp % 256 == 0:
            c1 = ops.gemm(n2, P.wb(ln.iw, n_elems=FF * H, shape=(FF, H)), act, Mp, FF, H, H, H, FF, 0, 0, ops.EPI_BF16_GELU,
                          m_valid=M, bias=P.w(ln.ib), out2=u, aux_u8=True, plan_only=True)
            c2 = ops.gemm(n2, wt, act, Mp, FF, H, H, H, FF, 0, 0, ops.EPI_BF16_DGELU, m_valid=M, aux=u,
                          colsum=P.gr(ln.ib), aux_u8=True, plan_only=True)
            if c1 in (5, 6) and c1 == c2:
                cfg = c1
        ws["gelu8_cfg"] = cfg
        return cfg

    def _plan_head_major(self, ws, key, a16, wname, rows_pad, rows, S, pr, train):
        """Rows per plane (= rows_pad) when this stack's qkv / dqkv live HEAD-MAJOR in this workspace, else 0.  Head-major
        ([3][heads][rows_pad][64] in the same buffers) makes a (batch, head) item's rows contiguous: the attention kernels'
        loads and the backward's dq / dk / dv stores stream instead of touching 128-byte segments at a 4.6 KB stride
        (tools/attn_bench.py, B = 256: backward 201-206 -> 188 us, LM shape 31 -> 28).  Needs every producer / consumer on a
        kernel that serves the layout - QKV forward on the 8-wave kernel (out_hm), QKV data gradient on the ring kernel (a_hm),
        weight gradients through the grouped ring launches (dy_hm), the single-pass attention backward (S <= 192) - which the
        library is asked about once per workspace (vault_gemm_plan); small batches, the precise and fp8-forward modes and the
        stage-level calls keep the row-major layout."""
        mode = (bool(self.fp8_forward), ops.GEMM_SCHED, bool(pr), bool(train), self.HEAD_MAJOR, self.WGRAD_GROUPED)
        if ws.get(key + "_mode") == mode:
            return ws[key]
        ws[key + "_mode"] = mode
        P, H, FF = self.params, ws["H"], ws["FF"]
        hm = 0
        if (self.HEAD_MAJOR and not pr and not self.fp8_forward and S <= 192 and rows_pad % 256 == 0 and H % 256 == 0 and FF % 256 == 0
                and self.HEAD_MAJOR_MIN_ROWS <= rows_pad <= self.WGRAD_BATCH_MAX_ROWS and os.environ.get("VAULT_ATTN_BWD", "1") != "0" and os.environ.get("VAULT_ATTN_BWD_S", "1") != "0"):
            w = P.wb(wname, n_elems=3 * H * H, shape=(3 * H, H))
            # (plan only: the pointers are not dereferenced, but must not be null)
            c1 = ops.gemm(a16, w, a16, rows_pad, 3 * H, H, H, H, 3 * H, 0, 0, ops.EPI_BF16, m_valid=rows,
                          bias=P.w(wname.replace("weight", "bias"), n_elems=3 * H, shape=(3 * H,)), out_hm=rows_pad, plan_only=True)
            ok = c1 in (5, 6)
            if ok and train:
                c2 = ops.gemm(a16, w, a16, rows_pad, H, 3 * H, 3 * H, H, H, 0, 1, ops.EPI_BF16, m_valid=rows, a_hm=rows_pad,
                              plan_only=True)
                ok = c2 in (3, 4, 8) and self.LM_WGRAD_BATCHED and self.WGRAD_GROUPED and self.WGRAD_BATCH_RING \
                    and P.gr(wname) is not None
            if ok:
                hm = rows_pad
        ws[key] = hm
        return hm

    def _use_stage(self, rows_pad: int, pr: bool) -> bool:
        e = os.environ.get("VAULT_STAGE_ABI")
        if pr or self.fp8_forward:
            return False
        if e in ("0", "1"):
            return e == "1"
        return rows_pad <= self.STAGE_MAX_ROWS

    def _stage_layer_args(self, ws, ln, style, i, rows, rows_pad, S, keymask, x_in, x_out, bufs, drops=None,
                          x_in_bf16=None, x_out_bf16=None):
        """vault_layer_args of one encoder layer (kept in the workspace: backward refers to it)."""
        P = self.params
        H, FF, heads, B = ws["H"], ws["FF"], ws["heads"], ws["B"]
        eps = self.spec.vilt.layer_norm_eps if style == "vilt" else self.spec.lm.layer_norm_eps
        kw = dict(B=B, S=S, H=H, FF=FF, heads=heads, rows=rows, rows_pad=rows_pad, eps=eps,
                  wqkv=P.wb(ln.qw, n_elems=3 * H * H, shape=(3 * H, H)), wo=P.wb(ln.ow), wi=P.wb(ln.iw), wf=P.wb(ln.fw),
                  wo_t=P.pbT.get(ln.ow), wf_t=P.pbT.get(ln.fw),
                  bqkv=P.w(ln.qb, n_elems=3 * H, shape=(3 * H,)), bo=P.w(ln.ob), bi=P.w(ln.ib), bf=P.w(ln.fb),
                  ln1w=P.w(ln.ln1w), ln1b=P.w(ln.ln1b), ln2w=P.w(ln.ln2w), ln2b=P.w(ln.ln2b),
                  x_in=x_in, x_out=x_out, x_in_bf16=x_in_bf16, x_out_bf16=x_out_bf16, keymask=keymask, **bufs)
        if drops is not None:
            da, dh = drops
            kw.update(attn_drop_thresh=da.thresh, attn_drop_scale=da.scale, hid_drop_thresh=dh.thresh, hid_drop_scale=dh.scale,
                      drop_seed=self.drop_seed & 0xFFFFFFFF, drop_stream_base=16 * i)
        a = ops.layer_args(**{k: v for k, v in kw.items() if v is not None})
        ws[f"stage_{style}{i}"] = a
        return a

    def _drop(self, p: float, stream: int, train: bool) -> Drop:
        return Drop(p, self.drop_seed, stream) if (train and p > 0.0) else NO_DROP

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, batch: Dict[str, torch.Tensor], train: bool = False, labels: Optional[torch.Tensor] = None,
                need_hidden: bool = True, loss_scale: Optional[float] = None,
                precise: bool = False, ws_tag: int = 0, image_token_type_idx: int = 1,
                advance_seed: bool = True) -> Dict[str, torch.Tensor]:
        """batch tensors must already be on the device (int64 ids / mask, f32 pixels).  Returns device
        tensors; in train mode keeps every activation needed by :meth:`backward` (in the workspace ``self.last``).
        ``ws_tag`` / ``image_token_type_idx`` / ``advance_seed=False``: further encoder passes over other images of
        the same samples (HF ``ViltForImagesAndTextClassification``: modality type i + 1 for image i, one LM pass -
        here one per image with identical dropout masks)."""
        with torch.cuda.device(self.device):
            return self._forward(batch, train, labels, need_hidden, loss_scale, precise, ws_tag, image_token_type_idx,
                                 advance_seed)

    def stage_inputs(self, batch: Dict[str, torch.Tensor], train: bool, labels: Optional[torch.Tensor] = None,
                     validate: bool = True, ws_tag: int = 0) -> dict:
        """Validate the batch (HF-style errors) and copy it into the persistent input buffers of the
        (B, T, train) workspace, so that every kernel argument of a step is pointer-stable (required for
        tape replay).  ``validate=False`` skips the pixel-mask check (it synchronises the device)."""
        spec, v = self.spec, self.spec.vilt
        ids = batch.get("input_ids")
        temb = batch.get("inputs_embeds")          # [B, T, H] f32 instead of token ids (ref model.py:170-200)
        if ids is None and temb is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        B, T = (ids.shape if ids is not None else temb.shape[:2])
        H, FF, heads = v.hidden_size, v.intermediate_size, v.num_attention_heads
        if temb is not None and tuple(temb.shape) != (B, T, H):
            raise ValueError(f"inputs_embeds must be [B, T, {H}]")
        iemb = batch.get("image_embeds")           # [B, L, H] f32 instead of pixels (HF modeling_vilt.py:190-207)
        if iemb is not None:
            return self._stage_image_embeds(batch, train, labels, ws_tag, ids, temb, iemb, B, T)
        if batch.get("pixel_patches") is not None:
            return self._stage_pixel_patches(batch, train, labels, ws_tag, ids, temb, B, T)
        pix = batch["pixel_values"]
        if pix.dim() != 4 or pix.shape[1] != v.num_channels or pix.shape[2] % v.patch_size or pix.shape[3] % v.patch_size:
            raise ValueError(f"pixel_values must be [B,{v.num_channels},HP,WP] with HP, WP multiples of the patch size "
                             f"{v.patch_size}")
        if pix.shape[0] != B:
            raise ValueError("The text inputs and image inputs need to have the same batch size")
        pm = batch.get("pixel_mask")
        HP, WP = int(pix.shape[2]), int(pix.shape[3])
        square = (HP == v.image_size and WP == v.image_size)
        # padded batches of differently sized images (HF visual_embed, modeling_vilt.py:92-178): the patch bookkeeping
        # runs on the host (like the reference's own python loops over the batch), the arithmetic on the device.
        # ``validate=False`` on the square canvas means "the caller vouches for an all-ones pixel_mask" (no sync).
        if pm is not None and tuple(pm.shape) != (B, HP, WP):
            raise ValueError("pixel_mask must be [B,HP,WP] like pixel_values")
        # only the patch grid of the mask matters (nearest-neighbour interpolation reads pixel_mask[:, ::ps, ::ps]):
        # subsample on the device, bring B x gh x gw bytes to the host
        grid_h = None
        vhw = batch.get("valid_hw")    # host-side hint: the valid (h, w) pixels of every image, top-left on the canvas (what an
        #                                image processor knows when it pads: DeviceImageProcessor returns it) - the patch
        #                                grid of the mask is then built on the host, no device -> host read of pixel_mask
        if vhw is not None:
            vhw = tuple((int(h_), int(w_)) for h_, w_ in vhw)
            if len(vhw) != B or any(h_ <= 0 or w_ <= 0 or h_ > HP or w_ > WP for h_, w_ in vhw):
                raise ValueError("valid_hw must list (h, w) <= the canvas for every image of the batch")
            ps = v.patch_size
            grid_h = np.zeros((B, HP // ps, WP // ps), np.uint8)
            for b_, (h_, w_) in enumerate(vhw):      # nearest-neighbour subsampling reads pixel (i ps, j ps): valid iff < (h, w)
                grid_h[b_, :(h_ + ps - 1) // ps, :(w_ + ps - 1) // ps] = 1
        elif pm is not None and (validate or not square):
            grid_h = (pm[:, ::v.patch_size, ::v.patch_size] != 0).to(torch.uint8).cpu().numpy()
        ragged = (not square) or (grid_h is not None and not bool(grid_h.all()))
        geom = (0, 0, 0)
        if ragged:
            if grid_h is None:
                grid_h = np.ones((B, HP // v.patch_size, WP // v.patch_size), np.uint8)
            # the bookkeeping of a batch depends on its patch-grid mask only: cached per mask (a data loader that buckets by
            # size repeats geometries; a repeated batch costs a dictionary lookup instead of the per-sample host loops)
            ckey = (T, grid_h.shape, grid_h.tobytes())
            hit = self._sel_cache.get(ckey)
            if hit is None:
                sel, valid, hw, (gh, gw), L0 = select_patches(grid_h, 1, getattr(v, "max_image_length", -1))
                # round the image part up to a multiple of 8 rows with more masked padding (fewer distinct geometries);
                # the attention kernels hold at most 320 keys
                cap = 320 - T - 1
                if L0 > cap:
                    raise ValueError(f"fused sequence {T + 1 + L0} exceeds the attention kernels' 320 keys")
                L = min(((L0 + 7) // 8) * 8, cap)
                if L > L0:   # extra rows repeat the last slot and are masked like any padding
                    sel = np.concatenate([sel, np.repeat(sel[:, -1:], L - L0, axis=1)], axis=1)
                    valid = np.concatenate([valid, np.zeros((B, L - L0), np.int32)], axis=1)
                hit = dict(L=L, gw=gw, n_valid=valid.sum(axis=1),
                           sel=torch.from_numpy(np.ascontiguousarray(sel)).to(self.device),
                           hw=torch.from_numpy(np.ascontiguousarray(hw)).to(self.device),
                           valid=torch.from_numpy(valid.astype(np.float32)).to(self.device))
                if len(self._sel_cache) >= 64:
                    self._sel_cache.pop(next(iter(self._sel_cache)))
                self._sel_cache[ckey] = hit
            L, gw = hit["L"], hit["gw"]
            geom = (L, HP, WP)
            NP = L
        else:
            NP = v.num_patches
        S = T + 1 + NP
        ws = self.workspace(B, T, train, geom, ws_tag)
        ws.update(S=S, M=B * S, Mp=_pad(B * S), H=H, FF=FF, heads=heads, NP=NP, train=train,
                  Ml=B * T, Mlp=_pad(B * T), ragged=ragged, HP=HP, WP=WP)
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        self._stage_text(ws, ids, temb, B, T, H)
        buf("in_pix", tuple(pix.shape)).copy_(pix)
        ws["img_embeds"] = None
        ws["patches_in"] = False
        km = buf("keymask", (B, S))
        am = batch.get("attention_mask")
        if am is None:
            km.fill_(1.0)
        else:
            km[:, :T] = am
            km[:, T:] = 1.0
        if ragged:
            ws["gw"] = gw
            buf("in_sel", (B, NP), torch.int32).copy_(hit["sel"])      # (device-to-device from the cached bookkeeping)
            buf("in_hw", (B, 2), torch.int32).copy_(hit["hw"])
            km[:, T + 1:] = hit["valid"]
            ws["sel"], ws["hw"], ws["n_valid"] = ws["in_sel"], ws["in_hw"], hit["n_valid"]
        buf("in_amf", (B, T)).copy_(km[:, :T])
        tt = batch.get("token_type_ids")
        ws["ids"], ws["pix"], ws["amf"] = ws["in_ids"], ws["in_pix"], ws["in_amf"]
        ws["tt"] = None if tt is None else buf("in_tt", (B, T), torch.int64).copy_(tt)
        ws["labels"] = self._stage_labels(buf, labels, B)
        return ws

    def _stage_text(self, ws, ids, temb, B, T, H):
        """Token ids, or text embeddings in their place (``inputs_embeds``: the word-embedding lookup is skipped; position
        ids then count every position like HF ``create_position_ids_from_inputs_embeds``: ids that are never the pad id)."""
        idb = self._buf(ws, "in_ids", (B, T), torch.int64)
        if temb is None:
            idb.copy_(ids)
            ws["txt_embeds"] = None
        else:
            pad = self.spec.lm.pad_token_id if self.spec.lm is not None else 0
            idb.fill_(pad + 1)
            ws["txt_embeds"] = self._buf(ws, "in_temb", (_pad(B * T), H), torch.float32)
            ws["txt_embeds"][:B * T].copy_(temb.reshape(B * T, H))

    def _stage_pixel_patches(self, batch, train, labels, ws_tag, ids, temb, B, T):
        """Staging for images that arrive as the patch-embedding GEMM's operand: ``pixel_patches`` = the bf16 unfold
        [B * patches, C ps ps] of square, fully valid ``image_size`` canvases (what ``vault_image_preprocess`` writes straight
        from its resize kernel: ``DeviceImageProcessor.from_packed(patch_out=...)``).  The f32 pixel tensor and the unfold pass
        do not exist on this path; a ``pixel_mask``, if given, must be all ones (not checked: it would synchronise)."""
        v = self.spec.vilt
        H, FF, heads = v.hidden_size, v.intermediate_size, v.num_attention_heads
        NP, Kp = v.num_patches, v.num_channels * v.patch_size * v.patch_size
        pp = batch["pixel_patches"]
        # what an image processor that padded knows (DeviceImageProcessor.from_packed returns both): this entry takes square,
        # fully valid canvases only - a padded image, or another canvas with the same patch count, would be attended to as
        # real tokens (no device synchronisation: host values)
        cv, vhw = batch.get("canvas"), batch.get("valid_hw")
        if cv is not None and tuple(int(c) for c in cv) != (v.image_size, v.image_size):
            raise ValueError(f"pixel_patches need the square {v.image_size} x {v.image_size} canvas, got {tuple(cv)}: pass pixel_values")
        if vhw is not None and any((int(h_), int(w_)) != (v.image_size, v.image_size) for h_, w_ in vhw):
            raise ValueError("pixel_patches need fully valid images (every valid_hw equal to the canvas): pass pixel_values + pixel_mask "
                             "for padded batches")
        if pp.dtype != self.hdt or pp.numel() != B * NP * Kp:
            raise ValueError(f"pixel_patches must be {self.half} [{B} * {NP}, {Kp}] (square {v.image_size} x {v.image_size} canvases)")
        S = T + 1 + NP
        ws = self.workspace(B, T, train, (0, 0, 0), ws_tag)
        ws.update(S=S, M=B * S, Mp=_pad(B * S), H=H, FF=FF, heads=heads, NP=NP, train=train, Ml=B * T, Mlp=_pad(B * T),
                  ragged=False, HP=v.image_size, WP=v.image_size)
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        self._stage_text(ws, ids, temb, B, T, H)
        ap = buf("apatch", (_pad(B * NP), Kp), self.hdt)
        ap[:B * NP].copy_(pp.reshape(B * NP, Kp))          # (onto itself when the caller wrote into input_buffers()["pixel_patches"])
        ws["img_embeds"] = None
        km = buf("keymask", (B, S))
        am = batch.get("attention_mask")
        if am is None:
            km.fill_(1.0)
        else:
            km[:, :T] = am
            km[:, T:] = 1.0
        buf("in_amf", (B, T)).copy_(km[:, :T])
        tt = batch.get("token_type_ids")
        ws["ids"], ws["pix"], ws["amf"] = ws["in_ids"], None, ws["in_amf"]
        ws["tt"] = None if tt is None else buf("in_tt", (B, T), torch.int64).copy_(tt)
        ws["labels"] = self._stage_labels(buf, labels, B)
        ws["patches_in"] = True
        return ws

    def _stage_image_embeds(self, batch, train, labels, ws_tag, ids, temb, iemb, B, T):
        """Staging for externally supplied image embeddings: the image part of the fused sequence is ``image_embeds`` +
        modality type, ``pixel_mask`` [B, L] is its key mask (HF: ``image_masks = pixel_mask.flatten(1)``)."""
        v = self.spec.vilt
        H, FF, heads = v.hidden_size, v.intermediate_size, v.num_attention_heads
        if iemb.dim() != 3 or iemb.shape[0] != B or iemb.shape[2] != H:
            raise ValueError(f"image_embeds must be [B, L, {H}]")
        L = int(iemb.shape[1])
        S = T + L
        if S > 320:
            raise ValueError(f"fused sequence {S} exceeds the attention kernels' 320 keys")
        ws = self.workspace(B, T, train, (L, -1, -1), ws_tag)
        ws.update(S=S, M=B * S, Mp=_pad(B * S), H=H, FF=FF, heads=heads, NP=L, train=train, Ml=B * T, Mlp=_pad(B * T),
                  ragged=False, HP=0, WP=0)
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        self._stage_text(ws, ids, temb, B, T, H)
        ws["img_embeds"] = buf("in_iemb", (_pad(B * L), H))
        ws["img_embeds"][:B * L].copy_(iemb.reshape(B * L, H))
        km = buf("keymask", (B, S))
        am = batch.get("attention_mask")
        km[:, :T] = 1.0 if am is None else am
        pm = batch.get("pixel_mask")
        km[:, T:] = 1.0 if pm is None else pm.reshape(B, L)
        buf("in_amf", (B, T)).copy_(km[:, :T])
        tt = batch.get("token_type_ids")
        ws["ids"], ws["pix"], ws["amf"] = ws["in_ids"], None, ws["in_amf"]
        ws["tt"] = None if tt is None else buf("in_tt", (B, T), torch.int64).copy_(tt)
        ws["labels"] = self._stage_labels(buf, labels, B)
        return ws

    def _stage_labels(self, buf, labels, B):
        """int64 class labels (cross-entropy, ref: tmsc_utils/trainer.py:241-242) or float targets of the single-logit
        head (BCE with logits, ref: models/vault/trainer.py:55-56), each in its own persistent buffer."""
        if labels is None:
            return None
        if labels.dtype.is_floating_point:
            if self.spec.n_classes != 1 or self.spec.head == "mlp":
                raise ValueError("float targets (BCE-with-logits) need the single-logit classifier (n_classes = 1)")
            return buf("in_targets", (B,), torch.float32).copy_(labels.reshape(B))
        return buf("in_labels", (B,), torch.int64).copy_(labels.reshape(B))

    def input_buffers(self, B: int, T: int, train: bool = True) -> Dict[str, torch.Tensor]:
        """The persistent input staging buffers of the (B, T) workspace on the square pre-training canvas
        (``input_ids``, ``pixel_values``, ``labels``).  A data loader may write its host->device copies straight into
        them and pass these very tensors to :meth:`stage_inputs` / ``TrainStep``: staging then copies nothing
        (``Tensor.copy_`` onto itself is a no-op), which saves one device-to-device pass over the pixels per step."""
        v = self.spec.vilt
        ws = self.workspace(B, T, train)
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        Kp = v.num_channels * v.patch_size * v.patch_size
        return {"input_ids": buf("in_ids", (B, T), torch.int64),
                "pixel_values": buf("in_pix", (B, v.num_channels, v.image_size, v.image_size)),
                # (alternative image input: the bf16 patch unfold, _stage_pixel_patches - the patch-embedding GEMM's own operand)
                "pixel_patches": buf("apatch", (_pad(B * v.num_patches), Kp), self.hdt)[:B * v.num_patches],
                "labels": buf("in_labels", (B,), torch.int64)}

    def _forward(self, batch, train, labels, need_hidden, loss_scale, precise=False, ws_tag=0, image_type_idx=1,
                 advance_seed=True):
        # (precise + train: split-bf16 FORWARD GEMMs - logits / loss at fp32 class - with the bf16 backward; every operand the
        #  backward reads is also kept in its plain bf16 form, see forward_staged)
        if not 0 < image_type_idx < self.spec.vilt.modality_type_vocab_size:
            raise ValueError("image_token_type_idx outside the modality type table")
        ws = self.stage_inputs(batch, train, labels, ws_tag=ws_tag)
        ws["img_type"] = image_type_idx
        if train and advance_seed:
            self.drop_seed = (self.drop_seed + 1) & 0xFFFFFFFF
        if precise:
            self.params.ensure_split3()
        return self.forward_staged(ws, need_hidden, loss_scale, precise)

    @_in_format
    def forward_staged(self, ws: dict, need_hidden: bool = True, loss_scale: Optional[float] = None,
                       precise: bool = False):
        """Forward over the staged inputs of ``ws`` (every launch goes through ops.* and can be taped)."""
        spec, P = self.spec, self.params
        v = spec.vilt
        train = ws["train"]
        B, T, S, M, Mp, H, FF, heads, NP = (ws[k] for k in ("B", "T", "S", "M", "Mp", "H", "FF", "heads", "NP"))
        ids, tt, amf, pix, labels, km = ws["ids"], ws["tt"], ws["amf"], ws["pix"], ws["labels"], ws["keymask"]
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        bf = self.hdt
        ws["drop_seed"] = self.drop_seed
        pr = precise
        W3 = 3 if pr else 1   # operand width multiplier of the split-bf16 path
        pt = pr and train     # precise forward of a training step: the plain bf16 operands of the backward are kept beside the split ones
        if pr:
            if pt:
                self.params._pb3_fresh = False   # (a recorded train step must carry the re-split of the weights the optimizer just wrote)
            self.params.ensure_split3()
        if self.fp8_forward and not pr:
            self._fp8_refresh_weights()

        # ------------------------------ language model ------------------------------
        if spec.lm is not None:
            lm = spec.lm
            Ml, Mlp = B * T, _pad(B * T)
            ws.update(Ml=Ml, Mlp=Mlp)
            lm_tt = tt if (tt is not None and lm.type_vocab_size >= 2) else 0   # ref: model.py:174-180
            ws["lm_tt"] = lm_tt
            pos = buf("lm_pos", (B, T), torch.int32)
            ops.position_ids(ids, pos, B, T, 1 if lm.kind == "roberta" else 0, lm.pad_token_id)
            esum = buf("lm_esum", (Mlp, H))
            te = ws.get("txt_embeds")
            ops.gather_sum(te, esum, [None if te is not None else (P.w("bert.embeddings.word_embeddings.weight"), ids),
                                      (P.w("bert.embeddings.position_embeddings.weight"), pos),
                                      (P.w("bert.embeddings.token_type_embeddings.weight"), lm_tt)], Ml, H)
            keep = train and not self.freeze_lm
            nl = lm.num_hidden_layers
            if keep and self.LM_WGRAD_BATCHED and H % 128 == 0 and FF % 128 == 0:
                # X operands of the deferred, batched weight gradients: one tensor per kind, a layer per slice
                self._stack(ws, "lm_yb", nl + 1, (Mlp, H), bf)
                for base, width in (("lm_ctx", H), ("lm_y1b", H), ("lm_act", FF)):
                    self._stack(ws, base, nl, (Mlp, width), bf)
            y = [buf(f"lm_y{i}" if keep else f"lm_y{i % 2}", (Mlp, H)) for i in range(nl + 1)]
            yb = [buf((f"lm_yb{i}" if keep else f"lm_yb{i % 2}") + ("_3" if pr else ""), (Mlp, W3 * H), bf)
                  for i in range(nl + 1)]
            ybs = [buf(f"lm_yb{i}" if keep else f"lm_yb{i % 2}", (Mlp, H), bf) for i in range(nl + 1)] if pt else None
            lm_train = train   # dropout stays active in a frozen LM too (ref: model.py:189 only disables grad)
            pdh, pda = lm.hidden_dropout_prob, lm.attention_probs_dropout_prob
            q8l = self._fp8_scratch(Mlp, H) if (self.fp8_forward and not pr and Mlp % 256 == 0) else (None, None)
            ops.layernorm_fwd(esum, P.w("bert.embeddings.LayerNorm.weight"), P.w("bert.embeddings.LayerNorm.bias"),
                              lm.layer_norm_eps, Ml, H, y_f32=y[0], y_bf16=(ybs[0] if pt else None) if pr else yb[0],
                              y_split3=yb[0] if pr else None, mean=buf("lm_emean", (Mlp,)),
                              rstd=buf("lm_erstd", (Mlp,)), drop=self._drop(pdh, 1, lm_train),
                              y_q=q8l[0], y_scale=q8l[1])
            lm_stage = self._use_stage(Mlp, pr)
            ws["lm_stage"] = lm_stage
            ops.pycall(lambda: self._prof_begin("lm_fwd"))
            for i, ln in enumerate(self.ll):
                sfx = f"{i}" if keep else ""
                qkv = buf(f"lm_qkv{sfx}", (Mlp, 3 * H), bf)
                p3 = "_3" if pr else ""
                ctx = buf(f"lm_ctx{sfx}{p3}", (Mlp, W3 * H), bf)
                lse = buf(f"lm_lse{sfx}", (B, heads, T))
                h1 = buf(f"lm_h1{sfx}", (Mlp, H)); y1 = buf(f"lm_y1{sfx}", (Mlp, H))
                y1b = buf(f"lm_y1b{sfx}{p3}", (Mlp, W3 * H), bf)
                u = buf(f"lm_u{sfx}", (Mlp, FF), bf) if keep else None
                act = buf(f"lm_act{sfx}{p3}", (Mlp, W3 * FF), bf)
                h2 = buf(f"lm_h2{sfx}", (Mlp, H))
                if lm_stage:
                    ws["lm_qkv_hm"], ws["lm_qkv_hm_mode"] = 0, None
                    da, dh = self._drop(pda, 16 * i + 2, lm_train), self._drop(pdh, 16 * i + 3, lm_train)
                    a = self._stage_layer_args(
                        ws, ln, "lm", i, Ml, Mlp, T, amf, y[i], y[i + 1],
                        dict(qkv=qkv, ctx=ctx, lse=lse, xm=h1, y1=y1, n2=y1b, act=act, u=u, h2=h2,
                             m1=buf(f"lm_m1{sfx}", (Mlp,)), r1=buf(f"lm_r1{sfx}", (Mlp,)),
                             m2=buf(f"lm_m2{sfx}", (Mlp,)), r2=buf(f"lm_r2{sfx}", (Mlp,))),
                        drops=(da, dh), x_in_bf16=yb[i], x_out_bf16=yb[i + 1])
                    ops.layer_call("vault_lm_layer_fwd", a, seeded=bool(da.thresh or dh.thresh))
                    continue
                lhm = self._plan_head_major(ws, "lm_qkv_hm", yb[i], ln.qw, Mlp, Ml, T, pr, keep)
                self._linear(yb[i], ln.qw, qkv, Mlp, 3 * H, H, ops.EPI_BF16, Ml,
                             bias=P.w(ln.qb, n_elems=3 * H, shape=(3 * H,)), precise=pr, prequant=q8l[0] is not None,
                             **(dict(out_hm=lhm) if lhm else {}))
                ops.attention_fwd(qkv, amf, None if pr else ctx, lse, B, T, H, heads,
                                  drop=self._drop(pda, 16 * i + 2, lm_train), ctx_split3=ctx if pr else None, qkv_hm=lhm)
                if pt:
                    self._keep_hi(ctx, buf(f"lm_ctx{sfx}", (Mlp, H), bf), H)
                self._linear(ctx, ln.ow, h1, Mlp, H, H, ops.EPI_F32_RES, Ml, bias=P.w(ln.ob), res=y[i],
                             drop=self._drop(pdh, 16 * i + 3, lm_train), precise=pr)
                ops.layernorm_fwd(h1, P.w(ln.ln1w), P.w(ln.ln1b), lm.layer_norm_eps, Ml, H, y_f32=y1,
                                  y_bf16=(buf(f"lm_y1b{sfx}", (Mlp, H), bf) if pt else None) if pr else y1b,
                                  y_split3=y1b if pr else None,
                                  mean=buf(f"lm_m1{sfx}", (Mlp,)), rstd=buf(f"lm_r1{sfx}", (Mlp,)),
                                  y_q=q8l[0], y_scale=q8l[1])
                ops.pycall(lambda: self._prof_begin("ffn1"))
                # (the epilogue addresses gelu' with the row stride of its main output: in the split form a [rows, 3 FF] buffer
                #  whose first third is written)
                u_out = buf(f"lm_u{sfx}_3", (Mlp, W3 * FF), bf) if (pt and u is not None) else u
                self._linear(y1b, ln.iw, act, Mlp, FF, H, ops.EPI_BF16_GELU, Ml, bias=P.w(ln.ib), out2=u_out, precise=pr,
                             split3=pr, ldo=W3 * FF, prequant=q8l[0] is not None)
                if pt:
                    self._keep_hi(act, buf(f"lm_act{sfx}", (Mlp, FF), bf), FF)
                    if u is not None:
                        self._keep_hi(u_out, u, FF)
                fl_l = 2.0 * Ml * FF * H * W3
                ops.pycall(lambda: self._prof_end("ffn1", fl_l))
                self._linear(act, ln.fw, h2, Mlp, H, FF, ops.EPI_F32_RES, Ml, bias=P.w(ln.fb), res=y1,
                             drop=self._drop(pdh, 16 * i + 4, lm_train), precise=pr)
                ops.layernorm_fwd(h2, P.w(ln.ln2w), P.w(ln.ln2b), lm.layer_norm_eps, Ml, H, y_f32=y[i + 1],
                                  y_bf16=(ybs[i + 1] if pt else None) if pr else yb[i + 1], y_split3=yb[i + 1] if pr else None,
                                  mean=buf(f"lm_m2{sfx}", (Mlp,)), rstd=buf(f"lm_r2{sfx}", (Mlp,)),
                                  y_q=q8l[0], y_scale=q8l[1])
            ops.pycall(lambda: self._prof_end("lm_fwd"))
            text_src = y[nl]
            use_pos = spec.use_vilt_position_embeddings
            tables = [(P.w("embeddings.text_embeddings.token_type_embeddings.weight"), tt if tt is not None else 0)]
        else:
            Ml, Mlp = B * T, _pad(B * T)
            ws.update(Ml=Ml, Mlp=Mlp)
            text_src = ws.get("txt_embeds")        # inputs_embeds replace ViLT's own word-embedding lookup
            use_pos = True
            tables = [(P.w("embeddings.text_embeddings.token_type_embeddings.weight"), tt if tt is not None else 0),
                      None if text_src is not None else (P.w("embeddings.text_embeddings.word_embeddings.weight"), ids)]
        if use_pos:
            tables.append((P.w("embeddings.text_embeddings.position_embeddings.weight"), "mod"))
        ws["use_pos"] = use_pos

        # ------------------------------ ViLT embeddings ------------------------------
        nv = v.num_hidden_layers
        x = [buf(f"x{i}" if (train or self.keep_layer_outputs) else f"x{i % 2}", (Mp, H)) for i in range(nv + 1)]
        vsum = buf("vt_sum", (Mlp, H))
        ops.gather_sum(text_src, vsum, tables, Ml, H, period=T)
        mt = P.w("embeddings.token_type_embeddings.weight")
        ops.layernorm_fwd(vsum, P.w("embeddings.text_embeddings.LayerNorm.weight"),
                          P.w("embeddings.text_embeddings.LayerNorm.bias"), v.layer_norm_eps, Ml, H, y_f32=x[0],
                          ymap=(T, S, 0), post_add=mt[0], mean=buf("vt_mean", (Mlp,)), rstd=buf("vt_rstd", (Mlp,)))
        Kp = v.num_channels * v.patch_size * v.patch_size
        Mpp = _pad(B * NP)
        ws.update(Kp=Kp, Mpp=Mpp)
        if ws.get("img_embeds") is not None:
            # externally supplied image embeddings: + modality type, straight into the image rows of the fused sequence
            ops.rows_add(ws["img_embeds"], mt[ws.get("img_type", 1)], x[0], B * NP, H, NP, S, T)
        else:
            self._patch_embed_forward(ws, x, mt, pix, pr, W3, Kp, Mpp, buf, bf)
        ws["lm_y"], ws["lm_yb"] = (y if spec.lm is not None else None), ((ybs if pt else yb) if spec.lm is not None else None)
        return self._forward_encoder(ws, x, need_hidden, loss_scale, pr, W3, labels, km, train, buf, bf)

    def _patch_embed_forward(self, ws, x, mt, pix, pr, W3, Kp, Mpp, buf, bf):
        spec, P = self.spec, self.params
        v = spec.vilt
        B, T, S, H, NP = (ws[k] for k in ("B", "T", "S", "H", "NP"))
        apatch = buf("apatch_3" if pr else "apatch", (Mpp, W3 * Kp), bf)
        addtab = buf("addtab", (NP, H))
        wpn = "embeddings.patch_embeddings.projection.weight"
        if ws["ragged"]:
            # padded batch of differently sized images: selected patch slots only, per-image resized position table
            ops.im2col_sel(pix, apatch, ws["sel"], B, NP, v.num_channels, ws["HP"], ws["WP"], v.patch_size, split3=pr)
            if pr and ws["train"]:
                ops.im2col_sel(pix, buf("apatch", (Mpp, Kp), bf), ws["sel"], B, NP, v.num_channels, ws["HP"], ws["WP"], v.patch_size)
            ops.image_sel_consts(P.w("embeddings.patch_embeddings.projection.bias"), P.w("embeddings.position_embeddings"),
                                 mt[ws.get("img_type", 1)], P.w("embeddings.cls_token"), addtab, x[0], NP, H, B, S, T)
        else:
            if ws.get("patches_in"):
                if pr:
                    raise ValueError("pixel_patches carry bf16 pixels: the precise (split-bf16) mode needs pixel_values")
            else:
                ops.im2col(pix, apatch, B, v.num_channels, v.image_size, v.patch_size, split3=pr)
                if pr and ws["train"]:    # the weight gradient's operand
                    ops.im2col(pix, buf("apatch", (Mpp, Kp), bf), B, v.num_channels, v.image_size, v.patch_size)
            ops.image_consts(P.w("embeddings.patch_embeddings.projection.bias"), P.w("embeddings.position_embeddings"),
                             mt[ws.get("img_type", 1)], P.w("embeddings.cls_token"), addtab, x[0], NP, H, B, S, T)
        ops.gemm(apatch, P.wb3(wpn, H, Kp) if pr else P.wb(wpn, shape=(H, Kp)), x[0], Mpp, H, W3 * Kp, W3 * Kp, W3 * Kp,
                 H, 0, 0, ops.EPI_F32_PATCH, m_valid=B * NP, addtab=addtab, rpg=NP, gstride=S, goff=T + 1)
        if ws["ragged"]:
            ops.image_pos_sel_fwd(x[0], P.w("embeddings.position_embeddings"), ws["sel"], ws["hw"], B, NP, S, T, H,
                                  ws["gw"], v.image_size // v.patch_size)

    def _forward_encoder(self, ws, x, need_hidden, loss_scale, pr, W3, labels, km, train, buf, bf):
        spec, P = self.spec, self.params
        v = spec.vilt
        B, T, S, M, Mp, H, FF, heads, NP = (ws[k] for k in ("B", "T", "S", "M", "Mp", "H", "FF", "heads", "NP"))
        nv = v.num_hidden_layers
        # ------------------------------ ViLT encoder ------------------------------
        pt = pr and train
        if (train and self.LM_WGRAD_BATCHED and Mp <= self.WGRAD_BATCH_MAX_ROWS and H % 128 == 0
                and FF % 128 == 0):
            # the ViLT layers' weight gradients are deferred and batched like the LM's
            for base, width in (("n1", H), ("ctx", H), ("n2", H), ("act", FF)):
                self._stack(ws, base, nv, (Mp, width), bf)
        ops.pycall(lambda: self._prof_begin("vilt_fwd"))
        for i, ln in enumerate(self.vl):
            sfx = f"{i}" if train else ""
            p3 = "_3" if pr else ""
            n1 = buf(f"n1{sfx}{p3}", (Mp, W3 * H), bf); qkv = buf(f"qkv{sfx}", (Mp, 3 * H), bf)
            ctx = buf(f"ctx{sfx}{p3}", (Mp, W3 * H), bf); lse = buf(f"lse{sfx}", (B, heads, S))
            xm = buf(f"xm{sfx}", (Mp, H)); n2 = buf(f"n2{sfx}{p3}", (Mp, W3 * H), bf)
            u = buf(f"u{sfx}", (Mp, FF), bf) if train else None
            act = buf(f"act{sfx}{p3}", (Mp, W3 * FF), bf)
            if self._use_stage(Mp, pr):
                ws["vilt_stage"] = True
                ws["qkv_hm"], ws["qkv_hm_mode"] = 0, None
                a = self._stage_layer_args(
                    ws, ln, "vilt", i, M, Mp, S, km, x[i], x[i + 1],
                    dict(n1=n1, qkv=qkv, ctx=ctx, lse=lse, xm=xm, n2=n2, act=act, u=u, m1=buf(f"m1{sfx}", (Mp,)),
                         r1=buf(f"r1{sfx}", (Mp,)), m2=buf(f"m2{sfx}", (Mp,)), r2=buf(f"r2{sfx}", (Mp,))))
                ops.layer_call("vault_vilt_layer_fwd", a)
                continue
            ws["vilt_stage"] = False
            q8 = self._fp8_scratch(Mp, H) if (self.fp8_forward and not pr and Mp % 256 == 0) else (None, None)
            ops.layernorm_fwd(x[i], P.w(ln.ln1w), P.w(ln.ln1b), v.layer_norm_eps, M, H,
                              y_bf16=(buf(f"n1{sfx}", (Mp, H), bf) if pt else None) if pr else n1, y_split3=n1 if pr else None, mean=buf(f"m1{sfx}", (Mp,)), rstd=buf(f"r1{sfx}", (Mp,)),
                              y_q=q8[0], y_scale=q8[1])
            vhm = self._plan_head_major(ws, "qkv_hm", n1, ln.qw, Mp, M, S, pr, train)
            self._linear(n1, ln.qw, qkv, Mp, 3 * H, H, ops.EPI_BF16, M, bias=P.w(ln.qb, n_elems=3 * H, shape=(3 * H,)),
                         precise=pr, prequant=q8[0] is not None, **(dict(out_hm=vhm) if vhm else {}))
            ops.attention_fwd(qkv, km, None if pr else ctx, lse, B, S, H, heads, ctx_split3=ctx if pr else None, qkv_hm=vhm)
            if pt:
                self._keep_hi(ctx, buf(f"ctx{sfx}", (Mp, H), bf), H)
            self._linear(ctx, ln.ow, xm, Mp, H, H, ops.EPI_F32_RES, M, bias=P.w(ln.ob), res=x[i], precise=pr)
            ops.layernorm_fwd(xm, P.w(ln.ln2w), P.w(ln.ln2b), v.layer_norm_eps, M, H,
                              y_bf16=(buf(f"n2{sfx}", (Mp, H), bf) if pt else None) if pr else n2, y_split3=n2 if pr else None, mean=buf(f"m2{sfx}", (Mp,)), rstd=buf(f"r2{sfx}", (Mp,)),
                              y_q=q8[0], y_scale=q8[1])
            ops.pycall(lambda: self._prof_begin("ffn1"))
            g8 = None if pr else self._plan_gelu8(ws, n2, act, u, ln, Mp, M)
            ws["gelu8_active"] = g8     # what THIS forward stored in `u` (8-bit tile image or plain 16-bit): backward reads this
            g8kw = dict(cfg=g8, aux_u8=True) if g8 is not None else {}
            u_out = buf(f"u{sfx}_3", (Mp, W3 * FF), bf) if (pt and u is not None) else u      # (row stride of the main output)
            self._linear(n2, ln.iw, act, Mp, FF, H, ops.EPI_BF16_GELU, M, bias=P.w(ln.ib), out2=u_out, precise=pr,
                         split3=pr, ldo=W3 * FF, prequant=q8[0] is not None, **g8kw)
            if pt:
                self._keep_hi(act, buf(f"act{sfx}", (Mp, FF), bf), FF)
                if u is not None:
                    self._keep_hi(u_out, u, FF)
            fl_v = 2.0 * M * FF * H * W3
            ops.pycall(lambda: self._prof_end("ffn1", fl_v))
            self._linear(act, ln.fw, x[i + 1], Mp, H, FF, ops.EPI_F32_RES, M, bias=P.w(ln.fb), res=xm, precise=pr)

        ops.pycall(lambda: self._prof_end("vilt_fwd"))
        # ------------------------------ tail ------------------------------
        out: Dict[str, torch.Tensor] = {}
        xl = x[nv]
        lw, lb = P.w("layernorm.weight"), P.w("layernorm.bias")
        if need_hidden:
            hid = buf("last_hidden", (Mp, H))
            ops.layernorm_fwd(xl, lw, lb, v.layer_norm_eps, M, H, y_f32=hid, mean=buf("f_mean_all", (Mp,)),
                              rstd=buf("f_rstd_all", (Mp,)))
            out["last_hidden_state"] = hid[:M].view(B, S, H)
        if spec.add_pooling_layer:
            Bp = _pad(B)
            ws["Bp"] = Bp
            h0b = buf("h0b_3" if pr else "h0b", (Bp, W3 * H), bf)
            ops.layernorm_fwd(xl, lw, lb, v.layer_norm_eps, B, H, y_bf16=(buf("h0b", (Bp, H), bf) if pt else None) if pr else h0b,
                              y_split3=h0b if pr else None, xmap=(1, S, 0), mean=buf("f_mean", (Bp,)),
                              rstd=buf("f_rstd", (Bp,)))
            pre = buf("pool_pre", (Bp, H))
            self._linear(h0b, "pooler.dense.weight", pre, Bp, H, H, ops.EPI_F32_RES, B, bias=P.w("pooler.dense.bias"),
                         precise=pr)
            pooled = buf("pooled", (Bp, H))
            if spec.n_classes > 0 and spec.head == "mlp":
                ops.head_fwd(pre, None, None, None, pooled, None, None, B, H, 0, 0.0)   # tanh
                if spec.num_images == 1:     # (multi-image heads run on the concatenated pooled outputs: mlp_head_forward)
                    out["logits"] = self._mlp_forward(ws, pooled, B)
            elif spec.n_classes > 0:
                C = spec.n_classes
                logits = buf("logits", (B, C))
                loss = buf("loss", (1,))
                ops.pycall(loss.zero_)
                hd = self._drop(self.classifier_dropout, 9001, train)
                ops.head_fwd(pre, P.w("classifier.1.weight"), P.w("classifier.1.bias"), labels, pooled, logits,
                             loss if labels is not None else None, B, H, C,
                             (1.0 / B) if loss_scale is None else loss_scale, drop=hd)
                out["logits"] = logits if C > 1 else logits.view(B)
                if labels is not None:
                    out["loss"] = loss
            else:
                ops.head_fwd(pre, None, None, None, pooled, None, None, B, H, 0, 0.0)   # VaultModel: tanh only
            out["pooler_output"] = pooled[:B]
        ws["x"] = x
        self.last = ws
        self._run_census(ws, "forward")
        return out

    # ---- MLP task head (HF ViltForQuestionAnswering / ViltForImagesAndTextClassification .classifier) ------------
    def _mlp_forward(self, ws: dict, x_f32: torch.Tensor, B: int) -> torch.Tensor:
        """logits = Linear(GELU(LayerNorm(Linear(x))))  for x [>= B rows, H_in] f32; Linear(H_in, H_mid) -
        LayerNorm(H_mid, eps 1e-5) - GELU - Linear(H_mid, L).  The output projection runs with L padded to 256 columns
        (readable slack behind the parameter buffers).  Buffers live in ``ws``."""
        spec, P = self.spec, self.params
        Hin, Hm = spec.mlp_dims
        L = spec.n_classes
        Lp, Bp = _pad(L), _pad(B)
        bf = self.hdt
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        xb = buf("mlp_xb", (Bp, Hin), bf)
        ops.cast_bf16(x_f32, xb, B * Hin)
        h1 = buf("mlp_h1", (Bp, Hm))
        self._linear(xb, "classifier.0.weight", h1, Bp, Hm, Hin, ops.EPI_F32_RES, B, bias=P.w("classifier.0.bias"))
        n1 = buf("mlp_n1", (Bp, Hm))
        ops.layernorm_fwd(h1, P.w("classifier.1.weight"), P.w("classifier.1.bias"), 1e-5, B, Hm, y_f32=n1,
                          mean=buf("mlp_mean", (Bp,)), rstd=buf("mlp_rstd", (Bp,)))
        a1 = buf("mlp_a1", (Bp, Hm), bf)
        ops.gelu_fwd(n1, a1, Bp * Hm)
        lg = buf("mlp_logits", (Bp, Lp))
        ops.gemm(a1, P.wb("classifier.3.weight", n_elems=Lp * Hm, shape=(Lp, Hm)), lg, Bp, Lp, Hm, Hm, Hm, Lp, 0, 0,
                 ops.EPI_F32_RES, m_valid=B, bias=P.w("classifier.3.bias", n_elems=Lp, shape=(Lp,)))
        return lg[:B, :L]

    def _mlp_backward(self, ws: dict, dlogits: torch.Tensor, B: int, scale: float = 1.0) -> torch.Tensor:
        """Parameter gradients of the MLP head (+=) and d/dx [Bp, H_in] f32 of the last :meth:`_mlp_forward` on ``ws``.
        ``scale``: the gradient scale of the operand format, applied to ``dlogits`` (every result is scaled by it)."""
        spec, P = self.spec, self.params
        Hin, Hm = spec.mlp_dims
        L = spec.n_classes
        Lp, Bp = _pad(L), _pad(B)
        bf = self.hdt
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        dl32 = buf("mlp_dlogits", (Bp, Lp))
        ops.pycall(dl32.zero_)
        dl32[:B, :L].copy_(dlogits.reshape(B, L))
        if scale != 1.0:
            ops.scale(dl32.view(-1), scale, Bp * Lp)
        dlb = buf("mlp_dlogits_b", (Bp, Lp), bf)
        ops.cast_bf16(dl32, dlb, Bp * Lp)
        # output projection: weight-gradient rows >= L are never written (m_valid); its bias gradient is the column sum
        # (the padded columns are zero and land in the slack behind the gradient buffer)
        self._wgrad(dlb, ws["mlp_a1"], "classifier.3.weight", "classifier.3.bias", Bp, Lp, Hm, B, out_rows=L)
        da1 = buf("mlp_da1", (Bp, Hm))
        ops.gemm(dlb, P.wb("classifier.3.weight", n_elems=Lp * Hm, shape=(Lp, Hm)), da1, Bp, Hm, Lp, Lp, Hm, Hm, 0, 1,
                 ops.EPI_F32_RES, m_valid=B)
        dn1 = buf("mlp_dn1", (Bp, Hm))
        ops.gelu_bwd(ws["mlp_n1"], da1, dn1, Bp * Hm)
        dh1b = buf("mlp_dh1b", (Bp, Hm), bf)
        ops.layernorm_bwd(ws["mlp_h1"], ws["mlp_mean"], ws["mlp_rstd"], P.w("classifier.1.weight"), B, Hm, dy_f32=dn1,
                          dx_bf16=dh1b, dgamma=P.gr("classifier.1.weight"), dbeta=P.gr("classifier.1.bias"),
                          dbias=P.gr("classifier.0.bias"))
        self._wgrad(dh1b, ws["mlp_xb"], "classifier.0.weight", None, Bp, Hm, Hin, B)
        dx = buf("mlp_dx", (Bp, Hin))
        ops.gemm(dh1b, P.wb("classifier.0.weight", shape=(Hm, Hin)), dx, Bp, Hin, Hm, Hm, Hin, Hin, 0, 1, ops.EPI_F32_RES,
                 m_valid=B)
        return dx

    # ---- MLM head (HF ViltMLMHead): dense(H, H) - GELU - LayerNorm - decoder tied to ViLT's word embeddings + bias --------
    @_in_format
    def mlm_head_forward(self, x_f32: torch.Tensor) -> torch.Tensor:
        """x [R, H] f32 (text rows of last_hidden_state) -> logits [R, V]."""
        with torch.cuda.device(self.device):
            spec, P = self.spec, self.params
            v = spec.vilt
            H, V = v.hidden_size, v.vocab_size
            R = x_f32.shape[0]
            Rp, Vp = _pad(R), _pad(V)
            ws = self._ws.setdefault(("mlm_head", R), {})
            bf = self.hdt
            buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
            xin = buf("x", (Rp, H))
            xin[:R].copy_(x_f32)
            xb = buf("xb", (Rp, H), bf)
            ops.cast_bf16(xin, xb, Rp * H)
            h1 = buf("h1", (Rp, H))
            self._linear(xb, "mlm_score.transform.dense.weight", h1, Rp, H, H, ops.EPI_F32_RES, R,
                         bias=P.w("mlm_score.transform.dense.bias"))
            a = buf("a", (Rp, H))
            ops.gelu_fwd_f32(h1, a, Rp * H)
            nb = buf("nb", (Rp, H), bf)
            ops.layernorm_fwd(a, P.w("mlm_score.transform.LayerNorm.weight"), P.w("mlm_score.transform.LayerNorm.bias"),
                              v.layer_norm_eps, R, H, y_bf16=nb, mean=buf("mean", (Rp,)), rstd=buf("rstd", (Rp,)))
            lg = buf("logits", (Rp, Vp))
            wn = "embeddings.text_embeddings.word_embeddings.weight"
            ops.gemm(nb, P.wb(wn, n_elems=Vp * H, shape=(Vp, H)), lg, Rp, Vp, H, H, H, Vp, 0, 0, ops.EPI_F32_RES, m_valid=R,
                     bias=P.w("mlm_score.bias", n_elems=Vp, shape=(Vp,)))
            return lg[:R, :V]

    @_in_format
    def mlm_head_backward(self, dlogits: torch.Tensor) -> torch.Tensor:
        self._api_backward_begins()
        with torch.cuda.device(self.device), self._grads_scaled():
            spec, P = self.spec, self.params
            v = spec.vilt
            H, V = v.hidden_size, v.vocab_size
            R = dlogits.shape[0]
            Rp, Vp = _pad(R), _pad(V)
            ws = self._ws[("mlm_head", R)]
            bf = self.hdt
            buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
            dl32 = buf("dlogits", (Rp, Vp))
            dl32.zero_()
            dl32[:R, :V].copy_(dlogits.reshape(R, V))
            if self.grad_scale != 1.0:
                ops.scale(dl32.view(-1), self.grad_scale, Rp * Vp)
            dlb = buf("dlogits_b", (Rp, Vp), bf)
            ops.cast_bf16(dl32, dlb, Rp * Vp)
            wn = "embeddings.text_embeddings.word_embeddings.weight"
            self._wgrad(dlb, ws["nb"], wn, "mlm_score.bias", Rp, Vp, H, R, out_rows=V)
            dn = buf("dn", (Rp, H))
            ops.gemm(dlb, P.wb(wn, n_elems=Vp * H, shape=(Vp, H)), dn, Rp, H, Vp, Vp, H, H, 0, 1, ops.EPI_F32_RES, m_valid=R)
            da = buf("da", (Rp, H))
            ops.layernorm_bwd(ws["a"], ws["mean"], ws["rstd"], P.w("mlm_score.transform.LayerNorm.weight"), R, H, dy_f32=dn,
                              dx_f32=da, dgamma=P.gr("mlm_score.transform.LayerNorm.weight"),
                              dbeta=P.gr("mlm_score.transform.LayerNorm.bias"))
            dh1 = buf("dh1", (Rp, H))
            ops.gelu_bwd(ws["h1"], da, dh1, Rp * H)
            dh1b = buf("dh1b", (Rp, H), bf)
            ops.cast_bf16(dh1, dh1b, Rp * H)
            self._wgrad(dh1b, ws["xb"], "mlm_score.transform.dense.weight", "mlm_score.transform.dense.bias", Rp, H, H, R)
            dx = buf("dx", (Rp, H))
            ops.gemm(dh1b, P.wb("mlm_score.transform.dense.weight", shape=(H, H)), dx, Rp, H, H, H, H, H, 0, 1,
                     ops.EPI_F32_RES, m_valid=R)
            if self.grad_scale != 1.0:
                ops.scale(dx.view(-1), 1.0 / self.grad_scale, dx.numel())
            return dx[:R]

    @_in_format
    def mlp_head_forward(self, x_f32: torch.Tensor, train: bool = True) -> torch.Tensor:
        """The MLP head on an external input [B, H_in] (concatenated pooled outputs of several encoder passes)."""
        with torch.cuda.device(self.device):
            B = x_f32.shape[0]
            ws = self._ws.setdefault(("mlp_head", B), {})
            xin = self._buf(ws, "mlp_xin", (_pad(B), self.spec.mlp_dims[0]), torch.float32)
            xin[:B].copy_(x_f32)
            return self._mlp_forward(ws, xin, B)

    @_in_format
    def mlp_head_backward(self, dlogits: torch.Tensor) -> torch.Tensor:
        self._api_backward_begins()
        with torch.cuda.device(self.device), self._grads_scaled():
            B = dlogits.shape[0]
            dx = self._mlp_backward(self._ws[("mlp_head", B)], dlogits.contiguous().float(), B, scale=self.grad_scale)
            if self.grad_scale != 1.0:
                ops.scale(dx.view(-1), 1.0 / self.grad_scale, dx.numel())
            return dx[:B]

    # ---- backward ---------------------------------------------------------------------------
    @_in_format
    def zero_grad(self):
        if self.params.g is not None:
            self.params.g.zero_()
        self._g_dirty = False
        self._g_stale_key = None

    def backward(self, grad_scale: Optional[float] = None, dlogits: Optional[torch.Tensor] = None,
                 dpooled: Optional[torch.Tensor] = None, dhidden: Optional[torch.Tensor] = None,
                 after_layer=None, ws: Optional[dict] = None):
        """Accumulate parameter gradients of the last train-mode forward into the flat grad buffer.

        Default (VaultForTMSC + labels): d(mean CE)/d(params), scaled by ``grad_scale`` (1/B).
        ``dlogits`` / ``dpooled`` / ``dhidden`` inject external output gradients (autograd bridge).
        ``after_layer(tag)`` is called after each stage so a DP driver can start all-reducing the
        gradient range that just became final.
        """
        self._api_backward_begins()
        with torch.cuda.device(self.device), self._grads_scaled():
            self._backward(grad_scale, dlogits, dpooled, dhidden, after_layer, ws)
            if self.grad_scale != 1.0:      # gradients handed back to the caller's autograd graph
                w_ = self.last if ws is None else ws
                for k in ("d_inputs_embeds", "d_image_embeds"):
                    t = w_.get(k)
                    if t is not None:
                        with ops.operand_format(self.half):
                            ops.scale(t, 1.0 / self.grad_scale, t.numel())

    def _api_backward_begins(self):
        """A backward outside the fused train step ACCUMULATES into the flat gradient buffer: ranges a fused step left un-zeroed
        (its next step would have stored over them) are cleared first; the buffer then holds gradients the fused step must not
        build on (it stores its un-split weight-gradient tiles: TrainStep zeroes when it finds the flag)."""
        if self._g_stale_key is not None:
            self.zero_grad()
        self._g_dirty = True

    def _grads_scaled(self):
        """Context for a backward outside the fused train step when the operand format carries a gradient scale (fp16): the
        flat gradient buffer may hold earlier, un-scaled contributions (gradient accumulation, several encoder passes): it
        is multiplied by the scale before and by its inverse after the backward - exact, a power of two."""
        eng = self

        class _Ctx:
            def __enter__(self_c):
                if eng.grad_scale != 1.0 and eng.params.g is not None:
                    with ops.operand_format(eng.half):
                        ops.scale(eng.params.g, eng.grad_scale, eng.params.n_train)

            def __exit__(self_c, *exc):
                if eng.grad_scale != 1.0 and eng.params.g is not None:
                    with ops.operand_format(eng.half):
                        ops.scale(eng.params.g, 1.0 / eng.grad_scale, eng.params.n_train)
                return False
        return _Ctx()

    def _scaled_in(self, ws, name, t):
        """An externally supplied output gradient (f32) times the gradient scale, in a workspace buffer (identity at 1)."""
        t = t.contiguous()
        if self.grad_scale == 1.0:
            return t
        b = self._buf(ws, name, tuple(t.shape), torch.float32)
        b.copy_(t)
        ops.scale(b.view(-1), self.grad_scale, b.numel())
        return b

    @_in_format
    def _backward(self, grad_scale, dlogits, dpooled, dhidden, after_layer, ws=None, grads_zero=False):
        # grads_zero: the caller vouches that the flat gradient buffer is all zero (TrainStep: the fused optimizer cleared
        # it) - un-split weight-gradient launches may then store instead of accumulate
        self._grads_zero = bool(grads_zero) and os.environ.get("VAULT_WGRAD_STORE", "1") != "0"
        # (element offset, length) of every weight-gradient matrix this backward writes with STORES only (whole matrix covered by
        # un-split launches): the fused optimizer need not zero them for the next step of the same shape (TrainStep)
        self._stored_ranges = []
        ws = self.last if ws is None else ws
        if ws is None or not ws.get("train"):
            raise RuntimeError("backward() needs a preceding forward(train=True)")
        spec, P = self.spec, self.params
        v = spec.vilt
        B, T, S, M, Mp, H, FF, heads, NP = (ws[k] for k in ("B", "T", "S", "M", "Mp", "H", "FF", "heads", "NP"))
        Ml, Mlp = ws["Ml"], ws["Mlp"]
        bf = self.hdt
        buf = lambda name, shape, dtype=torch.float32: self._buf(ws, name, shape, dtype)  # noqa: E731
        self.drop_seed = ws["drop_seed"]
        x = ws["x"]
        nv = v.num_hidden_layers
        if after_layer is not None:
            note = lambda tag: ops.pycall(lambda: after_layer(tag))  # noqa: E731
        else:
            note = lambda tag: None  # noqa: E731

        # deferred weight gradients beside the backward chain when its GEMMs are single partial rounds of tiles (same-box
        # A/B: B = 8 9.76 -> 9.48 ms/step, B = 64 16.12 -> 15.69; B = 256 43.5 -> 43.2: within noise, and concurrent
        # kernels would blur the per-kernel timings the roofline line is built on - serial there)
        e = os.environ.get("VAULT_WGRAD_STREAM")      # development A/B switch
        self._wgrad_side = (e == "1") if e in ("0", "1") else Mp <= self.WGRAD_STREAM_MAX_ROWS
        dx = [buf("dx_a", (Mp, H)), buf("dx_b", (Mp, H))]
        dxb = [buf("dxb_a", (Mp, H), bf), buf("dxb_b", (Mp, H), bf)]
        vbatch = self.LM_WGRAD_BATCHED and "act_all" in ws and P.gr(self.vl[0].fw) is not None
        if vbatch:
            # dY operands of every ViLT layer stay alive until their group's batched weight-gradient launches:
            # A = gradient at the layer output (FFN-out's dY), B = gradient behind the attention block (attn-out's dY)
            dxbA_all = self._stack(ws, "v_dxbA", nv, (Mp, H), bf); dxbB_all = self._stack(ws, "v_dxbB", nv, (Mp, H), bf)
            dU_all = self._stack(ws, "v_dU", nv, (Mp, FF), bf); dqkv_all = self._stack(ws, "v_dqkv", nv, (Mp, 3 * H), bf)
            vgroup = self._wgrad_group_size(nv, after_layer)
        dxb_top = dxbA_all[nv - 1] if vbatch else dxb[0]
        gbf = self.GRAD_STREAM_BF16       # the ViLT residual-gradient stream lives in bf16 only (below)
        if not gbf:
            ops.pycall(dx[0].zero_)
        ops.pycall(dxb_top.zero_)
        # ------------------------------ tail ------------------------------
        if spec.add_pooling_layer and (spec.n_classes > 0 or dpooled is not None):
            Bp = ws["Bp"]
            dpre = buf("dpre", (Bp, H), bf)
            if spec.n_classes > 0 and spec.head == "mlp" and dpooled is None:
                if dlogits is None:
                    raise ValueError("the MLP head has no built-in loss: pass dlogits (the autograd bridge does)")
                ops.tanh_bwd(ws["pooled"], self._mlp_backward(ws, dlogits, B, scale=self.grad_scale), dpre, B * H)
            elif spec.n_classes > 0 and dpooled is None:
                hd = self._drop(self.classifier_dropout, 9001, True)
                gs = ((1.0 / B) if grad_scale is None else grad_scale) * self.grad_scale
                if dlogits is not None:
                    dlogits = self._scaled_in(ws, "dlogits_scaled", dlogits)
                ops.head_bwd(ws["pooled"], ws["logits"], ws.get("labels"), P.w("classifier.1.weight"),
                             P.gr("classifier.1.weight"), P.gr("classifier.1.bias"), dpre, B, H, spec.n_classes, gs,
                             dlogits=dlogits, drop=hd)
            else:
                ops.tanh_bwd(ws["pooled"], self._scaled_in(ws, "dpooled_scaled", dpooled), dpre, B * H)
            self._wgrad(dpre, ws["h0b"], "pooler.dense.weight", "pooler.dense.bias", Bp, H, H, B)
            dh0 = buf("dh0", (Bp, H), bf)
            self._dgrad(dpre, "pooler.dense.weight", dh0, Bp, H, H, ops.EPI_BF16, B)
            ops.layernorm_bwd(x[nv], ws["f_mean"], ws["f_rstd"], P.w("layernorm.weight"), B, H, dy_bf16=dh0,
                              dx_f32=None if gbf else dx[0], dx_bf16=dxb_top, dgamma=P.gr("layernorm.weight"),
                              dbeta=P.gr("layernorm.bias"), xmap=(1, S, 0), dxmap=(1, S, 0),
                              dbias=None if dhidden is not None else P.gr(self.vl[nv - 1].fb))
        if dhidden is not None:
            # gradient w.r.t. last_hidden_state (all rows): LN backward over all rows, added on top
            ops.layernorm_bwd(x[nv], ws["f_mean_all"], ws["f_rstd_all"], P.w("layernorm.weight"), M, H,
                              dy_f32=self._scaled_in(ws, "dhidden_scaled", dhidden).view(M, H), dres=None if gbf else dx[0],
                              dres_bf16=dxb_top if gbf else None,        # (in place: every element is read, then written, by one lane)
                              dx_f32=None if gbf else dx[0], dx_bf16=dxb_top,
                              dgamma=P.gr("layernorm.weight"), dbeta=P.gr("layernorm.bias"),
                              dbias=P.gr(self.vl[nv - 1].fb))
        note("head")

        # ------------------------------ ViLT encoder ------------------------------
        dN = buf("dN", (Mp, H), bf); dctx = buf("dctx", (Mp, H), bf)
        if not vbatch:
            dU = buf("dU", (Mp, FF), bf); dqkv = buf("dqkv", (Mp, 3 * H), bf)
        km = ws["keymask"]
        cur = 0
        # Residual-gradient stream of the pre-LN ViLT stack in bf16 only (GRAD_STREAM_BF16): a layer's incoming gradient is ONE
        # bf16 tensor - stream and FFN-out dY at once -, the LayerNorm backward adds it as `dres_bf16` and writes only the bf16
        # result (10 instead of 16 B per element); the bottom layer also writes f32 for the embedding backward.
        ops.pycall(lambda: self._prof_begin("vilt_bwd"))
        for i in reversed(range(nv)):
            ln = self.vl[i]
            g = lambda k: ws[f"{k}{i}"]  # noqa: E731
            if vbatch:
                dyA, dyB, dU, dqkv = dxbA_all[i], dxbB_all[i], dU_all[i], dqkv_all[i]
                dyN = dxbA_all[i - 1] if i > 0 else dxb[0]
            else:
                dyA, dyB, dyN = dxb[cur], dxb[cur ^ 1], dxb[cur]
            if ws.get("vilt_stage"):
                # the whole layer backward in one C call (csrc/stage.hip: the same kernels in the same order as below)
                nxt = cur ^ 1
                stream_f32 = {} if gbf else dict(dy_f32=dx[cur], dmid_f32=dx[nxt])
                gb = ops.layer_bwd_args(
                    ws[f"stage_vilt{i}"], dy_bf16=dyA, dx_f32=dx[cur] if (not gbf or i == 0) else None, dx_bf16=dyN, dU=dU, dN=dN,
                    dctx=dctx, dqkv=dqkv, dmid_bf16=dyB, do_wgrad=0 if vbatch else 1, **stream_f32,
                    g_wqkv=P.gr(ln.qw, n_elems=3 * H * H, shape=(3 * H, H)),
                    g_bqkv=None if vbatch else P.gr(ln.qb, n_elems=3 * H, shape=(3 * H,)),      # (batched: with the group's launches)
                    g_wo=P.gr(ln.ow), g_bo=P.gr(ln.ob), g_wi=P.gr(ln.iw), g_bi=P.gr(ln.ib), g_wf=P.gr(ln.fw),
                    g_ln1w=P.gr(ln.ln1w), g_ln1b=P.gr(ln.ln1b), g_ln2w=P.gr(ln.ln2w), g_ln2b=P.gr(ln.ln2b),
                    g_bf_below=P.gr(self.vl[i - 1].fb) if i > 0 else None)
                ws[f"stage_vilt_bwd{i}"] = gb
                ops.layer_call("vault_vilt_layer_bwd", gb)
                if not vbatch:
                    note(f"vilt{i}")
                elif i % vgroup == 0:
                    hi = min(nv, i + vgroup)
                    def launch(i=i, hi=hi):
                        self._qkv_bias_grads_batched(dqkv_all, self.vl, i, hi, 3 * H, M, 3 * H)
                        self._wgrad_group(((dxbA_all, ws["act_all"], "fw", H, FF), (dU_all, ws["n2_all"], "iw", FF, H),
                                           (dxbB_all, ws["ctx_all"], "ow", H, H), (dqkv_all, ws["n1_all"], "qw", 3 * H, H)),
                                          self.vl, i, hi, Mp, M)
                    self._wgrads_aside(launch, after_layer)
                    for j in reversed(range(i, hi)):
                        note(f"vilt{j}")
                continue
            # FFN
            # (bias gradients are column sums of dY: fused into the kernel that PRODUCES dY - the LayerNorm
            #  backward for the residual-stream gradient, the GEMM epilogue for dU)
            g8 = ws.get("gelu8_active")
            g8kw = dict(cfg=g8, aux_u8=True) if g8 is not None else {}
            self._dgrad(dyA, ln.fw, dU, Mp, FF, H, ops.EPI_BF16_DGELU, M, aux=g("u"), colsum=P.gr(ln.ib), **g8kw)
            if not vbatch:
                self._wgrad(dyA, g("act"), ln.fw, None, Mp, H, FF, M)
            self._dgrad(dU, ln.iw, dN, Mp, H, FF, ops.EPI_BF16, M)
            if not vbatch:
                self._wgrad(dU, g("n2"), ln.iw, None, Mp, FF, H, M)
            nxt = cur ^ 1
            if gbf:
                ops.layernorm_bwd(g("xm"), g("m2"), g("r2"), P.w(ln.ln2w), M, H, dy_bf16=dN, dres_bf16=dyA, dx_bf16=dyB,
                                  dgamma=P.gr(ln.ln2w), dbeta=P.gr(ln.ln2b), dbias=P.gr(ln.ob))
            else:
                ops.layernorm_bwd(g("xm"), g("m2"), g("r2"), P.w(ln.ln2w), M, H, dy_bf16=dN, dres=dx[cur], dx_f32=dx[nxt],
                                  dx_bf16=dyB, dgamma=P.gr(ln.ln2w), dbeta=P.gr(ln.ln2b), dbias=P.gr(ln.ob))
                cur = nxt
            # attention
            # QKV bias gradient without a pass over all of dqkv (QKV_BIAS_SHORTCUT; the ViLT stack has no attention dropout,
            # D2): softmax rows sum to one, so  sum_keys dV = sum_queries dO  - the value bias gradient is the column sum of
            # dctx, taken in the epilogue of the GEMM that produces dctx; sum_keys dS = 0 for every query, so the key bias
            # gradient is zero (the reference's autograd leaves rounding noise of 1e-9 there); only the query third is summed
            short = vbatch and self.QKV_BIAS_SHORTCUT
            gqb = P.gr(ln.qb, n_elems=3 * H, shape=(3 * H,))
            self._dgrad(dyB, ln.ow, dctx, Mp, H, H, ops.EPI_BF16, M, **(dict(colsum=gqb[2 * H:]) if short else {}))
            if not vbatch:
                self._wgrad(dyB, g("ctx"), ln.ow, None, Mp, H, H, M)
            vhm = ws.get("qkv_hm", 0)
            ops.attention_bwd(g("qkv"), km, g("ctx"), g("lse"), dctx, dqkv, B, S, H, heads, qkv_hm=vhm)
            self._dgrad(dqkv, ln.qw, dN, Mp, H, 3 * H, ops.EPI_BF16, M, **(dict(a_hm=vhm) if vhm else {}))
            if not vbatch:
                self._wgrad(dqkv, g("n1"), ln.qw, ln.qb, Mp, 3 * H, H, M)
            # (vbatch: the query third - or, without the shortcut, all of it - with the group's batched launches below)
            nxt = cur ^ 1
            if gbf:
                ops.layernorm_bwd(x[i], g("m1"), g("r1"), P.w(ln.ln1w), M, H, dy_bf16=dN, dres_bf16=dyB,
                                  dx_f32=dx[cur] if i == 0 else None, dx_bf16=dyN, dgamma=P.gr(ln.ln1w), dbeta=P.gr(ln.ln1b),
                                  dbias=P.gr(self.vl[i - 1].fb) if i > 0 else None)
            else:
                ops.layernorm_bwd(x[i], g("m1"), g("r1"), P.w(ln.ln1w), M, H, dy_bf16=dN, dres=dx[cur], dx_f32=dx[nxt],
                                  dx_bf16=dyN, dgamma=P.gr(ln.ln1w), dbeta=P.gr(ln.ln1b),
                                  dbias=P.gr(self.vl[i - 1].fb) if i > 0 else None)
                cur = nxt
            if not vbatch:
                note(f"vilt{i}")
            elif i % vgroup == 0:
                hi = min(nv, i + vgroup)
                def launch(i=i, hi=hi, short=short, vhm=vhm):
                    self._qkv_bias_grads_batched(dqkv_all, self.vl, i, hi, 3 * H, M, H if short else 3 * H, hm=vhm)
                    self._wgrad_group(((dxbA_all, ws["act_all"], "fw", H, FF), (dU_all, ws["n2_all"], "iw", FF, H),
                                       (dxbB_all, ws["ctx_all"], "ow", H, H), (dqkv_all, ws["n1_all"], "qw", 3 * H, H, vhm)),
                                      self.vl, i, hi, Mp, M)
                self._wgrads_aside(launch, after_layer)
                for j in reversed(range(i, hi)):
                    note(f"vilt{j}")

        ops.pycall(lambda: self._prof_end("vilt_bwd"))
        # ------------------------------ ViLT embeddings ------------------------------
        dx0 = dx[cur]
        Kp, Mpp = ws["Kp"], ws["Mpp"]
        dyp = buf("dyp", (Mpp, H), bf)
        gpos = P.gr("embeddings.position_embeddings", shape=(v.num_patches + 1, H))
        gmt = P.gr("embeddings.token_type_embeddings.weight")
        if ws.get("img_embeds") is not None:
            # externally supplied image embeddings: their gradient (for the caller's autograd) and the modality type's
            die = buf("d_iemb", (_pad(B * NP), H))
            ops.rows_gather_bwd(dx0, die, gmt[ws.get("img_type", 1)], B * NP, H, NP, S, T)
            ws["d_image_embeds"] = die[:B * NP].view(B, NP, H)
        elif ws["ragged"]:
            ops.image_sel_bwd(dx0, gpos, gmt[ws.get("img_type", 1)], P.gr("embeddings.cls_token", shape=(H,)),
                              P.gr("embeddings.patch_embeddings.projection.bias"), dyp, ws["sel"], ws["hw"], B, NP, S, T, H,
                              ws["gw"], v.image_size // v.patch_size)
        else:
            ops.image_rows_bwd(dx0, gpos, gmt[ws.get("img_type", 1)], P.gr("embeddings.cls_token", shape=(H,)),
                               P.gr("embeddings.patch_embeddings.projection.bias"), dyp, NP, H, B, S, T)
        if ws.get("img_embeds") is None:
            self._wgrad(dyp, ws["apatch"], "embeddings.patch_embeddings.projection.weight", None, Mpp, H, Kp, B * NP)
        dvs = buf("d_vt_sum", (Mlp, H))
        # text rows: out = LN(.) + mtype[0]  =>  d mtype[0] = sum dy = THIS backward's d beta: taken through a scratch
        # vector (the gradient buffers accumulate across backward passes: multi-image heads, gradient accumulation)
        dbeta_now = buf("d_vt_beta", (H,))
        ops.pycall(dbeta_now.zero_)
        ops.layernorm_bwd(ws["vt_sum"], ws["vt_mean"], ws["vt_rstd"], P.w("embeddings.text_embeddings.LayerNorm.weight"),
                          Ml, H, dy_f32=dx0, dymap=(T, S, 0), dx_f32=dvs,
                          dgamma=P.gr("embeddings.text_embeddings.LayerNorm.weight"), dbeta=dbeta_now)
        ops.axpy(P.gr("embeddings.text_embeddings.LayerNorm.bias"), dbeta_now, 1.0, H)
        ops.axpy(gmt[0], dbeta_now, 1.0, H)
        tt = ws["tt"]
        gt = [(P.gr("embeddings.text_embeddings.token_type_embeddings.weight"), tt if tt is not None else 0)]
        if spec.lm is None:
            if ws.get("txt_embeds") is not None:
                ws["d_inputs_embeds"] = dvs[:Ml].view(B, T, H)     # inputs_embeds stood in for the word embeddings
            else:
                gt.append((P.gr("embeddings.text_embeddings.word_embeddings.weight"), ws["ids"]))
        if ws["use_pos"]:
            gt.append((P.gr("embeddings.text_embeddings.position_embeddings.weight"), "mod"))
        ops.scatter_add(dvs, gt, Ml, H, period=T)
        note("vilt_embed")
        if spec.lm is None or self.freeze_lm:
            self._join_wgrads()
            self._run_census(ws, "backward")
            return

        # ------------------------------ language model ------------------------------
        lm = spec.lm
        nl = lm.num_hidden_layers
        y, yb = ws["lm_y"], ws["lm_yb"]
        amf = ws["amf"]
        pdh, pda = lm.hidden_dropout_prob, lm.attention_probs_dropout_prob
        dh = buf("lm_dh", (Mlp, H)); dh1 = buf("lm_dh1", (Mlp, H))
        batched = self.LM_WGRAD_BATCHED and "lm_act_all" in ws and P.gr(self.ll[0].fw) is not None
        if batched:
            # dY operands of every layer stay alive until their group's batched weight-gradient launches
            dhb_all = self._stack(ws, "lm_dhb", nl, (Mlp, H), bf); dh1b_all = self._stack(ws, "lm_dh1b", nl, (Mlp, H), bf)
            ldU_all = self._stack(ws, "lm_dU", nl, (Mlp, FF), bf); ldqkv_all = self._stack(ws, "lm_dqkv", nl, (Mlp, 3 * H), bf)
            group = self._wgrad_group_size(nl, after_layer)
        else:
            dhb = buf("lm_dhb", (Mlp, H), bf); dh1b = buf("lm_dh1b", (Mlp, H), bf)
            ldU = buf("lm_dU", (Mlp, FF), bf); ldqkv = buf("lm_dqkv", (Mlp, 3 * H), bf)
        ldN = buf("lm_dN", (Mlp, H), bf); ldctx = buf("lm_dctx", (Mlp, H), bf)
        dyb = None          # bf16 part of d y2 (from the next layer's QKV dgrad)
        dyf = dvs           # f32 part of d y2
        embed_done = False

        def embed_backward(dyb_, dyf_):
            # embeddings: y0 = dropout(LN(esum))
            desum = buf("lm_desum", (Mlp, H))
            ops.layernorm_bwd(ws["lm_esum"], ws["lm_emean"], ws["lm_erstd"], P.w("bert.embeddings.LayerNorm.weight"), Ml, H,
                              dy_bf16=dyb_, dy_f32=dyf_, dx_f32=desum, dgamma=P.gr("bert.embeddings.LayerNorm.weight"),
                              dbeta=P.gr("bert.embeddings.LayerNorm.bias"), drop=self._drop(pdh, 1, True), drop_on_dy=True)
            if ws.get("txt_embeds") is not None:
                ws["d_inputs_embeds"] = desum[:Ml].view(B, T, H)
            ops.scatter_add(desum, [None if ws.get("txt_embeds") is not None else
                                    (P.gr("bert.embeddings.word_embeddings.weight"), ws["ids"]),
                                    (P.gr("bert.embeddings.position_embeddings.weight"), ws["lm_pos"]),
                                    (P.gr("bert.embeddings.token_type_embeddings.weight"), ws["lm_tt"])], Ml, H,
                            rowmask=amf)   # padded positions are masked keys everywhere: their gradient is exactly 0
        ops.pycall(lambda: self._prof_begin("lm_bwd"))
        for i in reversed(range(nl)):
            ln = self.ll[i]
            g = lambda k: ws[f"lm_{k}{i}"]  # noqa: E731
            if batched:
                dhb, dh1b, ldU, ldqkv = dhb_all[i], dh1b_all[i], ldU_all[i], ldqkv_all[i]
            if ws.get("lm_stage"):
                a = ws[f"stage_lm{i}"]
                a.drop_seed = self.drop_seed & 0xFFFFFFFF
                gb = ops.layer_bwd_args(
                    a, dy_bf16=dyb, dy_f32=dyf, dx_f32=dh1, dx_bf16=ldN, dU=ldU, dN=ldN, dctx=ldctx, dqkv=ldqkv,
                    dmid_bf16=dhb, dh1_bf16=dh1b, dmid_f32=dh, do_wgrad=0 if batched else 1,
                    g_wqkv=P.gr(ln.qw, n_elems=3 * H * H, shape=(3 * H, H)),
                    g_bqkv=None if batched else P.gr(ln.qb, n_elems=3 * H, shape=(3 * H,)),
                    g_wo=P.gr(ln.ow), g_bo=P.gr(ln.ob), g_wi=P.gr(ln.iw), g_bi=P.gr(ln.ib), g_wf=P.gr(ln.fw), g_bf=P.gr(ln.fb),
                    g_ln1w=P.gr(ln.ln1w), g_ln1b=P.gr(ln.ln1b), g_ln2w=P.gr(ln.ln2w), g_ln2b=P.gr(ln.ln2b))
                ws[f"stage_lm_bwd{i}"] = gb
                ops.layer_call("vault_lm_layer_bwd", gb, seeded=bool(a.attn_drop_thresh or a.hid_drop_thresh))
                dyb, dyf = ldN, dh1
                if not batched:
                    note(f"lm{i}")
                elif i % group == 0:
                    hi = min(nl, i + group)
                    if i == 0 and after_layer is not None:
                        # data-parallel step: the embedding tables' gradient (a third of the bytes on the wire) first, so
                        # that its all-reduce runs under the last group's weight-gradient launches (train.BucketReducer)
                        embed_backward(dyb, dyf)
                        embed_done = True
                        note("lm_embed")
                    def launch(i=i, hi=hi):
                        self._qkv_bias_grads_batched(ldqkv_all, self.ll, i, hi, 3 * H, Ml, 3 * H)
                        self._wgrad_group(((dhb_all, ws["lm_act_all"], "fw", H, FF), (ldU_all, ws["lm_y1b_all"], "iw", FF, H),
                                           (dh1b_all, ws["lm_ctx_all"], "ow", H, H), (ldqkv_all, ws["lm_yb_all"], "qw", 3 * H, H)),
                                          self.ll, i, hi, Mlp, Ml)
                    self._wgrads_aside(launch, after_layer)
                    for j in reversed(range(i, hi)):
                        note(f"lm{j}")
                continue
            # y2 = LN2(h2)
            ops.layernorm_bwd(g("h2"), g("m2"), g("r2"), P.w(ln.ln2w), Ml, H, dy_bf16=dyb, dy_f32=dyf, dx_f32=dh,
                              dx_bf16=dhb, dgamma=P.gr(ln.ln2w), dbeta=P.gr(ln.ln2b),
                              drop=self._drop(pdh, 16 * i + 4, True), dbias=P.gr(ln.fb))
            self._dgrad(dhb, ln.fw, ldU, Mlp, FF, H, ops.EPI_BF16_DGELU, Ml, aux=g("u"), colsum=P.gr(ln.ib))
            if not batched:
                self._wgrad(dhb, g("act"), ln.fw, None, Mlp, H, FF, Ml)
            self._dgrad(ldU, ln.iw, ldN, Mlp, H, FF, ops.EPI_BF16, Ml)
            if not batched:
                self._wgrad(ldU, g("y1b"), ln.iw, None, Mlp, FF, H, Ml)
            # y1 = LN1(h1) ; d y1 = dgrad(bf16) + dh (residual)
            ops.layernorm_bwd(g("h1"), g("m1"), g("r1"), P.w(ln.ln1w), Ml, H, dy_bf16=ldN, dy_f32=dh, dx_f32=dh1,
                              dx_bf16=dh1b, dgamma=P.gr(ln.ln1w), dbeta=P.gr(ln.ln1b),
                              drop=self._drop(pdh, 16 * i + 3, True), dbias=P.gr(ln.ob))
            self._dgrad(dh1b, ln.ow, ldctx, Mlp, H, H, ops.EPI_BF16, Ml)
            if not batched:
                self._wgrad(dh1b, g("ctx"), ln.ow, None, Mlp, H, H, Ml)
            lhm = ws.get("lm_qkv_hm", 0)
            ops.attention_bwd(g("qkv"), amf, g("ctx"), g("lse"), ldctx, ldqkv, B, T, H, heads,
                              drop=self._drop(pda, 16 * i + 2, True), qkv_hm=lhm)
            self._dgrad(ldqkv, ln.qw, ldN, Mlp, H, 3 * H, ops.EPI_BF16, Ml, **(dict(a_hm=lhm) if lhm else {}))
            if not batched:
                self._wgrad(ldqkv, yb[i], ln.qw, ln.qb, Mlp, 3 * H, H, Ml)
            # (batched: the QKV bias gradient with the group's launches below)
            dyb, dyf = ldN, dh1   # consumed by the next iteration's LN2 backward before being overwritten
            if not batched:
                note(f"lm{i}")
            elif i % group == 0:
                # the weight gradients of layers i .. hi - 1, one launch per kind (dY, X: slices i.. of the stacks)
                hi = min(nl, i + group)
                if i == 0 and after_layer is not None:     # (data-parallel step: embedding gradient first, see above)
                    embed_backward(dyb, dyf)
                    embed_done = True
                    note("lm_embed")
                def launch(i=i, hi=hi, lhm=lhm):
                    self._qkv_bias_grads_batched(ldqkv_all, self.ll, i, hi, 3 * H, Ml, 3 * H, hm=lhm)
                    self._wgrad_group(((dhb_all, ws["lm_act_all"], "fw", H, FF), (ldU_all, ws["lm_y1b_all"], "iw", FF, H),
                                       (dh1b_all, ws["lm_ctx_all"], "ow", H, H), (ldqkv_all, ws["lm_yb_all"], "qw", 3 * H, H, lhm)),
                                      self.ll, i, hi, Mlp, Ml)
                self._wgrads_aside(launch, after_layer)
                for j in reversed(range(i, hi)):
                    note(f"lm{j}")
        ops.pycall(lambda: self._prof_end("lm_bwd"))
        if not embed_done:
            embed_backward(dyb, dyf)
        self._join_wgrads()
        if not embed_done:
            note("lm_embed")
        self._run_census(ws, "backward")
