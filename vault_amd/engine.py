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
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .backward import BackwardMixin
from .heads import HeadsMixin
from .ops import Drop, NO_DROP
from .params import ParamStore, _LayerNames, _in_format, _pad
from .spec import VaultSpec
from .staging import StagingMixin


class VaultEngine(StagingMixin, HeadsMixin, BackwardMixin):
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
    # Encoder layers through the stage-level C ABI (vault_{vilt,lm}_layer_{fwd,bwd}: one C call per layer and direction,
    # the kernel order lives in csrc/stage.hip) when the step is host-launch-bound: up to this many (padded) fused token
    # rows; larger batches keep the per-kernel calls below (same kernels, same order) so that bench.py can bracket single
    # GEMM call sites with events.
    STAGE_MAX_ROWS = 8192
    LM_BIAS_PARTIALS = False       # LM QKV bias gradient from the attention backward's in-kernel partial sums (see backward.py)
    QKV_BIAS_SHORTCUT = True       # ViLT QKV bias gradient: value part from the dctx GEMM's epilogue, key part zero (see _backward)
    GELU8 = True                   # ViLT FFN: gelu' kept for backward as the 8-wave kernel's 8-bit tile-native image (vault_gemm aux_u8)
    GRAD_STREAM_BF16 = True        # (fixed) ViLT residual-gradient stream as ONE 16-bit tensor per layer (what autocast training carries): see _backward
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
        self.vl = [_LayerNames(f"encoder.layer.{i}", "vilt") for i in range(spec.vilt.num_hidden_layers)]
        self.ll = ([_LayerNames(f"bert.encoder.layer.{i}", "bert") for i in range(spec.lm.num_hidden_layers)]
                   if spec.lm else [])
        if with_grads:
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
        self._wgrad_items = 0
        # True: a `pixel_patches` tensor (contiguous, whole padded row count) becomes the patch-embedding GEMM's operand AS IS -
        # forward and weight gradient read the caller's memory, nothing is copied (a recorded step re-points its launches:
        # ops.Tape.rebind).  The caller then keeps the tensor unchanged until the step's kernels have run (an event recorded
        # after the step), e.g. by alternating two tensors; False: the patches are copied into the engine's own buffer
        self.adopt_pixel_patches = False
        self.dp_world = 1          # ranks of the data-parallel step this engine runs in (TrainStep sets it: backward._wgrad_group_size)
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
    def _sk_ws(self):
        """Workspace the library may use to split the contraction of a GEMM that fills less than a round of the chip - the N = H
        Linears with K = FF / 3H at small batches (FFN-out forward, FFN-in / QKV data gradients: 96 tiles of 48 K tiles at B = 32
        become 192 items of 24) - and reduce inside the launch (csrc/gemm256.hip SK).  One per engine: its GEMMs run on one
        stream (the weight gradients of the second stream accumulate by atomics and take none); counters zero at allocation,
        every launch leaves them zero.  64 MiB: two splits of up to 170 tiles, four of 85.  None when switched off."""
        if not self.SPLITK:
            return None
        t = self._ws.get("splitk_ws")
        if t is None:
            t = torch.zeros(self.SPLITK_WS_BYTES, dtype=torch.uint8, device=self.device)
            self._ws["splitk_ws"] = t
        return t

    def _fp8_scratch(self, M, K):
        """MXFP8 image of the A operand of the GEMM about to run (consumed at once: one buffer per shape)."""
        key = ("fp8_a", M, K)
        if key not in self._ws:
            self._ws[key] = (torch.empty(M * K, dtype=torch.uint8, device=self.device),
                             torch.empty(M * (K // 32), dtype=torch.uint8, device=self.device))
        return self._ws[key]

    def _fp8_refresh_weights(self):
        """MXFP8 shadow of the encoder Linear weights, re-quantised from the bf16 shadow on the tape of a train step (every step
        sees the weights the optimizer just wrote): ONE launch per encoder stack over its contiguous range of the flat shadow
        (parameter offsets and sizes are multiples of 64 elements, so a block of 32 never straddles two tensors; the biases,
        LayerNorm vectors and the Linears that stay bf16 in between are quantised along and never read: 2 launches of ~45 us
        instead of 48 of 4.7 us)."""
        P = self.params
        if "fp8_flat" not in self._ws:
            self._ws["fp8_flat"] = (torch.empty(P.n_total, dtype=torch.uint8, device=self.device),
                                    torch.empty(P.n_total // 32, dtype=torch.uint8, device=self.device))
        pq, psc = self._ws["fp8_flat"]
        for stack in (self.ll, self.vl):
            # The two Linears fed by a LayerNorm (K = H), which writes the MXFP8 image of its output itself.  FFN_OUT_FP8: also
            # FFN-out, on the image the FFN-in GEMM's GELU epilogue writes (vault_gemm_args.out_q) - measured in the step at
            # B = 256 and NOT taken (profiles/r05_dev_fp8_8wave_form.txt): the ViLT FFN-out gains 12 us (228 against 240: its f32
            # residual epilogue and 24 K tiles leave little to halve), the image costs the FFN-in epilogue 20 us, the LM stack's
            # FFN-out (dropout: simple kernel) loses 22 us.  Attention-out is bound by its f32 residual epilogue either way
            # (tools/mx8_bench.py: 102 against 101 us) and its operand would need a pass of its own.
            spans = []
            for ln in stack:
                for wname in ((ln.qw, ln.iw, ln.fw) if self.FFN_OUT_FP8 else (ln.qw, ln.iw)):
                    o, shp = P.offsets[wname]
                    N = 3 * shp[0] if wname == ln.qw else shp[0]     # fused QKV: three [H, H] blocks stored back to back
                    K = shp[1]
                    if N % 256 or K % 128:
                        continue
                    spans.append((o, o + N * K))
                    if wname not in self._w8:
                        self._w8[wname] = (pq[o:o + N * K], psc[o // 32:(o + N * K) // 32])
            if spans:
                lo, hi = min(a for a, _ in spans), max(b for _, b in spans)
                ops.quant_mxfp8(P.pb[lo:], (hi - lo) // 64, 64, 64, pq[lo:], psc[lo // 32:])

    @staticmethod
    def _keep_hi(split3: torch.Tensor, plain: torch.Tensor, K: int):
        """plain[:, :] = the `hi` third of a [rows, 3K] = [hi | lo | hi] split-bf16 operand: the bf16 value the fast mode would
        have stored (a strided device copy, no arithmetic) - what the bf16 backward reads in a precise-forward training step."""
        ops.pycall(lambda: plain.copy_(split3[:, :K]))

    def _linear(self, a_bf16, wname, out, M, N, K, epi, m_valid, bias=None, precise=False, ldo=None, prequant=False,
                emit_q=False, **kw):
        """out = epilogue(A . W^T).  ``precise``: A is a [M, 3K] = [hi | lo | hi] split-bf16 operand and the
        weight its [N, 3K] = [hi | hi | lo] counterpart: the same kernel over a 3x longer contraction.
        fp8-forward mode: ``prequant`` - the MXFP8 image of A already lies in the (M, K) scratch (False for a Linear whose
        producer could not write it: that Linear then runs on bf16 operands unless ``prequant`` is None = quantise here);
        ``emit_q`` - this GEMM's epilogue also writes the MXFP8 image of its 16-bit output into the (M, N) scratch when the
        kernel can (asked from the library); returns whether it did."""
        P = self.params
        if precise:
            ops.gemm(a_bf16, P.wb3(wname, N, K), out, M, N, 3 * K, 3 * K, 3 * K, N if ldo is None else ldo, 0, 0, epi,
                     m_valid=m_valid, bias=bias, **kw)
        elif (self.fp8_forward and wname in self._w8 and M % 256 == 0 and prequant is not False
              and self._fp8_kernel_ok(M, N, K, epi, out, bias, kw)):
            wq, wsc = self._w8[wname]
            aq, asc = self._fp8_scratch(M, K)
            if prequant is None:      # (no producer wrote the image)
                ops.quant_mxfp8(a_bf16, M, K, K, aq, asc)
            kw.pop("split3", None)    # (precise mode only)
            if emit_q:
                oq, osc = self._fp8_scratch(M, N)
                key = ("emit_q", M, N, K, epi, kw.get("cfg", -1), bool(kw.get("aux_u8")))
                if key not in self._ws:
                    self._ws[key] = ops.gemm_mxfp8(aq, asc, wq, wsc, out, M, N, K, N, epi, m_valid=m_valid, bias=bias, out_q=oq,
                                                   out_scale=osc, plan_only=True, **{k: v for k, v in kw.items() if k != "drop"}) == 5
                emit_q = self._ws[key]
                if emit_q:
                    kw.update(out_q=oq, out_scale=osc)
            ops.gemm_mxfp8(aq, asc, wq, wsc, out, M, N, K, N, epi, m_valid=m_valid, bias=bias, **kw)
            return emit_q
        else:
            if epi == ops.EPI_F32_RES and K >= 1536:
                kw.setdefault("splitk_ws", self._sk_ws())
            ops.gemm(a_bf16, P.wb(wname, n_elems=N * K, shape=(N, K)), out, M, N, K, K, K, N if ldo is None else ldo,
                     0, 0, epi, m_valid=m_valid, bias=bias, **kw)

    def _fp8_kernel_ok(self, M, N, K, epi, out, bias, kw) -> bool:
        """The Linears fed by a LayerNorm always take the MXFP8 path; one whose operand image comes out of another GEMM's epilogue
        (FFN-out, FFN_OUT_FP8) only where the 8-wave form takes the call - the simple kernel is slower than the bf16 ring kernel
        there (LM stack in training: dropout in the residual epilogue)."""
        if epi != ops.EPI_F32_RES:
            return True
        drop = kw.get("drop", NO_DROP)
        key = ("fp8_ok", M, N, K, epi, bool(drop.thresh))
        if key not in self._ws:
            self._ws[key] = ops.gemm_mxfp8(out, out, out, out, out, M, N, K, N, epi, bias=bias, res=kw.get("res"), drop=drop,
                                           plan_only=True) in (5, 6)
        return self._ws[key]

    def _dgrad(self, dy_bf16, wname, out, M, Kin, Nout, epi, m_valid, **kw):
        # dX[M,Kin] = dY[M,Nout] . W[Nout,Kin]
        P = self.params
        wt = P.pbT.get(wname)
        if wt is not None and M % 256 == 0:
            # forward-form operands on the transposed shadow W^T [Kin][Nout]: the register-direct GEMM
            ops.gemm(dy_bf16, wt, out, M, Kin, Nout, Nout, Nout, Kin, 0, 0, epi, m_valid=m_valid, **kw)
            return
        if epi == ops.EPI_BF16 and Nout >= 1536:
            kw.setdefault("splitk_ws", self._sk_ws())
        ops.gemm(dy_bf16, P.wb(wname, n_elems=Nout * Kin, shape=(Nout, Kin)), out, M, Kin, Nout, Nout, Kin, Kin, 0, 1,
                 epi, m_valid=m_valid, **kw)


    FFN_OUT_FP8 = False            # fp8-forward mode: FFN-out on MXFP8 operands too (see _fp8_refresh_weights: measured, slower)
    HEAD_MAJOR = True              # qkv / dqkv of large batches in the head-major layout [3][heads][rows][64] (see _plan_head_major)
    HEAD_MAJOR_MIN_ROWS = 16384    # ... from this many (padded) token rows of a stack
    SPLITK = True                  # lend the GEMMs a workspace for split-K with the in-launch reduction (vault_gemm_args.splitk_ws; see _sk_ws)
    SPLITK_WS_BYTES = 16384 + (64 << 20)
    # items per grouped launch of the ViLT stack's weight gradients when they run BESIDE the LM backward on the second stream
    # (0: in front of it, serial).  Measured at B = 256, one box (profiles/r06_dev_wgrads_beside_lm.txt): 37.85 ms serial, 37.66 /
    # 37.79 / 37.58 / 37.57 / 38.17 with 256 / 224 / 192 / 160 / 128 items - the LM chain's launches keep their CUs busy (86 % of
    # their CU time is tile time, not idle CUs), so there is little to fill; and the shared CUs stretch the weight-gradient
    # launches the `roofline` line times (0.53 -> 0.40 of peak as measured by the stream's events).  Off.
    WGRAD_BESIDE_LM_ITEMS = 0
    WGRAD_SIDE_ITEMS = 224         # items per grouped launch on the second stream (B = 64, same box: 256: 13.50 / 13.59 ms, 224: 13.32 / 13.46, 192: 13.23 / 13.47, 160: 13.63 / 13.53)


    def _plan_gelu8(self, ws, n2, act, u, ln, Mp, M):
        """Kernel configuration (5 / 6) on which BOTH the FFN-in forward and the gelu'-product data gradient of this ViLT
        workspace run with the 8-bit tile-native gelu' (vault_gemm aux_u8: an opaque image only the same kernel and shape reads
        back), or None: asked from the library once per workspace (vault_gemm_plan) - the automatic kernel choice must land on
        the 8-wave kernel with equal tile width for both (not in data-parallel steps or small batches; fp8-forward: the forward
        half of the pair is the 8-wave kernel's MXFP8 form, vault_gemm_mxfp8_plan)."""
        mode = (bool(self.fp8_forward), ops.GEMM_SCHED, self.GELU8)
        if ws.get("gelu8_mode") == mode:
            return ws["gelu8_cfg"]
        ws["gelu8_mode"] = mode
        P, H, FF = self.params, ws["H"], ws["FF"]
        cfg, wt = None, P.pbT.get(ln.fw)
        if self.GELU8 and u is not None and wt is not None and Mp % 256 == 0:
            if self.fp8_forward:
                c1 = (ops.gemm_mxfp8(n2, n2, n2, n2, act, Mp, FF, H, FF, ops.EPI_BF16_GELU, m_valid=M, bias=P.w(ln.ib), out2=u,
                                     aux_u8=True, plan_only=True) if ln.iw in self._w8 else -1)
            else:
                c1 = ops.gemm(n2, P.wb(ln.iw, n_elems=FF * H, shape=(FF, H)), act, Mp, FF, H, H, H, FF, 0, 0,
                              ops.EPI_BF16_GELU, m_valid=M, bias=P.w(ln.ib), out2=u, aux_u8=True, plan_only=True)
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
        library is asked about once per workspace (vault_gemm_plan / vault_gemm_mxfp8_plan); small batches, the precise mode and
        the stage-level calls keep the row-major layout."""
        mode = (bool(self.fp8_forward), ops.GEMM_SCHED, bool(pr), bool(train), self.HEAD_MAJOR)
        if ws.get(key + "_mode") == mode:
            return ws[key]
        ws[key + "_mode"] = mode
        P, H, FF = self.params, ws["H"], ws["FF"]
        hm = 0
        if (self.HEAD_MAJOR and not pr and S <= 192 and rows_pad % 256 == 0 and H % 256 == 0 and FF % 256 == 0
                and self.HEAD_MAJOR_MIN_ROWS <= rows_pad <= self.WGRAD_BATCH_MAX_ROWS):
            w = P.wb(wname, n_elems=3 * H * H, shape=(3 * H, H))
            # (plan only: the pointers are not dereferenced, but must not be null)
            qb = P.w(wname.replace("weight", "bias"), n_elems=3 * H, shape=(3 * H,))
            if self.fp8_forward:
                c1 = (ops.gemm_mxfp8(a16, a16, a16, a16, a16, rows_pad, 3 * H, H, 3 * H, ops.EPI_BF16, m_valid=rows, bias=qb,
                                     out_hm=rows_pad, plan_only=True) if wname in self._w8 else -1)
            else:
                c1 = ops.gemm(a16, w, a16, rows_pad, 3 * H, H, H, H, 3 * H, 0, 0, ops.EPI_BF16, m_valid=rows, bias=qb,
                              out_hm=rows_pad, plan_only=True)
            ok = c1 in (5, 6)
            if ok and train:
                c2 = ops.gemm(a16, w, a16, rows_pad, H, 3 * H, 3 * H, H, H, 0, 1, ops.EPI_BF16, m_valid=rows, a_hm=rows_pad,
                              plan_only=True)
                ok = c2 in (3, 4, 8) and self.LM_WGRAD_BATCHED and P.gr(wname) is not None
            if ok:
                hm = rows_pad
        ws[key] = hm
        return hm

    def _use_stage(self, rows_pad: int, pr: bool) -> bool:
        if pr or self.fp8_forward:
            return False
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
        sk = self._sk_ws()
        if sk is not None:
            kw.update(splitk_ws=sk, splitk_bytes=sk.numel())
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
                actq = self._linear(y1b, ln.iw, act, Mlp, FF, H, ops.EPI_BF16_GELU, Ml, bias=P.w(ln.ib), out2=u_out, precise=pr,
                                    split3=pr, ldo=W3 * FF, prequant=q8l[0] is not None or None, emit_q=self.FFN_OUT_FP8)
                if pt:
                    self._keep_hi(act, buf(f"lm_act{sfx}", (Mlp, FF), bf), FF)
                    if u is not None:
                        self._keep_hi(u_out, u, FF)
                fl_l = 2.0 * Ml * FF * H * W3
                ops.pycall(lambda: self._prof_end("ffn1", fl_l))
                self._linear(act, ln.fw, h2, Mlp, H, FF, ops.EPI_F32_RES, Ml, bias=P.w(ln.fb), res=y1,
                             drop=self._drop(pdh, 16 * i + 4, lm_train), precise=pr, prequant=bool(actq))
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
        if ws.get("patches_in"):
            apatch = ws["apatch_in"]          # (the engine's own buffer or the caller's tensor: staging._stage_pixel_patches)
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
                # the f32 pixel staging buffer (input_buffers()["pixel_values"]) is free from here on: nothing later in the step
                # reads it (the weight gradient contracts the unfolded operand).  A loader that writes its host -> device copy
                # straight into that buffer waits for this event on its copy stream (bench.py, with_h2d_input_copies): the next
                # batch's pixels then travel under the rest of THIS step, without a device-to-device restage
                ev = ws.get("pixels_consumed")
                if ev is None:
                    ev = ws["pixels_consumed"] = torch.cuda.Event()
                ops.pycall(lambda ev=ev: ev.record(torch.cuda.current_stream()))
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
            actq = self._linear(n2, ln.iw, act, Mp, FF, H, ops.EPI_BF16_GELU, M, bias=P.w(ln.ib), out2=u_out, precise=pr,
                                split3=pr, ldo=W3 * FF, prequant=q8[0] is not None or None, emit_q=self.FFN_OUT_FP8, **g8kw)
            if pt:
                self._keep_hi(act, buf(f"act{sfx}", (Mp, FF), bf), FF)
                if u is not None:
                    self._keep_hi(u_out, u, FF)
            fl_v = 2.0 * M * FF * H * W3
            ops.pycall(lambda: self._prof_end("ffn1", fl_v))
            self._linear(act, ln.fw, x[i + 1], Mp, H, FF, ops.EPI_F32_RES, M, bias=P.w(ln.fb), res=xm, precise=pr,
                         prequant=bool(actq))

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

