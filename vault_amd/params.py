"""Flat parameter / gradient / optimizer-state buffers of the HIP engine in HBM (see engine.py for the layout) and the
HF ``state_dict`` names as views of them (ref: vault/models/vault/model.py:92-128 loads / saves by these names)."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from .spec import VaultSpec, build_state, param_entries


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

