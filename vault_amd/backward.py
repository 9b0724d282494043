"""Backward pass of the HIP engine (what autograd does for ref: vault/models/vault/model.py:151-218 under
vault/tmsc_utils/trainer.py:365): the chain of data gradients layer by layer, top down, and the deferred weight
gradients of a stack packed into full rounds of 256 x 256 tiles."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops
from .params import _in_format, _pad


class BackwardMixin:
    # ---- deferred weight gradients on a second stream -------------------------------------------
    def _wgrads_aside(self, launch, after_layer, items=0):
        """Run ``launch()`` (the batched weight-gradient GEMMs of a group of layers) on the engine's second stream when the
        backward chain leaves CUs idle (few token rows: a chain GEMM of a small batch is a single partial round of tiles).
        Nothing in the chain reads the weight gradients: only the optimizer, which waits for the stream (_join_wgrads).
        Not in data-parallel steps (``after_layer``: the reducer starts on the main stream's events)."""
        if not (self._wgrad_side or items > 0) or after_layer is not None:
            launch()
            return
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(self.device)
        side, main, ev = self._wgrad_stream, torch.cuda.current_stream(), torch.cuda.Event()
        ops.pycall(lambda: ev.record(main))
        ops.pycall(lambda: side.wait_event(ev))
        with torch.cuda.stream(side):
            self._wgrad_items = items       # (items per grouped launch of THIS group: _wgrad_group)
            try:
                launch()
            finally:
                self._wgrad_items = 0
        self._wgrad_pending = True

    def _join_wgrads(self):
        if self._wgrad_pending:
            side, main, ev = self._wgrad_stream, torch.cuda.current_stream(), torch.cuda.Event()
            ops.pycall(lambda: ev.record(side))
            ops.pycall(lambda: main.wait_event(ev))
            self._wgrad_pending = False

    def _wgrad(self, dy_bf16, x_bf16, wname, bname, Mtok_pad, Nout, Kin, m_valid, out_rows=0):
        # dW[Nout,Kin] += dY[Mtok,Nout]^T . X[Mtok,Kin] ; db[Nout] += colsum(dY)
        P = self.params
        gw = P.gr(wname, n_elems=Nout * Kin, shape=(Nout, Kin))
        if gw is None:
            return
        nk = Mtok_pad // 64
        # (a WIDE weight with a short contraction - the patch projection's 768 x 3072 at B <= 113: 36 ring tiles x 7 splits fill
        #  the chip; as 144 simple tiles x 3 splits it took 352 us of the B = 64 step, 0.12 PFLOP/s: profiles/r06_small_batch_*)
        wide = Nout % 256 == 0 and Kin % 256 == 0 and (Nout // 256) * (Kin // 256) >= 24
        if Mtok_pad <= 16384 and Nout % 128 == 0 and Kin % 128 == 0 and not wide:
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
        if Nout % 256 == 0 and Kin % 256 == 0:
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
    def _wgrad_group_size(self, n_layers, after_layer, stack="lm"):
        """Layers per deferred weight-gradient group.  A single process takes the whole stack (1,296 tiles = five full rounds +
        16 tiles, against two remainders of 136: B = 256, same box, 40.2 -> 39.8 ms per step; equal at B = 64) - so does a
        step that runs the reducer on ONE rank (VAULT_FORCE_DP: nothing goes on a wire, nothing is there to overlap).  With
        more ranks (``dp_world``, set by TrainStep) the LM stack keeps groups of LM_WGRAD_GROUP layers - the upper group's
        gradient range goes on the wire under the backward of the layers below it - and the ViLT stack stays whole when a
        trained LM stack follows: its whole range (half of the gradient bytes) is exchanged under the LM backward."""
        if after_layer is None or self.dp_world <= 1:
            return n_layers
        if stack == "vilt" and self.spec.lm is not None and not self.freeze_lm:
            return n_layers
        g = self.LM_WGRAD_GROUP
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
        ok = all(no % 256 == 0 and ki % 256 == 0 for *_, no, ki in kinds)
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
        CU = self._wgrad_items or (self.WGRAD_SIDE_ITEMS if self._wgrad_side else 256)
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

    def _qkv_bias_grads_batched(self, dqkv_all, layers, i0, hi, ld, rows, N, hm=0, parts=None):
        """QKV bias gradients of layers i0 .. hi - 1 (column sums over the token rows of the first N columns of their dqkv) in
        ONE launch, issued with the group's batched weight gradients: at small batches a single layer's pass is a 4 us read
        behind a 10 us launch + reduction tail, and next to the weight gradients it is off the backward chain."""
        P = self.params
        offs = [P.offsets[l_.qb][0] for l_ in layers[i0:hi]]
        stride_o = (offs[1] - offs[0]) if len(offs) > 1 else 0
        if any(offs[k + 1] - offs[k] != stride_o for k in range(len(offs) - 1)):
            raise RuntimeError("batched bias gradients need identically laid out layers")
        gqb = P.gr(layers[i0].qb, n_elems=ld, shape=(ld,))
        if parts is not None:      # the attention backward left its workgroups' partial column sums: add the rows up
            part_all, nparts = parts
            ops.colsum_partials(part_all[i0], nparts, N, gqb, hi - i0, part_all.stride(0), stride_o)
            return
        if hm:       # head-major dqkv: plane p = columns 64 p .. 64 p + 63
            ops.colsum_hm(dqkv_all[i0], rows, hm, N // 64, gqb, hi - i0, dqkv_all.stride(0), stride_o)
            return
        ops.colsum_batched(dqkv_all[i0], ld, rows, N, gqb, hi - i0, dqkv_all.stride(0), stride_o)
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
        self._grads_zero = bool(grads_zero)
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
        self._wgrad_side = Mp <= self.WGRAD_STREAM_MAX_ROWS
        dx = [buf("dx_a", (Mp, H))]          # f32 gradient at the bottom of the ViLT stack (the embedding backward reads it)
        dxb = [buf("dxb_a", (Mp, H), bf), buf("dxb_b", (Mp, H), bf)]
        vbatch = self.LM_WGRAD_BATCHED and "act_all" in ws and P.gr(self.vl[0].fw) is not None
        if vbatch:
            # dY operands of every ViLT layer stay alive until their group's batched weight-gradient launches:
            # A = gradient at the layer output (FFN-out's dY), B = gradient behind the attention block (attn-out's dY)
            dxbA_all = self._stack(ws, "v_dxbA", nv, (Mp, H), bf); dxbB_all = self._stack(ws, "v_dxbB", nv, (Mp, H), bf)
            dU_all = self._stack(ws, "v_dU", nv, (Mp, FF), bf); dqkv_all = self._stack(ws, "v_dqkv", nv, (Mp, 3 * H), bf)
            vgroup = self._wgrad_group_size(nv, after_layer, "vilt")
        dxb_top = dxbA_all[nv - 1] if vbatch else dxb[0]
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
                              dx_bf16=dxb_top, dgamma=P.gr("layernorm.weight"),
                              dbeta=P.gr("layernorm.bias"), xmap=(1, S, 0), dxmap=(1, S, 0),
                              dbias=None if dhidden is not None else P.gr(self.vl[nv - 1].fb))
        if dhidden is not None:
            # gradient w.r.t. last_hidden_state (all rows): LN backward over all rows, added on top
            ops.layernorm_bwd(x[nv], ws["f_mean_all"], ws["f_rstd_all"], P.w("layernorm.weight"), M, H,
                              dy_f32=self._scaled_in(ws, "dhidden_scaled", dhidden).view(M, H),
                              dres_bf16=dxb_top,        # (in place: every element is read, then written, by one lane)
                              dx_bf16=dxb_top,
                              dgamma=P.gr("layernorm.weight"), dbeta=P.gr("layernorm.bias"),
                              dbias=P.gr(self.vl[nv - 1].fb))
        note("head")

        # ------------------------------ ViLT encoder ------------------------------
        dN = buf("dN", (Mp, H), bf); dctx = buf("dctx", (Mp, H), bf)
        if not vbatch:
            dU = buf("dU", (Mp, FF), bf); dqkv = buf("dqkv", (Mp, 3 * H), bf)
        # QKV bias gradient from inside the attention backward (vault_attn_args.bias_partials) where the deferred launches would
        # otherwise re-read dqkv for it: the query third (the shortcut below covers key / value: no attention dropout here)
        vparts, vthirds = None, 1
        if vbatch and self.QKV_BIAS_SHORTCUT and not ws.get("vilt_stage"):
            npart = ops.attention_bwd_partials(B, S, H, heads, vthirds)
            if npart:
                vparts = (self._stack(ws, "v_qbpart", nv, (npart, vthirds * H), torch.float32), npart)
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
                gb = ops.layer_bwd_args(
                    ws[f"stage_vilt{i}"], dy_bf16=dyA, dx_f32=dx[cur] if i == 0 else None, dx_bf16=dyN, dU=dU, dN=dN,
                    dctx=dctx, dqkv=dqkv, dmid_bf16=dyB, do_wgrad=0 if vbatch else 1,
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
            ops.layernorm_bwd(g("xm"), g("m2"), g("r2"), P.w(ln.ln2w), M, H, dy_bf16=dN, dres_bf16=dyA, dx_bf16=dyB,
                              dgamma=P.gr(ln.ln2w), dbeta=P.gr(ln.ln2b), dbias=P.gr(ln.ob))
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
            ops.attention_bwd(g("qkv"), km, g("ctx"), g("lse"), dctx, dqkv, B, S, H, heads, qkv_hm=vhm,
                              **(dict(bias_partials=vparts[0][i], bias_thirds=vthirds) if vparts else {}))
            self._dgrad(dqkv, ln.qw, dN, Mp, H, 3 * H, ops.EPI_BF16, M, **(dict(a_hm=vhm) if vhm else {}))
            if not vbatch:
                self._wgrad(dqkv, g("n1"), ln.qw, ln.qb, Mp, 3 * H, H, M)
            # (vbatch: the query third - or, without the shortcut, all of it - with the group's batched launches below)
            ops.layernorm_bwd(x[i], g("m1"), g("r1"), P.w(ln.ln1w), M, H, dy_bf16=dN, dres_bf16=dyB,
                              dx_f32=dx[cur] if i == 0 else None, dx_bf16=dyN, dgamma=P.gr(ln.ln1w), dbeta=P.gr(ln.ln1b),
                              dbias=P.gr(self.vl[i - 1].fb) if i > 0 else None)
            if not vbatch:
                note(f"vilt{i}")
            elif i % vgroup == 0:
                hi = min(nv, i + vgroup)
                def launch(i=i, hi=hi, short=short, vhm=vhm):
                    self._qkv_bias_grads_batched(dqkv_all, self.vl, i, hi, 3 * H, M, H if short else 3 * H, hm=vhm, parts=vparts)
                    self._wgrad_group(((dxbA_all, ws["act_all"], "fw", H, FF), (dU_all, ws["n2_all"], "iw", FF, H),
                                       (dxbB_all, ws["ctx_all"], "ow", H, H), (dqkv_all, ws["n1_all"], "qw", 3 * H, H, vhm)),
                                      self.vl, i, hi, Mp, M)
                # the whole stack's group, launched when the ViLT chain is through, with a trained LM stack's backward next: that
                # chain's GEMMs are partial rounds of tiles at any batch (40 row panels at B = 256) - the group runs beside it
                beside = self.WGRAD_BESIDE_LM_ITEMS if (i == 0 and hi == nv and spec.lm is not None and not self.freeze_lm and not self._wgrad_side) else 0
                self._wgrads_aside(launch, after_layer, items=beside)
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
            # (small batches: beside the chain, like the stack's deferred launches - nothing below reads this gradient)
            self._wgrads_aside(lambda: self._wgrad(dyp, ws["apatch_in"] if ws.get("patches_in") else ws["apatch"],
                                                   "embeddings.patch_embeddings.projection.weight", None,
                                                   Mpp, H, Kp, B * NP), after_layer)
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
        # (the LM has attention dropout: it needs all three thirds of the QKV bias gradient - in-kernel partial sums of three
        #  tiles per wave cost the S <= 64 kernel 11 us per launch against the 10 us per layer of the batched pass over dqkv:
        #  LM_BIAS_PARTIALS stays off; the form is exercised by tests/test_gpu_ops.py)
        lparts = None
        if self.LM_BIAS_PARTIALS and batched and not ws.get("lm_stage"):
            npart = ops.attention_bwd_partials(B, T, H, heads, 3)
            if npart:
                lparts = (self._stack(ws, "lm_qbpart", nl, (npart, 3 * H), torch.float32), npart)
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
                              drop=self._drop(pda, 16 * i + 2, True), qkv_hm=lhm,
                              **(dict(bias_partials=lparts[0][i], bias_thirds=3) if lparts else {}))
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
                    self._qkv_bias_grads_batched(ldqkv_all, self.ll, i, hi, 3 * H, Ml, 3 * H, hm=lhm, parts=lparts)
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

