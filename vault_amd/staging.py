"""Input staging of the HIP engine: validation (HF-style errors) and the persistent input buffers of a workspace
(ref: the kwargs of vault/models/vault/trainer.py:19-36; HF modeling_vilt.py:585-608 for the errors)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .params import _pad
from .spec import select_patches


class StagingMixin:
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
        adopt = bool(self.adopt_pixel_patches) and pp.is_contiguous() and _pad(B * NP) == B * NP and pp.data_ptr() % 256 == 0 \
            and pp.device == ap.device
        if adopt:
            ws["apatch_in"] = pp.reshape(B * NP, Kp)       # the GEMMs read the caller's tensor (VaultEngine.adopt_pixel_patches)
        else:
            ap[:B * NP].copy_(pp.reshape(B * NP, Kp))      # (onto itself when the caller wrote into input_buffers()["pixel_patches"])
            ws["apatch_in"] = ap
        ws["patch_adopted"] = adopt
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
