"""Device-side image preprocessing: a drop-in for the image half of ``ViltProcessor`` /
``tokenizer.feature_extractor(image, return_tensors="pt")`` as the reference calls it per item
(ref: vault/models/vault/dataset.py:337-341), for whole batches of differently sized uint8 images.

The host plans - output sizes (HF:models/vilt/image_processing_pil_vilt.py:70-98) and, per image and axis, the taps of
Pillow's antialiased bicubic filter (``Resample.c`` ``precompute_coeffs`` / ``normalize_coeffs_8bpc``, evaluated here in
float64 with the same operation order) - the GPU does the byte work (csrc/preprocess.hip: ``vault_image_preprocess``).
The result equals the HuggingFace processor's bit for bit (tests/test_gpu_preprocess.py against goldens generated from it).
"""
import ctypes as C
from functools import lru_cache
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L

PRECISION_BITS = 22


class ImageDesc(C.Structure):
    """vault_image_desc (include/vault_hip.h)."""
    _fields_ = [("src_off", C.c_longlong), ("tmp_off", C.c_longlong)] + [
        (n, C.c_int) for n in ("h_in", "w_in", "h_out", "w_out", "ksize_h", "ksize_v", "hb_off", "hk_off", "vb_off", "vk_off")]


class PreprocessArgs(C.Structure):
    """vault_preprocess_args (include/vault_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("src", "tmp", "plan", "desc", "lut", "pixel_values", "pixel_mask", "pixel_mask_f32")] + [
        (n, C.c_int) for n in ("B", "H", "W", "max_h_in", "max_w_out", "max_w_in")] + [("src_bytes", C.c_longlong),
                                                                                        ("patch_unfold_bf16", C.c_void_p), ("ps", C.c_int),
                                                                                        ("ksize_max", C.c_int), ("band_rows_max", C.c_int)]


def resize_output_size(h: int, w: int, shorter: int = 384, size_divisor: int = 32) -> Tuple[int, int]:
    """Shorter side -> ``shorter``, longer side capped at int(1333 / 800 * shorter), both floored to multiples of
    ``size_divisor`` (HF:models/vilt/image_processing_pil_vilt.py:70-98,150)."""
    longer = int(1333 / 800 * shorter)
    s = shorter / min(h, w)
    nh, nw = (shorter, s * w) if h < w else (s * h, shorter)
    if max(nh, nw) > longer:
        s = longer / max(nh, nw)
        nh, nw = s * nh, s * nw
    nh, nw = int(nh + 0.5), int(nw + 0.5)
    return nh // size_divisor * size_divisor, nw // size_divisor * size_divisor


@lru_cache(maxsize=4096)
def resample_taps(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """(bounds int32 [out][2] = first tap / tap count, weights int32 [out][ksize]) of Pillow's 8-bit bicubic resampling."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(np.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # C's (int) truncation; arguments are > -1
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    k = np.arange(ksize, dtype=np.float64)[None, :]
    x = np.abs((k + xmin[:, None] - center[:, None] + 0.5) * (1.0 / fscale))
    a = -0.5
    w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))
    w = np.where(np.arange(ksize)[None, :] < n[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):                                                  # sequential sum, like the C loop
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    q = np.trunc(np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))).astype(np.int32)
    bounds = np.stack([xmin, n], 1).astype(np.int32)
    # columns past the widest window hold zeros only (Pillow's ksize is an upper bound: 480 -> 384 has ksize 7 and 5-tap
    # windows): the table keeps the widest window's width - it is the kernels' row stride AND the one-launch kernel's tap count
    return bounds, np.ascontiguousarray(q[:, :max(int(n.max()), 1)])


BAND_ROWS = 32          # output rows one workgroup of the one-launch form resamples (csrc/preprocess.hip FB_ROWS)


@lru_cache(maxsize=4096)
def band_source_rows(in_size: int, out_size: int) -> int:
    """The most source rows one band of BAND_ROWS output rows reads (vault_preprocess_args.band_rows_max for one image)."""
    b, _ = resample_taps(in_size, out_size)
    first = b[0::BAND_ROWS, 0]
    last = np.minimum(np.arange(0, out_size, BAND_ROWS) + BAND_ROWS, out_size) - 1
    return int(np.max(b[last, 0] + b[last, 1] - first))


def normalise_lut(rescale_factor: float, mean: Sequence[float], std: Sequence[float]) -> np.ndarray:
    """[3][256] float32: the value the HF processor produces for each 8-bit level (rescale in float64, cast to float32,
    normalise in float32; HF:image_transforms.py:118-122,417-439)."""
    v = (np.arange(256).astype(np.float64) * rescale_factor).astype(np.float32)
    return np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)


def _as_hwc_u8(img) -> np.ndarray:
    if isinstance(img, torch.Tensor):
        img = img.cpu().numpy()
    if not isinstance(img, np.ndarray):               # PIL image
        img = np.asarray(img.convert("RGB") if getattr(img, "mode", "RGB") != "RGB" else img)
    if img.dtype != np.uint8 or img.ndim != 3:
        raise TypeError("images must be 8-bit RGB: PIL images or [H][W][3] (or [3][H][W]) uint8 arrays")
    if img.shape[2] != 3 and img.shape[0] == 3:
        img = img.transpose(1, 2, 0)
    if img.shape[2] != 3:
        raise ValueError("images need three channels")
    return np.ascontiguousarray(img)


class DeviceImageProcessor:
    """``processor(images, return_tensors="pt")`` -> ``{"pixel_values": [B,3,H,W] float32, "pixel_mask": [B,H,W] int64}`` on
    the GPU, equal to ``ViltImageProcessor``'s output.  Keyword defaults are the dandelin/vilt-b32 checkpoints' settings."""

    model_input_names = ["pixel_values", "pixel_mask"]

    def __init__(self, device="cuda:0", shortest_edge: int = 384, size_divisor: int = 32, rescale_factor: float = 1 / 255,
                 image_mean: Sequence[float] = (0.5, 0.5, 0.5), image_std: Sequence[float] = (0.5, 0.5, 0.5),
                 mask_dtype: torch.dtype = torch.int64, fused: bool = True):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceImageProcessor needs a GPU (the HIP library has no CPU path)")
        L.load()
        if size_divisor % 4:
            raise ValueError("size_divisor must be a multiple of 4 (the kernels write four pixels per thread)")
        self.device = torch.device(device)
        self.shortest_edge, self.size_divisor = shortest_edge, size_divisor
        self.mask_dtype = mask_dtype
        # one launch with the 8-bit intermediate in LDS when a band of the batch fits (csrc/preprocess.hip resize_fused_kernel);
        # False: always the two-pass form (intermediate in HBM) - same bytes out, for comparison
        self.fused = fused
        self._lut = torch.from_numpy(normalise_lut(rescale_factor, image_mean, image_std)).to(self.device)
        self._plan_key, self._plan_dev = None, None        # device-side plan of the last batch geometry (loaders repeat it)

    def plan(self, sizes: List[Tuple[int, int]]):
        """Host plan of a batch: (descriptor bytes, plan int32 array, src bytes, tmp bytes, H, W, max_h_in, max_w_out,
        ksize_max, band_rows_max) - the last two for the one-launch form (vault_preprocess_args, ABI 12)."""
        descs = (ImageDesc * len(sizes))()
        parts: List[np.ndarray] = []
        off = 0

        def put(a: np.ndarray) -> int:
            nonlocal off
            o = off
            parts.append(a.reshape(-1))
            off += a.size
            return o

        cache: Dict[Tuple[int, int], Tuple[int, int, int]] = {}
        src_off = tmp_off = 0
        H = W = max_h_in = max_w_out = ksize_max = band_rows_max = 0
        taps_fit = True       # every tap within the one-launch kernel's 24-bit multiplier (weights are below 2 in magnitude: always)
        for i, (h, w) in enumerate(sizes):
            oh, ow = resize_output_size(h, w, self.shortest_edge, self.size_divisor)
            if oh <= 0 or ow <= 0:
                raise ValueError(f"image {i} ({h} x {w}) resizes to an empty image")
            d = descs[i]
            d.src_off, d.tmp_off, d.h_in, d.w_in, d.h_out, d.w_out = src_off, tmp_off, h, w, oh, ow
            for axis, (n_in, n_out) in (("h", (w, ow)), ("v", (h, oh))):
                if (n_in, n_out) not in cache:
                    b, q = resample_taps(n_in, n_out)
                    taps_fit = taps_fit and int(np.abs(q).max()) < (1 << 23)
                    cache[(n_in, n_out)] = (put(b), put(q), q.shape[1])
                bo, ko, ks = cache[(n_in, n_out)]
                ksize_max = max(ksize_max, ks)
                if axis == "h":
                    d.hb_off, d.hk_off, d.ksize_h = bo, ko, ks
                else:
                    d.vb_off, d.vk_off, d.ksize_v = bo, ko, ks
            band_rows_max = max(band_rows_max, band_source_rows(h, oh))
            src_off += h * w * 3
            tmp_off += h * ((ow * 3 + 3) & ~3)            # intermediate rows padded to 4-byte multiples
            H, W, max_h_in, max_w_out = max(H, oh), max(W, ow), max(max_h_in, h), max(max_w_out, ow)
        if off >= 2 ** 31:
            raise ValueError("plan too large")
        if not taps_fit:
            ksize_max = band_rows_max = 0                  # (unknown maxima: the two-pass form, 32-bit multiplies)
        return bytes(descs), np.concatenate(parts).astype(np.int32), src_off, tmp_off, H, W, max_h_in, max_w_out, ksize_max, band_rows_max

    def __call__(self, images, return_tensors: str = "pt", **unused) -> Dict[str, torch.Tensor]:
        if return_tensors != "pt":
            raise ValueError("the device processor returns torch tensors on the GPU (return_tensors='pt')")
        if not isinstance(images, (list, tuple)):
            images = [images]
        imgs = [_as_hwc_u8(im) for im in images]
        host = torch.empty(sum(im.size for im in imgs), dtype=torch.uint8, pin_memory=True)
        hv = host.numpy()
        o = 0
        for im in imgs:
            hv[o:o + im.size] = im.reshape(-1)
            o += im.size
        out = self.from_packed(host, [im.shape[:2] for im in imgs])
        return {"pixel_values": out["pixel_values"], "pixel_mask": out["pixel_mask"]}   # (the HF processor's keys only)

    def from_packed(self, host_u8: torch.Tensor, sizes, out: Dict[str, torch.Tensor] = None,
                    patch_out: Optional[torch.Tensor] = None, patch_size: int = 32, want_mask: bool = True) -> Dict[str, torch.Tensor]:
        """The loader-facing form: ``host_u8`` = the images back to back ([h][w][3] uint8 each, ``sizes`` = their (h, w)) in ONE
        host tensor - pinned, as a decoder writing straight into a staging buffer leaves them - copied and processed on the
        current stream.  ``out``: optional preallocated ``pixel_values`` / ``pixel_mask`` of the batch's padded shape (e.g. the
        engine's own input staging buffers: no device-to-device copy afterwards).  ``patch_out``: a bf16
        [B * (H / ps) * (W / ps), 3 ps ps] tensor that receives the patch-embedding GEMM's operand (the unfold of the
        padded canvas) straight from the resize kernel; the f32 ``pixel_values`` tensor is then not written at all
        (the returned dict carries ``pixel_patches`` instead): hand that to the engine as ``batch["pixel_patches"]``.
        ``want_mask=False``: the pixel mask is not written (``valid_hw`` carries the same information from the host; a
        [B, H, W] int64 mask is 302 MB of writes per 256 images)."""
        sizes = [tuple(int(v) for v in hw) for hw in sizes]
        B = len(sizes)
        key = tuple(sizes)
        if self._plan_key != key:
            desc_b, plan, src_bytes, tmp_bytes, H, W, max_h_in, max_w_out, ksize_max, band_rows_max = self.plan(sizes)
            dev = self.device
            if not self.fused:
                ksize_max = band_rows_max = 0               # (unknown maxima: the library runs the two-pass form)
            probe = PreprocessArgs()
            probe.max_w_in, probe.max_w_out = max(w for _, w in sizes), max_w_out
            probe.ksize_max, probe.band_rows_max = ksize_max, band_rows_max
            one_launch = bool(L.load().vault_image_preprocess_is_fused(C.byref(probe)))
            self._plan_key, self._plan_dev = key, (
                torch.from_numpy(plan).to(dev), torch.frombuffer(bytearray(desc_b), dtype=torch.uint8).to(dev),
                # the 8-bit intermediate of the two-pass form (the one-launch form keeps it in LDS)
                None if one_launch else torch.empty(tmp_bytes, dtype=torch.uint8, device=dev),
                src_bytes, H, W, max_h_in, max_w_out, ksize_max, band_rows_max)
        plan_d, desc_d, tmp, src_bytes, H, W, max_h_in, max_w_out, ksize_max, band_rows_max = self._plan_dev
        if host_u8.dtype != torch.uint8 or host_u8.numel() != src_bytes:
            raise ValueError("host_u8 must hold exactly the images of `sizes`, uint8")
        dev = self.device
        with torch.cuda.device(dev):
            src = host_u8.to(dev, non_blocking=True)
            pv = None
            if patch_out is None:
                pv = out["pixel_values"] if out is not None else torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
            else:
                ps = int(patch_size)
                if H % ps or W % ps or ps % 4 or patch_out.dtype not in (torch.bfloat16, torch.float16) or not patch_out.is_contiguous() or \
                        patch_out.numel() < B * (H // ps) * (W // ps) * 3 * ps * ps:
                    raise ValueError(f"patch_out must be contiguous bf16 / fp16 with >= {B * (H // ps) * (W // ps)} rows of {3 * ps * ps}")
            pm = None
            if want_mask:
                pm = out["pixel_mask"] if out is not None else torch.empty(B, H, W, dtype=self.mask_dtype, device=dev)
            if (pv is not None and (tuple(pv.shape) != (B, 3, H, W) or pv.dtype != torch.float32)) or \
                    (pm is not None and (tuple(pm.shape) != (B, H, W) or pm.dtype != self.mask_dtype)):
                raise ValueError(f"out tensors must be pixel_values [{B},3,{H},{W}] float32 and pixel_mask [{B},{H},{W}] {self.mask_dtype}")
            if self.mask_dtype not in (torch.int64, torch.float32):
                raise ValueError("mask_dtype must be torch.int64 (HF) or torch.float32")
            a = PreprocessArgs()
            a.src, a.tmp, a.plan, a.desc, a.lut, a.pixel_values = (src.data_ptr(), None if tmp is None else tmp.data_ptr(), plan_d.data_ptr(), desc_d.data_ptr(),
                                                                   self._lut.data_ptr(), None if pv is None else pv.data_ptr())
            if patch_out is not None:
                a.patch_unfold_bf16, a.ps = patch_out.data_ptr(), int(patch_size)
            if pm is not None and self.mask_dtype == torch.int64:
                a.pixel_mask = pm.data_ptr()
            elif pm is not None:
                a.pixel_mask_f32 = pm.data_ptr()
            a.ksize_max, a.band_rows_max = ksize_max, band_rows_max
            a.B, a.H, a.W, a.max_h_in, a.max_w_out = B, H, W, max_h_in, max_w_out
            a.max_w_in, a.src_bytes = max(w for _, w in sizes), src_bytes
            # (the unfold is written in patch_out's own 16-bit format: by the library built for it)
            fmt = "fp16" if (patch_out is not None and patch_out.dtype == torch.float16) else "bf16"
            L.check(L.load(fmt).vault_image_preprocess(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "vault_image_preprocess")
            # (src is freed on this stream after the launches; plan / descriptors / intermediate belong to the cached plan: a
            #  batch of other sizes on ANOTHER stream must not start before this one has passed them)
            src.record_stream(torch.cuda.current_stream())
        # valid_hw: the resized (h, w) of every image on the padded canvas, known on the host - passed on to the engine
        # (batch["valid_hw"]) it replaces the device -> host read of the pixel mask in the padded-image path
        valid_hw = [resize_output_size(h, w, self.shortest_edge, self.size_divisor) for h, w in sizes]
        if patch_out is not None:
            return {"pixel_patches": patch_out, "pixel_mask": pm, "canvas": (H, W), "valid_hw": valid_hw}
        return {"pixel_values": pv, "pixel_mask": pm, "valid_hw": valid_hw}
