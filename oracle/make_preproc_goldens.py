"""Generates tests/golden/preproc_*.npz from HuggingFace's ViLT image processor (the callee of
ref: vault/models/vault/dataset.py:337-341) as installed in the build container (transformers 5.15.0, Pillow 12.2.0):
seeded synthetic uint8 images -> SHA-256 of the float32 ``pixel_values`` and of ``pixel_mask`` + sampled values.
Small fixtures: the inputs are stored, the megabytes of output are pinned by their digests.

    python oracle/make_preproc_goldens.py
"""
import hashlib
import os

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def synth_image(rng, h, w):
    """Smooth gradients + a few hard-edged rectangles + noise: exercises antialiasing, clipping and both resize directions."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([128 + 127 * np.sin(xx / (7 + 3 * c) + c) * np.cos(yy / (5 + 2 * c)) for c in range(3)], -1)
    for _ in range(6):
        y0, x0 = rng.integers(0, h), rng.integers(0, w)
        y1, x1 = min(h, y0 + rng.integers(1, max(2, h // 3))), min(w, x0 + rng.integers(1, max(2, w // 3)))
        img[y0:y1, x0:x1] = rng.integers(0, 256, size=3)          # saturated blocks: overshoot beyond 0..255 gets clipped
    if h * w <= 50000:                                            # (noise does not compress: small images only)
        img += rng.normal(0, 6, size=img.shape)
    else:
        img[: h // 8] += rng.normal(0, 6, size=img[: h // 8].shape)
    return np.clip(img, 0, 255).astype(np.uint8)


CASES = {
    # name: list of (h, w)
    "preproc_single_landscape": [(375, 500)],
    "preproc_single_upscale": [(90, 61)],
    "preproc_batch_mixed": [(188, 250), (300, 200), (60, 350), (384, 384), (50, 70), (320, 107)],
}


def main():
    from transformers.models.vilt.image_processing_pil_vilt import ViltImageProcessorPil
    proc = ViltImageProcessorPil()
    for k, (name, sizes) in enumerate(CASES.items()):
        rng = np.random.default_rng(100 + k)
        imgs = [synth_image(rng, h, w) for h, w in sizes]
        out = proc([Image.fromarray(im) for im in imgs], return_tensors="np")
        pv, pm = np.ascontiguousarray(out["pixel_values"]), np.ascontiguousarray(out["pixel_mask"])
        assert pv.dtype == np.float32 and pm.dtype == np.int64
        pos = np.stack([rng.integers(0, s, size=4096) for s in pv.shape], 1)
        data = {f"image_{i}": im for i, im in enumerate(imgs)}
        data.update(n_images=np.int64(len(imgs)), out_shape=np.array(pv.shape, dtype=np.int64),
                    pixel_values_sha256=np.frombuffer(hashlib.sha256(pv.tobytes()).digest(), dtype=np.uint8),
                    pixel_mask_sha256=np.frombuffer(hashlib.sha256(pm.tobytes()).digest(), dtype=np.uint8),
                    sample_pos=pos, sample_val=pv[tuple(pos.T)],
                    valid_hw=np.array([[int(m.sum(0).max()), int(m.sum(1).max())] for m in pm], dtype=np.int64))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
        print(name, pv.shape, sum(v.nbytes for v in data.values()))


if __name__ == "__main__":
    main()
