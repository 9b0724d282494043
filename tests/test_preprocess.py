"""CPU tests of the image-preprocessing oracle and of the host-side planner (ref: vault/models/vault/dataset.py:
337-341 -> HF ViltImageProcessor): the oracle reproduces the HuggingFace processor's output bit for bit (SHA-256 digests in
tests/golden/preproc_*.npz, made by oracle/make_preproc_goldens.py), and the product's planner gives Pillow's taps."""
import hashlib
import itertools
import os

import numpy as np
import pytest

from oracle import preprocess_oracle as PO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["preproc_single_landscape", "preproc_single_upscale", "preproc_batch_mixed"]


def load_case(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    return g, [g[f"image_{i}"] for i in range(int(g["n_images"]))]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_hf_processor_bit_for_bit(name):
    g, imgs = load_case(name)
    pv, pm = PO.preprocess(imgs)
    assert pv.shape == tuple(int(v) for v in g["out_shape"]) and pv.dtype == np.float32 and pm.dtype == np.int64
    assert hashlib.sha256(pv.tobytes()).digest() == g["pixel_values_sha256"].tobytes()
    assert hashlib.sha256(pm.tobytes()).digest() == g["pixel_mask_sha256"].tobytes()
    assert np.array_equal(pv[tuple(g["sample_pos"].T)], g["sample_val"])
    assert np.array_equal(np.array([[int(m.sum(0).max()), int(m.sum(1).max())] for m in pm]), g["valid_hw"])


def test_host_planner_gives_pillows_taps_and_hf_sizes():
    from vault_amd.preprocess import normalise_lut, resample_taps, resize_output_size
    for n_in, n_out in [(500, 512), (375, 384), (90, 544), (61, 384), (1000, 608), (333, 192), (384, 384), (50, 384), (70, 512),
                        (700, 608), (2000, 384), (3, 384), (1, 32)]:
        bounds, q = resample_taps(n_in, n_out)
        assert q.dtype == np.int32 and bounds.shape == (n_out, 2)
        for xx, (xmin, n, kk) in enumerate(PO._coeffs(n_in, n_out)):
            assert (bounds[xx, 0], bounds[xx, 1]) == (xmin, n)
            assert np.array_equal(q[xx, :n], kk) and not q[xx, n:].any()
    for h, w in itertools.product([1, 17, 50, 384, 385, 500, 999, 1333, 4000], [1, 33, 384, 640, 641, 1000]):
        assert resize_output_size(h, w) == PO.output_size(h, w)
    assert np.array_equal(normalise_lut(1 / 255, (0.5,) * 3, (0.5,) * 3)[1], PO.normalise_lut())
    # shorter side 384, longer side at most int(1333 / 800 * 384) = 639, both floored to multiples of 32
    assert resize_output_size(480, 640) == (384, 512) and resize_output_size(300, 1200) == (160, 608)


def test_device_processor_fails_loudly_without_a_gpu():
    import torch
    from vault_amd.preprocess import DeviceImageProcessor
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        DeviceImageProcessor()
