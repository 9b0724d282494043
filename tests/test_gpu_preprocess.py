"""GPU parity of the image-preprocessing kernels (csrc/preprocess.hip through vault_amd.preprocess.DeviceImageProcessor)
against the HuggingFace processor's own output (golden digests) and against the CPU oracle on more shapes: bit-exact."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as PO
from tests.test_preprocess import CASES, load_case
from vault_amd.preprocess import DeviceImageProcessor

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", CASES)
def test_device_processor_equals_the_hf_processor(name):
    g, imgs = load_case(name)
    out = DeviceImageProcessor()(imgs, return_tensors="pt")
    pv, pm = out["pixel_values"].cpu().numpy(), out["pixel_mask"].cpu().numpy()
    assert pv.dtype == np.float32 and pm.dtype == np.int64 and out["pixel_values"].is_cuda
    assert hashlib.sha256(np.ascontiguousarray(pv).tobytes()).digest() == g["pixel_values_sha256"].tobytes()
    assert hashlib.sha256(np.ascontiguousarray(pm).tobytes()).digest() == g["pixel_mask_sha256"].tobytes()


def test_device_processor_equals_the_oracle_on_ragged_extreme_batches():
    rng = np.random.default_rng(7)
    sizes = [(33, 33), (64, 1200), (1200, 64), (384, 640), (385, 641), (2, 3), (700, 701), (96, 64), (1, 1)]
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    imgs[3][:] = 255; imgs[4][:] = 0                       # saturated: overshoot clipping
    imgs[6][::2] = 255; imgs[6][1::2] = 0                  # hardest ringing case
    proc = DeviceImageProcessor(mask_dtype=torch.float32)
    out = proc(imgs)
    pv_ref, pm_ref = PO.preprocess(imgs)
    assert np.array_equal(out["pixel_values"].cpu().numpy(), pv_ref)
    assert out["pixel_mask"].dtype == torch.float32 and np.array_equal(out["pixel_mask"].cpu().numpy(), pm_ref.astype(np.float32))
    # PIL images, channel-first arrays and tensors are accepted alike
    from PIL import Image
    mixed = [Image.fromarray(imgs[0]), imgs[1].transpose(2, 0, 1), torch.from_numpy(imgs[2])]
    o2 = DeviceImageProcessor()(mixed)
    r2 = PO.preprocess(imgs[:3])
    assert np.array_equal(o2["pixel_values"].cpu().numpy(), r2[0]) and np.array_equal(o2["pixel_mask"].cpu().numpy(), r2[1])
    with pytest.raises(TypeError):
        proc([imgs[0].astype(np.float32)])
    with pytest.raises(ValueError):
        proc([np.zeros((1, 4000, 3), dtype=np.uint8)])     # resizes to an empty image (HF fails on it too)


def test_preprocessed_batch_feeds_the_model():
    """pixel_values / pixel_mask from the device processor go straight into the engine's padded-image path."""
    from vault_amd.engine import VaultEngine
    from vault_amd.spec import VaultSpec, build_state, synthetic_batch
    spec = VaultSpec.tiny(3, "bert")
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in [(100, 150), (120, 90)]]
    px = DeviceImageProcessor(shortest_edge=96, size_divisor=16)(imgs)        # tiny model: 16-pixel patches
    bn = synthetic_batch(spec, 2, seed=1)
    batch = {"input_ids": torch.from_numpy(bn["input_ids"]).cuda(), "attention_mask": torch.from_numpy(bn["attention_mask"]).cuda(),
             "pixel_values": px["pixel_values"], "pixel_mask": px["pixel_mask"]}
    eng = VaultEngine(spec, "cuda:0", state=build_state(spec, 0))
    out = eng.forward(batch, train=False)
    assert out["logits"].shape == (2, 3) and torch.isfinite(out["logits"]).all()


def test_packed_host_buffer_into_preallocated_outputs_on_a_side_stream():
    """The loader-facing form: one pinned uint8 tensor + sizes, results written into preallocated tensors (the engine's input
    staging buffers in a training loop), on a non-default stream, twice with the same geometry (cached device-side plan) and
    once with another."""
    rng = np.random.default_rng(11)
    proc = DeviceImageProcessor()
    side = torch.cuda.Stream()
    for sizes in ([(480, 480)] * 3, [(480, 480)] * 3, [(100, 150), (384, 384)]):
        imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
        host = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).pin_memory()
        pv_ref, pm_ref = PO.preprocess(imgs)
        out = {"pixel_values": torch.full(pv_ref.shape, 7.0, device="cuda"),
               "pixel_mask": torch.full(pm_ref.shape, 7, dtype=torch.int64, device="cuda")}
        with torch.cuda.stream(side):
            res = proc.from_packed(host, sizes, out=out)
        side.synchronize()
        assert res["pixel_values"] is out["pixel_values"]
        assert np.array_equal(out["pixel_values"].cpu().numpy(), pv_ref) and np.array_equal(out["pixel_mask"].cpu().numpy(), pm_ref)
    with pytest.raises(ValueError):
        proc.from_packed(host[:-3], sizes)
    with pytest.raises(ValueError):
        proc.from_packed(host, sizes, out={"pixel_values": torch.zeros(2, 3, 8, 8, device="cuda"), "pixel_mask": out["pixel_mask"]})
