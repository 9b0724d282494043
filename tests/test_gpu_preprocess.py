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
    eng = VaultEngine(spec, "cuda:0", state=build_state(spec, 0), half="bf16")
    out = eng.forward(batch, train=False)
    assert out["logits"].shape == (2, 3) and torch.isfinite(out["logits"]).all()
    # the loader-facing form also returns the valid sizes it padded from: with that host-side hint the engine builds the patch
    # grid of the mask itself (no device -> host read of pixel_mask per step) - same result; and the patch bookkeeping of a
    # repeated geometry comes from the engine's cache
    lg = out["logits"].clone()
    host = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).pin_memory()
    px2 = DeviceImageProcessor(shortest_edge=96, size_divisor=16).from_packed(host, [im.shape[:2] for im in imgs])
    assert px2["valid_hw"] == [(96, 144), (128, 96)]
    n_cached = len(eng._sel_cache)
    out2 = eng.forward(dict(batch, valid_hw=px2["valid_hw"]), train=False)
    assert torch.equal(out2["logits"], lg) and len(eng._sel_cache) == n_cached == 1
    with pytest.raises(ValueError):
        eng.forward(dict(batch, valid_hw=[(96, 144)]), train=False)


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


def test_patch_unfold_straight_from_the_resize_kernel():
    """``from_packed(patch_out=...)``: the patch-embedding GEMM's bf16 operand written by the vertical resize pass itself
    (no f32 ``pixel_values`` tensor, no ``vault_im2col`` pass) equals bf16(unfold(pixel_values)) of the HF-exact pixels bit
    for bit, for square and for padded non-square canvases; fed to the engine as ``pixel_patches`` it gives the logits of the
    ``pixel_values`` path exactly, in ``TrainStep`` too."""
    import torch.nn.functional as F
    from vault_amd.engine import VaultEngine
    from vault_amd.spec import VaultSpec, build_state, synthetic_batch
    from vault_amd.train import TrainStep
    rng = np.random.default_rng(21)
    for sizes, kw, ps in (([(480, 480)] * 3, {}, 32), ([(100, 150), (384, 384), (200, 120)], {}, 32),
                          ([(260, 260)] * 4, dict(shortest_edge=192, size_divisor=16), 16)):   # the tiny model's 192 x 192 canvas
        proc = DeviceImageProcessor(**kw)
        imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
        host = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).pin_memory()
        ref = proc.from_packed(host, sizes)
        pv = ref["pixel_values"]
        B, _, H, W = pv.shape
        rows, Kp = B * (H // ps) * (W // ps), 3 * ps * ps
        want = F.unfold(pv, kernel_size=ps, stride=ps).transpose(1, 2).reshape(rows, Kp).bfloat16()
        po = torch.full((rows, Kp), 7.0, dtype=torch.bfloat16, device="cuda")
        res = proc.from_packed(host, sizes, patch_out=po, patch_size=ps)
        torch.cuda.synchronize()
        assert res["pixel_patches"] is po and "pixel_values" not in res and res["canvas"] == (H, W)
        assert torch.equal(po, want) and torch.equal(res["pixel_mask"], ref["pixel_mask"])
        # the same through the fp16 build of the library: the unfold in IEEE half (the fp16 engine's patch operand)
        po16 = torch.full((rows, Kp), 7.0, dtype=torch.float16, device="cuda")
        proc.from_packed(host, sizes, patch_out=po16, patch_size=ps)
        torch.cuda.synchronize()
        assert torch.equal(po16, F.unfold(pv, kernel_size=ps, stride=ps).transpose(1, 2).reshape(rows, Kp).half())
    # the tiny model (96 x 96 canvases, 16-pixel patches): same logits and the same training steps from either input form
    spec = VaultSpec.tiny(3, "roberta")
    spec.lm.hidden_dropout_prob = 0.0; spec.lm.attention_probs_dropout_prob = 0.0
    bn = synthetic_batch(spec, 4, seed=2, n_classes=3)
    ids, am = torch.from_numpy(bn["input_ids"]).cuda(), torch.from_numpy(bn["attention_mask"]).cuda()
    labels = torch.from_numpy(bn["labels"]).cuda()
    state = build_state(spec, 0)
    outs, params = [], []
    for form in ("pixel_values", "pixel_patches"):
        eng = VaultEngine(spec, "cuda:0", state=state, classifier_dropout=0.0, half="bf16")
        img = {"pixel_values": pv} if form == "pixel_values" else {"pixel_patches": po}
        batch = dict(input_ids=ids, attention_mask=am, **img)
        outs.append(eng.forward(batch, train=False)["logits"].clone())
        step = TrainStep(eng, learning_rate=1e-4, warmup_ratio=0.0, total_steps=10, assume_full_pixel_mask=True)
        for _ in range(3):
            step(batch, labels)
        torch.cuda.synchronize()
        params.append(eng.params.p.clone())
    assert torch.equal(outs[0], outs[1])
    d = (params[0] - params[1]).abs()
    # (float-atomic summation order only; round 6: the patch projection's weight gradient is 7 splits of ring tiles now)
    assert float(d.mean()) < 3e-6 and float((d > 2e-5).float().mean()) < 0.03
    with pytest.raises(ValueError):
        eng.forward(dict(input_ids=ids, attention_mask=am, pixel_patches=po[:-1]), train=False)
    # what the processor knows about padding travels with the patches: a padded image or another canvas is refused on the host
    size = spec.vilt.image_size
    ok = dict(input_ids=ids, attention_mask=am, pixel_patches=po, canvas=(size, size), valid_hw=[(size, size)] * 4)
    plain = eng.forward(dict(input_ids=ids, attention_mask=am, pixel_patches=po), train=False)["logits"].clone()
    assert torch.equal(eng.forward(ok, train=False)["logits"], plain)
    with pytest.raises(ValueError):
        eng.forward(dict(ok, valid_hw=[(size, size)] * 3 + [(size, size - 16)]), train=False)
    with pytest.raises(ValueError):
        eng.forward(dict(ok, canvas=(size, 2 * size)), train=False)


def test_one_launch_form_equals_the_two_pass_form_and_the_oracle():
    """The fused kernel (both passes of a 32-row band in one workgroup, 8-bit intermediate in LDS) against the two-pass form
    (intermediate in HBM) and the CPU oracle: same bytes - down- and upscaling, tap counts beyond one 8-tap chunk, ragged
    batches with padded bands, every output form (f32 canvas, int64 / f32 mask, 16-bit unfold), odd source offsets (the
    16-byte staging loads start at any byte), and no mask at all."""
    import torch.nn.functional as F
    rng = np.random.default_rng(31)
    batches = ([(480, 480)] * 3, [(100, 150), (384, 384), (200, 120)], [(700, 701), (500, 333)], [(33, 33), (96, 64), (2, 3), (1, 1)],
               [(385, 641), (641, 385), (37, 41)], [(901, 517)])
    for sizes in batches:
        imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
        if len(imgs) > 1:
            imgs[1][::2] = 255; imgs[1][1::2] = 0              # ringing: the clip after each pass matters
        host = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).pin_memory()
        one, two = DeviceImageProcessor(), DeviceImageProcessor(fused=False)
        a, b = one.from_packed(host, sizes), two.from_packed(host, sizes)
        assert one._plan_dev[2] is None and two._plan_dev[2] is not None          # (no HBM intermediate / the HBM intermediate)
        pv_ref, pm_ref = PO.preprocess(imgs)
        assert np.array_equal(a["pixel_values"].cpu().numpy(), pv_ref) and np.array_equal(a["pixel_mask"].cpu().numpy(), pm_ref)
        assert torch.equal(a["pixel_values"], b["pixel_values"]) and torch.equal(a["pixel_mask"], b["pixel_mask"])
        mf = DeviceImageProcessor(mask_dtype=torch.float32).from_packed(host, sizes)
        assert torch.equal(mf["pixel_mask"], a["pixel_mask"].float()) and torch.equal(mf["pixel_values"], a["pixel_values"])
        B, _, H, W = a["pixel_values"].shape
        rows, Kp = B * (H // 32) * (W // 32), 3 * 32 * 32
        po = torch.full((rows, Kp), 7.0, dtype=torch.bfloat16, device="cuda")
        res = one.from_packed(host, sizes, patch_out=po, want_mask=False)
        assert res["pixel_mask"] is None and res["valid_hw"] == a["valid_hw"]
        assert torch.equal(po, F.unfold(a["pixel_values"], kernel_size=32, stride=32).transpose(1, 2).reshape(rows, Kp).bfloat16())
    # a batch whose band does not fit the LDS runs the two-pass form by itself
    big = DeviceImageProcessor()
    imgs = [rng.integers(0, 256, size=(64, 1200, 3), dtype=np.uint8)]
    o = big.from_packed(torch.from_numpy(imgs[0].reshape(-1)).pin_memory(), [(64, 1200)])
    assert big._plan_dev[2] is not None and np.array_equal(o["pixel_values"].cpu().numpy(), PO.preprocess(imgs)[0])


def test_one_launch_form_on_random_geometries():
    """Twelve random ragged batches (sides 17 .. 1100, any aspect): the one-launch kernel's bytes equal the two-pass kernels'
    wherever the library takes it (and the library takes it for most), for both output forms."""
    rng = np.random.default_rng(123)
    fused_runs = 0
    for trial in range(12):
        n = int(rng.integers(1, 5))
        sizes = [(int(rng.integers(17, 1100)), int(rng.integers(17, 1100))) for _ in range(n)]
        sizes = [(h, w) for h, w in sizes if max(h, w) / min(h, w) < 12] or [(64, 64)]      # (very thin images resize to nothing)
        imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
        host = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).pin_memory()
        one, two = DeviceImageProcessor(), DeviceImageProcessor(fused=False)
        try:
            a = one.from_packed(host, sizes)
        except ValueError:                       # an image that resizes to an empty one: the HF processor fails on it too
            continue
        b = two.from_packed(host, sizes)
        fused_runs += one._plan_dev[2] is None
        assert torch.equal(a["pixel_values"], b["pixel_values"]) and torch.equal(a["pixel_mask"], b["pixel_mask"]), sizes
        B, _, H, W = a["pixel_values"].shape
        po1 = torch.full((B * (H // 32) * (W // 32), 3072), 7.0, dtype=torch.bfloat16, device="cuda")
        po2 = torch.full_like(po1, 3.0)
        one.from_packed(host, sizes, patch_out=po1, want_mask=False)
        two.from_packed(host, sizes, patch_out=po2)
        assert torch.equal(po1, po2), sizes
    assert fused_runs >= 6
