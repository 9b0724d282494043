"""Throughput of the device image preprocessing (vault_image_preprocess), kernels only (inputs resident, HIP events on the
launch stream): the one-launch form (8-bit intermediate in LDS) next to the two-pass form (intermediate in HBM), for the HF
contract's outputs (f32 canvas + int64 mask) and for the loader's (the 16-bit patch unfold only); then the whole call with the
host plan + H2D copy, and the HuggingFace CPU processor on a bounded sample.

  python tools/preprocess_bench.py [B=256]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vault_amd import lib as L
from vault_amd.preprocess import DeviceImageProcessor, PreprocessArgs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
fn = L.load().vault_image_preprocess
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def kernels(proc, sizes, src, form, fused):
    desc_b, plan, src_bytes, tmp_bytes, H, W, mh, mw, ks, br = proc.plan(sizes)
    plan_d = torch.from_numpy(plan).cuda(); desc_d = torch.frombuffer(bytearray(desc_b), dtype=torch.uint8).cuda()
    tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device="cuda")
    a = PreprocessArgs()
    a.src, a.tmp, a.plan, a.desc, a.lut = src.data_ptr(), tmp.data_ptr(), plan_d.data_ptr(), desc_d.data_ptr(), proc._lut.data_ptr()
    outs = []
    if form == "hf":
        pv = torch.empty(B, 3, H, W, device="cuda"); pm = torch.empty(B, H, W, dtype=torch.int64, device="cuda")
        a.pixel_values, a.pixel_mask = pv.data_ptr(), pm.data_ptr()
        outs, out_bytes = [pv, pm], pv.numel() * 4 + pm.numel() * 8
    else:
        po = torch.empty(B * (H // 32) * (W // 32), 3 * 32 * 32, dtype=torch.bfloat16, device="cuda")
        a.patch_unfold_bf16, a.ps = po.data_ptr(), 32
        outs, out_bytes = [po], po.numel() * 2
    a.B, a.H, a.W, a.max_h_in, a.max_w_out = B, H, W, mh, mw
    a.max_w_in, a.src_bytes = max(w for _, w in sizes), src_bytes
    if fused:
        a.ksize_max, a.band_rows_max = ks, br
    assert bool(L.load().vault_image_preprocess_is_fused(C.byref(a))) == fused
    for _ in range(3):
        L.check(fn(C.byref(a), st), "vault_image_preprocess")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn(C.byref(a), st)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    byts = src_bytes + out_bytes + (0 if fused else 2 * tmp_bytes)
    return t, byts, outs


for hw in ((375, 500), (480, 480)):
    imgs = [rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8) for _ in range(B)]
    sizes = [hw] * B
    src = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).cuda()
    proc = DeviceImageProcessor()
    for form, what in (("hf", "f32 canvas + int64 mask"), ("loader", "16-bit patch unfold only")):
        res = {}
        for fused in (True, False):
            t, byts, outs = kernels(proc, sizes, src, form, fused)
            res[fused] = [o.clone() for o in outs]
            print(f"B={B} {hw[0]}x{hw[1]} -> {what}: {'one launch' if fused else 'two passes'} {t*1e3:.3f} ms = {B/t:,.0f} images/s, "
                  f"{byts/t/1e9:.0f} GB/s of {byts/1e6:.0f} MB algorithmic")
        print("   identical:", all(torch.equal(x, y) for x, y in zip(res[True], res[False])))
imgs = [rng.integers(0, 256, size=(375, 500, 3), dtype=np.uint8) for _ in range(B)]
proc = DeviceImageProcessor()
out = proc(imgs)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    out = proc(imgs)
torch.cuda.synchronize()
t_all = (time.time() - t0) / 5
print(f"B={B} 375x500 with host plan + H2D {t_all*1e3:.1f} ms = {B/t_all:,.0f} images/s")
try:
    from PIL import Image
    from transformers.models.vilt.image_processing_pil_vilt import ViltImageProcessorPil
    hf = ViltImageProcessorPil()
    n = min(B, 32)
    pil = [Image.fromarray(im) for im in imgs[:n]]
    t0 = time.time(); ref = hf(pil, return_tensors="np"); t_hf = time.time() - t0
    same = np.array_equal(ref["pixel_values"], out["pixel_values"][:n].cpu().numpy())
    print(f"HF ViltImageProcessorPil, 1 thread, {n} images: {t_hf*1e3:.0f} ms = {n/t_hf:,.0f} images/s; identical output: {same}")
except Exception as ex:  # noqa: BLE001
    print("HF processor not available:", ex)
