"""Throughput of the device image preprocessing (vault_image_preprocess) on a batch of 500x375 uint8 images, kernels only
(inputs resident) and including the host plan + H2D copy, next to the HuggingFace CPU processor on a bounded sample."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from vault_amd.preprocess import DeviceImageProcessor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=(375, 500, 3), dtype=np.uint8) for _ in range(B)]
proc = DeviceImageProcessor()
out = proc(imgs)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    out = proc(imgs)
torch.cuda.synchronize()
t_all = (time.time() - t0) / 5
# kernels only: CUDA events around repeated calls are dominated by the host plan; use the profiler-free estimate
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import ctypes as C
from vault_amd import lib as L
from vault_amd.preprocess import PreprocessArgs
desc_b, plan, src_bytes, tmp_bytes, H, W, mh, mw = proc.plan([im.shape[:2] for im in imgs])
src = torch.from_numpy(np.concatenate([im.reshape(-1) for im in imgs])).cuda()
plan_d = torch.from_numpy(plan).cuda(); desc_d = torch.frombuffer(bytearray(desc_b), dtype=torch.uint8).cuda()
tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device="cuda"); pv = torch.empty(B, 3, H, W, device="cuda")
pm = torch.empty(B, H, W, dtype=torch.int64, device="cuda")
a = PreprocessArgs()
a.src, a.tmp, a.plan, a.desc, a.lut, a.pixel_values, a.pixel_mask = (src.data_ptr(), tmp.data_ptr(), plan_d.data_ptr(), desc_d.data_ptr(),
                                                                       proc._lut.data_ptr(), pv.data_ptr(), pm.data_ptr())
a.B, a.H, a.W, a.max_h_in, a.max_w_out = B, H, W, mh, mw
a.max_w_in, a.src_bytes = max(im.shape[1] for im in imgs), src_bytes
fn = L.load().vault_image_preprocess
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    fn(C.byref(a), st)
torch.cuda.synchronize()
s.record()
for _ in range(10):
    fn(C.byref(a), st)
e.record(); torch.cuda.synchronize()
t_k = s.elapsed_time(e) / 10 * 1e-3
byts = src_bytes + 2 * tmp_bytes + pv.numel() * 4 + pm.numel() * 8
print(f"B={B}: kernels {t_k*1e3:.3f} ms = {B/t_k:,.0f} images/s, {byts/t_k/1e9:.0f} GB/s of {byts/1e6:.0f} MB algorithmic; "
      f"with host plan + H2D {t_all*1e3:.1f} ms = {B/t_all:,.0f} images/s")
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
