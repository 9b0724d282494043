"""Where the one-launch preprocessing kernel spends its time: the library built with parts of the kernel compiled out
(csrc/preprocess.hip FB_ABL: 1 no horizontal arithmetic, 2 no vertical arithmetic, 4 no output writes, 8 no source loads),
480 x 480 -> the 16-bit patch unfold, B = 256.   python tools/micro/preprocess_ablation.py  (builds nothing: expects
build_ab/libvault_hip_pabl<N>.so from tools/build_variant.py pabl<N> preprocess.hip:-DFB_ABL=<N>)"""
import ctypes as C
import glob
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from vault_amd import lib as L
from vault_amd.preprocess import DeviceImageProcessor, PreprocessArgs

B, hw = 256, (480, 480)
rng = np.random.default_rng(0)
src = torch.from_numpy(rng.integers(0, 256, size=(B * hw[0] * hw[1] * 3,), dtype=np.uint8)).cuda()
proc = DeviceImageProcessor()
sizes = [hw] * B
desc_b, plan, src_bytes, tmp_bytes, H, W, mh, mw, ks, br = proc.plan(sizes)
plan_d = torch.from_numpy(plan).cuda(); desc_d = torch.frombuffer(bytearray(desc_b), dtype=torch.uint8).cuda()
po = torch.empty(B * (H // 32) * (W // 32), 3 * 32 * 32, dtype=torch.bfloat16, device="cuda")
a = PreprocessArgs()
a.src, a.plan, a.desc, a.lut = src.data_ptr(), plan_d.data_ptr(), desc_d.data_ptr(), proc._lut.data_ptr()
a.patch_unfold_bf16, a.ps = po.data_ptr(), 32
a.B, a.H, a.W, a.max_h_in, a.max_w_out, a.max_w_in, a.src_bytes = B, H, W, mh, mw, hw[1], src_bytes
a.ksize_max, a.band_rows_max = ks, br
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = [("in-tree", None)] + sorted((os.path.basename(p), p) for p in glob.glob("build_ab/libvault_hip_pabl*.so"))
for name, path in libs:
    fn = (L.load() if path is None else C.CDLL(os.path.abspath(path))).vault_image_preprocess
    for _ in range(3):
        assert fn(C.byref(a), st) == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn(C.byref(a), st)
    e.record(); torch.cuda.synchronize()
    print(f"{name:40s} {s.elapsed_time(e) / 20 * 1e3:8.1f} us")
