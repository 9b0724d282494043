#!/bin/bash
# what bounds the 8-wave main loop: W8_ABLATE 5 = half of the fragment LDS reads, 6 = no staging (LDS-DMA) in the loop, 7 = both
# (outputs are garbage: timing only).  bf16 column of tools/mx8_bench.py, QKV and FFN-in rows.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c33; mkdir -p $O
for i in 1 2; do for L in tree a5 a6 a7; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  echo "== $L"; VAULT_HIP_LIB=$P timeout -k 10 120 python tools/mx8_bench.py 256 2>&1 | grep -v amdgpu | grep "qkv\|ffn1" | sed 's/.*| bf16 GEMM/bf16 GEMM/'
done; done 2>&1 | tee $O/abl.txt
