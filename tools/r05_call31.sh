#!/bin/bash
# attention forward: query fragments requested before the K / V staging (tree) against the previous library (before)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c31; mkdir -p $O
for i in 1 2 3; do for L in before tree; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  echo "== $L"; VAULT_HIP_LIB=$P timeout -k 10 120 python tools/attn_bench.py 256 2>&1 | grep -v amdgpu
done; done 2>&1 | tee $O/attn.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -2
