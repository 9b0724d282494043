#!/bin/bash
# Round 6, call 19: four-stage 128 x 128 tiles for K = 768 launches of at most 256 tiles (the LM stack at B <= 32): micro-benchmark + step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c19; mkdir -p $O
for i in 1 2; do for L in tree ns4k768; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  echo "== $L"; VAULT_HIP_LIB=$P timeout -k 10 120 python tools/lm_shapes_bench.py 1280 640 2>&1 | grep -v amdgpu | cut -c1-120
  VAULT_HIP_LIB=$P python bench.py --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-h2d 2>/dev/null | tail -1 | cut -c90-200
done; done 2>&1 | tee $O/ns4.txt
