#!/bin/bash
# Round 6, call 4: attention backward with 2 / 3 key tiles per wave against 1 (round 5), its tests; the tests that failed in call 3.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c4; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q > $O/test_ops.log 2>&1; echo "ops tests rc=$?"; tail -4 $O/test_ops.log
for i in 1 2; do for L in tree attn_kt1 attn_kt3; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  echo "== $L"; VAULT_HIP_LIB=$P timeout -k 10 120 python tools/attn_bench.py 256 185 2>&1 | grep -v amdgpu
done; done > $O/attn_kt.txt 2>&1; cat $O/attn_kt.txt
timeout -k 10 600 python -m pytest tests -q -m gpu -k "split_k or batched_weight_gradients or sub_batches or patch_unfold or rccl_single_rank or optimizer_leaves" > $O/test_sel.log 2>&1; echo "selected tests rc=$?"; tail -6 $O/test_sel.log
for i in 1 2; do for L in tree attn_kt1 attn_kt3; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  echo "== $L"; VAULT_HIP_LIB=$P timeout -k 10 200 python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-h2d 2>/dev/null | cut -c1-230
done; done > $O/bench_kt.txt 2>&1; cat $O/bench_kt.txt
