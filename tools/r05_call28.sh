#!/bin/bash
# 8-wave GEMM: placement of a K tile's staging pieces against its fragment reads (W8_ORDER 0 = tree, 1 = reads first, 2 = pieces first), in the step
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c28; mkdir -p $O
for i in 1 2 3; do for L in tree w1 w2; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  VAULT_HIP_LIB=$P python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step_median'], d['vilt_block_frac'], d['lm_block_frac'], d['roofline_ffn1']['avg_launch_ms'])"
done; done 2>&1 | tee $O/ab.txt
