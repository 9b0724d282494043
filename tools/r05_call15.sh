#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c15; mkdir -p $O
for i in 1 2 3; do for L in tree prio ord1 both; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  VAULT_HIP_LIB=$P python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step_median'], d['vilt_block_frac'], d['lm_block_frac'], d['roofline']['avg_launch_ms'], d['roofline_ffn1']['avg_launch_ms'])"
done; done | tee $O/ab.txt
