#!/bin/bash
# R256_ORDER: per-instantiation default (tree: weight gradients 0, the rest 1) against 1 everywhere (o1) - time of the step and
# HBM fetch of the weight-gradient launches, same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c27; mkdir -p $O
for i in 1 2 3; do for L in tree o1; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  VAULT_HIP_LIB=$P python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step_median'], d['vilt_block_frac'], d['lm_block_frac'], d['roofline']['avg_launch_ms'])"
done; done 2>&1 | tee $O/time_ab.txt
for L in tree o1; do
  if [ $L = tree ]; then export VAULT_HIP_LIB=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else export VAULT_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  rm -rf $O/pmc
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/pmc_$L.log 2>&1 || exit 1
  python tools/pmc_summary.py $O/pmc FETCH_SIZE > $O/fetch_$L.txt
  echo "== $L"; sed -n 1,4p $O/fetch_$L.txt
done 2>&1 | tee $O/fetch_ab.txt
rm -rf $O/pmc
