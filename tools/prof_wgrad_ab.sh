#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wgrad_ab; mkdir -p $O
for V in 0 1; do
  VAULT_WGRAD_GROUPED=$V rocprofv3 --kernel-trace --output-format csv -d $O/t$V -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/log$V.txt 2>&1
  echo "VAULT_WGRAD_GROUPED=$V" >> $O/seq.txt
  python tools/prof_seq.py $O/t$V "gemm256_kernel<1, 1, 5, 4>" $([ $V = 0 ] && echo 17 || echo 13) >> $O/seq.txt
  rm -rf $O/t$V
done
cat $O/seq.txt
