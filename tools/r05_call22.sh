#!/bin/bash
# kernel tables of bf16 and fp8-forward steps on one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c22; mkdir -p $O
for m in bf16 fp8; do
  F=""; [ $m = fp8 ] && F="--fp8-forward"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$m -o run -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs $F > $O/$m.log 2>&1 || exit 1
  python tools/prof_by_shape.py $O/st_$m > $O/by_shape_$m.txt 2>&1
  rm -rf $O/st_$m
done
head -40 $O/by_shape_fp8.txt
