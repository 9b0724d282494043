#!/bin/bash
# Round 6, call 18: the optimizer pass in the step, round 5's tree against this one, ONE box (kernel traces; B = 32 and 256).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c18; mkdir -p $O
for B in 32 256; do
  for i in 1 2; do
  (cd _r05 && rocprofv3 --kernel-trace --output-format csv -d $O/a -o run -- python3 bench.py --batch $B --steps 12 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/a.log 2>&1)
  echo "r05 B=$B: $(python tools/prof_seq.py $O/a adamw_kernel 10 | tail -1)"; rm -rf $O/a
  rocprofv3 --kernel-trace --output-format csv -d $O/b -o run -- python3 bench.py --batch $B --steps 12 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/b.log 2>&1
  echo "r06 B=$B: $(python tools/prof_seq.py $O/b adamw_kernel 10 | tail -1)"; rm -rf $O/b
  done
done 2>&1 | tee $O/adamw_in_step.txt
