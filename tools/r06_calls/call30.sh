#!/bin/bash
# Round 6, call 30: per-step timeline of one bench.py run with the one-launch preprocessing + adopted pixel_patches (as calls 10 / 23).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c30; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o run -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-other-configs > $O/bench_traced.log 2>&1
python tools/timeline.py $O/tr 3 adamw_kernel --per-step > $O/per_step.txt 2>&1; cat $O/per_step.txt | cut -c1-250
tail -1 $O/bench_traced.log | cut -c1-200
rm -rf $O/tr
