#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace of a short bench, grouped by (kernel, grid): $1 = output tag, rest = bench args
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$1; mkdir -p $O; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-h2d "$@" > $O/bench.log 2>&1
python tools/prof_by_shape.py $O/stats > $O/by_shape.txt 2>&1
rm -rf $O/stats
head -34 $O/by_shape.txt
