#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c13; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp8 -o run -- python3 bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/fp8_rocprof.log 2>&1
python tools/prof_by_shape.py $O/fp8 > $O/fp8_by_shape.txt 2>&1; rm -rf $O/fp8
head -30 $O/fp8_by_shape.txt | cut -c1-150
