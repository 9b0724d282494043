#!/bin/bash
# Round 6, call 21: the driver's bench command on the final tree, timed by the shell.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c21; mkdir -p $O
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
echo "bench.py took $SECONDS s, rc=$?"; cut -c1-300 $O/bench_driver_cmd.json
