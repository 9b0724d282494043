#!/bin/bash
# Round 6, call 20: what the driver runs at the end of the round - the GPU suite, smoke(), the default bench line - on the final tree.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c20; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/test_all.log 2>&1; echo "all tests rc=$?"; tail -3 $O/test_all.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; grep "Elapsed" $O/bench_driver_cmd.err; cut -c1-400 $O/bench_driver_cmd.json
