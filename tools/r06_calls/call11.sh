#!/bin/bash
# Round 6, call 11: LayerNorm backward of the post-LN stack (test + bench), whole GPU suite.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c11; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q > $O/test_ops.log 2>&1; echo "ops tests rc=$?"; tail -3 $O/test_ops.log
timeout -k 10 300 python tools/ln_bench.py 2>&1 | grep -v amdgpu | grep "post-LN" | tee $O/ln_post.txt
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; echo "all tests rc=$?"; tail -4 $O/test_all.log
python bench.py --no-cpu-baseline --no-other-configs --no-parity --no-h2d 2>/dev/null | tail -1 | cut -c90-400
