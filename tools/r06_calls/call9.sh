#!/bin/bash
# Round 6, call 9: kernel choices at the LM stack's shapes; tests of the changed resolver.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c9; mkdir -p $O
timeout -k 10 300 python tools/lm_shapes_bench.py 2>&1 | grep -v amdgpu | tee $O/lm_shapes.txt
timeout -k 10 300 python -m pytest tests/test_gpu_gemm.py -x -q > $O/test_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -3 $O/test_gemm.log
