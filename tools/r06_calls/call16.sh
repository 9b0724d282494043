#!/bin/bash
# Round 6, call 16: the optimizer pass alone with a share of never-touched elements.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c16; mkdir -p $O
timeout -k 10 300 python tools/adamw_bench.py 2>&1 | grep -v amdgpu | tee $O/adamw_idle.txt
