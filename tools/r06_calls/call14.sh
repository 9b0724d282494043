#!/bin/bash
# Round 6, call 14: the vendor library at the ViLT layer's GEMM shapes (yardstick).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c14; mkdir -p $O
timeout -k 10 300 python tools/vendor_gemm_bench.py 2>&1 | grep -v amdgpu | tee $O/vendor_gemm.txt
timeout -k 10 300 python tools/pf_bench.py 47360 dgrad,res,wgrad 2>&1 | grep -v amdgpu | tee $O/pf_bench.txt
