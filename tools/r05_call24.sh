#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c24; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mx8.py -x -q > $O/t.txt 2>&1
rc=$?; tail -3 $O/t.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/mx8_bench.py 256 2>&1 | grep -v amdgpu | tee $O/mx8_bench.txt
timeout -k 10 600 python tools/ab_attr.py FFN_OUT_FP8 False,True 256 3 fp8 2>&1 | grep -v amdgpu | tee $O/ffn_out.txt
