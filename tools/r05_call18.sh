#!/bin/bash
# MXFP8 form of the 8-wave GEMM: op tests, then the layer shapes against the simple kernel and bf16
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c18; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mx8.py -x -q -k "not engine and not train_step and not model_class" > $O/t_mx8.txt 2>&1
rc=$?; tail -15 $O/t_mx8.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/mx8_bench.py 256 2>&1 | grep -v amdgpu | tee $O/mx8_bench.txt
