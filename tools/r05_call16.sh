#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c16; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
