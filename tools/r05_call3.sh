#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q > $O/gemm_tests.log 2>&1; echo "gemm tests rc=$?"; tail -3 $O/gemm_tests.log
bash tools/ab_libs.sh "pf0 tree pf2 pf5" 2 python tools/pf_bench.py 47360 dgrad,res,wgrad,lm 2>&1 | tee $O/pf_bench.txt
