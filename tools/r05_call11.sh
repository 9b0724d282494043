#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c11; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q > $O/gemm_tests.log 2>&1; echo "gemm tests (tree) rc=$?"; tail -2 $O/gemm_tests.log
bash tools/ab_libs.sh "tree ord1 ord2" 2 python tools/pf_bench.py 47360 dgrad,res,wgrad,lm 2>&1 | tee $O/order.txt
