# one-launch preprocessing: tests + kernel timing
set -e
O=gpurun_out/c26; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_preprocess.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/preprocess_bench.py 256 > $O/preprocess.txt 2>&1 || { tail -30 $O/preprocess.txt; exit 1; }
grep -v amdgpu $O/preprocess.txt
