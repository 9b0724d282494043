# the one-launch image preprocessing: parity tests, kernel timing next to the two-pass form
set -e
mkdir -p gpurun_out/c24
timeout -k 10 600 python -m pytest tests/test_gpu_preprocess.py -x -q > gpurun_out/c24/tests.log 2>&1 || { tail -30 gpurun_out/c24/tests.log; exit 1; }
tail -2 gpurun_out/c24/tests.log
timeout -k 10 300 python tools/preprocess_bench.py 256 > gpurun_out/c24/preprocess.txt 2>&1 || { tail -30 gpurun_out/c24/preprocess.txt; exit 1; }
grep -v amdgpu gpurun_out/c24/preprocess.txt
