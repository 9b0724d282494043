#!/bin/bash
# Round 6, call 2: split-K tests + micro-benchmark, whole GPU suite, small-batch lines after the patch-projection change.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c2; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_gemm.py -x -q -k "split_k" > $O/test_splitk.log 2>&1; echo "split-k tests rc=$?"; tail -5 $O/test_splitk.log
timeout -k 10 200 python tools/splitk_bench.py > $O/splitk_bench.txt 2>&1; echo "bench rc=$?"; cat $O/splitk_bench.txt | grep -v amdgpu
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/test_all.log 2>&1; echo "all tests rc=$?"; tail -5 $O/test_all.log
python bench.py --config 2 --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b64.json 2> $O/bench_b64.err
python bench.py --batch 32 --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b32.json 2> $O/bench_b32.err
python bench.py --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b256.json 2> $O/bench_b256.err
cut -c1-260 $O/bench_b256.json; echo; cut -c1-260 $O/bench_b64.json; echo; cut -c1-260 $O/bench_b32.json; echo
