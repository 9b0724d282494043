#!/bin/bash
# Round 6, call 6: tail-split tests, whole GPU suite, small-batch lines.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_gemm.py -x -q -k "tail_rows or split_k" > $O/test_tail.log 2>&1; echo "tail tests rc=$?"; tail -4 $O/test_tail.log
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; echo "all tests rc=$?"; tail -6 $O/test_all.log
python bench.py --batch 32 --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b32.json 2> $O/bench_b32.err
python bench.py --config 2 --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b64.json 2> $O/bench_b64.err
python bench.py --config 4 --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python bench.py --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b256.json 2> $O/bench_b256.err
for f in bench_b32 bench_b64 bench_cfg4 bench_b256; do cut -c90-260 $O/$f.json; echo; done
