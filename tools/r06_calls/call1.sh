#!/bin/bash
# Round 6, call 1: same-box baselines (default line, config 2, batch 32) and kernel traces with timestamps of B = 64 / 32 / 256
# for tools/timeline.py (where a small-batch step spends its wall time).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c1; mkdir -p $O
python bench.py --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b256.json 2> $O/bench_b256.err
python bench.py --config 2 --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b64.json 2> $O/bench_b64.err
python bench.py --batch 32 --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_b32.json 2> $O/bench_b32.err
for B in 64 32 256; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t$B -o run -- python3 bench.py --batch $B --steps 6 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/t${B}_rocprof.log 2>&1
  python tools/timeline.py $O/t$B 3 > $O/timeline_b$B.txt 2>&1
  python tools/prof_by_shape.py $O/t$B > $O/by_shape_b$B.txt 2>&1
  rm -rf $O/t$B
done
cut -c1-400 $O/bench_b256.json; echo; cut -c1-300 $O/bench_b64.json; echo; cut -c1-300 $O/bench_b32.json; echo
head -5 $O/timeline_b64.txt
