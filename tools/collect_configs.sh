#!/bin/bash
# Run on the GPU box (gpurun): bench lines (+ rocprofv3 kernel stats) of the other BASELINE configurations; writes under
# gpurun_out/extra/ (copied into profiles/ afterwards).  rocprofv3 gets the program itself after `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/extra; mkdir -p $O
python bench.py --config 2 --no-cpu-baseline --no-other-configs > $O/bench_b64.json 2> $O/bench_b64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b64 -o run -- python3 bench.py --config 2 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/b64_rocprof.log 2>&1
python tools/prof_summary.py $O/b64 > $O/b64_kernel_stats.txt; rm -rf $O/b64
python bench.py --config 4 --no-cpu-baseline --no-other-configs > $O/bench_cfg4.json 2> $O/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o run -- python3 bench.py --config 4 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/cfg4_rocprof.log 2>&1
python tools/prof_summary.py $O/cfg4 > $O/cfg4_kernel_stats.txt; rm -rf $O/cfg4
python bench.py --config 5 --no-cpu-baseline --no-other-configs > $O/bench_fp8.json 2> $O/bench_fp8.err
python bench.py --no-cpu-baseline --no-other-configs > $O/bench_bf16_same_box.json 2> $O/bench_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp8 -o run -- python3 bench.py --config 5 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/fp8_rocprof.log 2>&1
python tools/prof_summary.py $O/fp8 > $O/fp8_kernel_stats.txt; rm -rf $O/fp8
python tools/ragged_bench.py > $O/ragged.txt 2>&1
python bench.py --batch 8 --no-cpu-baseline --no-other-configs > $O/bench_b8.json 2> $O/bench_b8.err
python tools/preprocess_bench.py 256 > $O/preprocess.txt 2>&1
VAULT_FORCE_DP=1 python bench.py --no-cpu-baseline --no-other-configs --no-h2d --no-parity > $O/bench_dp1_fp32_wire.json 2> $O/bench_dp1_fp32.err
VAULT_FORCE_DP=1 python bench.py --no-cpu-baseline --no-other-configs --no-h2d --no-parity --wire bf16 > $O/bench_dp1_bf16_wire.json 2> $O/bench_dp1_bf16.err
ls -la $O
