#!/bin/bash
# Run on the GPU box (gpurun): bench lines (+ rocprofv3 kernel stats) of the other BASELINE configurations; writes under
# gpurun_out/extra/ (copied into profiles/ afterwards).  rocprofv3 gets the program itself after `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/extra; mkdir -p $O
python bench.py --batch 64 --no-cpu-baseline > $O/bench_b64.json 2> $O/bench_b64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b64 -o run -- python3 bench.py --batch 64 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d > $O/b64_rocprof.log 2>&1
python tools/prof_summary.py $O/b64 > $O/b64_kernel_stats.txt; rm -rf $O/b64
python bench.py --lm bert-base-uncased --freeze-lm --batch 128 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o run -- python3 bench.py --lm bert-base-uncased --freeze-lm --batch 128 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d > $O/cfg4_rocprof.log 2>&1
python tools/prof_summary.py $O/cfg4 > $O/cfg4_kernel_stats.txt; rm -rf $O/cfg4
python bench.py --fp8-forward --no-cpu-baseline > $O/bench_fp8.json 2> $O/bench_fp8.err
python bench.py --no-cpu-baseline > $O/bench_bf16_same_box.json 2> $O/bench_bf16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp8 -o run -- python3 bench.py --fp8-forward --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d > $O/fp8_rocprof.log 2>&1
python tools/prof_summary.py $O/fp8 > $O/fp8_kernel_stats.txt; rm -rf $O/fp8
VAULT_GEMM_SCHED=3 python bench.py --no-cpu-baseline > $O/bench_dp_mode_one_gpu.json 2> $O/bench_dp.err
python tools/ragged_bench.py > $O/ragged.txt 2>&1
python bench.py --batch 8 --no-cpu-baseline > $O/bench_b8.json 2> $O/bench_b8.err
python tools/preprocess_bench.py 256 > $O/preprocess.txt 2>&1
ls -la $O
