#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats of the default bench + two PMC passes; writes summaries under
# gpurun_out/final/ (copied into profiles/ afterwards).  rocprofv3 gets the program itself after `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/bench_under_rocprof.log 2>&1
grep '^{"metric' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
python tools/prof_summary.py $O/stats > $O/kernel_stats.txt
python tools/prof_by_shape.py $O/stats > $O/kernel_by_shape.txt 2>&1
rm -rf $O/stats
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/pmc_$C.log 2>&1
  python tools/pmc_summary.py $O/pmc_$C $C > $O/pmc_$C.txt
done
python tools/make_pmc_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "gemm256_kernel<1, 1, 5, 4," $O/pmc_gemm_wgrad.json 256 bertweet "gemm256_kernel<1,1,5,4> (weight-gradient ring GEMM, EPI_F32_ATOMIC): all launches of one step - grouped launches (tiles of all four Linear kinds packed into rounds of 256) + the patch projection"
python tools/make_pmc_json.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE "gemm8w_kernel<7, 4|gemm8w_kernel<1, 4" $O/pmc_gemm_ffn1.json 256 bertweet "gemm8w_kernel<7,4,true> / <1,4,true> (8-wave register-direct GEMM, three A slots, GELU epilogue with 8-bit / bf16 gelu', 256-wide tiles): FFN-in forward; per step 12 ViLT launches (M=47360, 8-bit gelu') + 12 LM launches (M=10240, bf16 gelu')"
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/pmc_mfma.log 2>&1
python tools/pmc_mfma.py $O/pmc_mfma > $O/pmc_mfma_util.txt
rm -rf $O/pmc_mfma
# the fp16 operand build: kernel table of the same steps (same box)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats16 -o run -- python3 bench.py --half fp16 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/bench_fp16_under_rocprof.log 2>&1
grep '^{"metric' $O/bench_fp16_under_rocprof.log | tail -1 > $O/bench_fp16_under_rocprof.json
python tools/prof_summary.py $O/stats16 > $O/kernel_stats_fp16.txt
rm -rf $O/stats16
python bench.py > $O/bench_default.json 2> $O/bench_default.err
ls -la $O
