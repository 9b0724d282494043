#!/bin/bash
# Copy the summaries of tools/collect_profiles.sh / collect_configs.sh (gpurun_out/final, gpurun_out/extra) into profiles/ under
# the names of round $1 (e.g. r02)
R=$1; F=gpurun_out/final; E=gpurun_out/extra; P=profiles
cp $F/bench_default.json $P/${R}_bench_b256_default.json; cp $F/kernel_stats.txt $P/${R}_bench_b256_kernel_stats.txt
cp $F/kernel_by_shape.txt $P/${R}_bench_b256_kernel_by_shape.txt; cp $F/bench_under_rocprof.json $P/${R}_bench_b256_under_rocprof.json
cp $F/pmc_FETCH_SIZE.txt $P/${R}_pmc_FETCH_SIZE_bench_b256.txt; cp $F/pmc_WRITE_SIZE.txt $P/${R}_pmc_WRITE_SIZE_bench_b256.txt
cp $F/pmc_gemm_wgrad.json $P/${R}_pmc_gemm_wgrad.json; cp $F/pmc_gemm_ffn1.json $P/${R}_pmc_gemm_ffn1.json
cp $F/pmc_mfma_util.txt $P/${R}_pmc_mfma_util_bench_b256.txt
[ -f $F/kernel_stats_fp16.txt ] && cp $F/kernel_stats_fp16.txt $P/${R}_bench_b256_fp16_kernel_stats.txt && cp $F/bench_fp16_under_rocprof.json $P/${R}_bench_b256_fp16_under_rocprof.json
cp $E/bench_b64.json $P/${R}_bench_b64.json; cp $E/b64_kernel_stats.txt $P/${R}_bench_b64_kernel_stats.txt; cp $E/bench_b8.json $P/${R}_bench_b8.json
cp $E/bench_cfg4.json $P/${R}_bench_cfg4_frozen_bert_b128.json; cp $E/cfg4_kernel_stats.txt $P/${R}_bench_cfg4_kernel_stats.txt
cp $E/bench_fp8.json $P/${R}_bench_b256_fp8_forward.json; cp $E/bench_bf16_same_box.json $P/${R}_bench_b256_bf16_same_box_as_fp8.json
cp $E/fp8_kernel_stats.txt $P/${R}_bench_b256_fp8_forward_kernel_stats.txt
cp $E/ragged.txt $P/${R}_ragged_384x640.txt; cp $E/preprocess.txt $P/${R}_preprocess_b256.txt
for f in bench_b256_default bench_b256_under_rocprof bench_b64 bench_b8 bench_cfg4_frozen_bert_b128 bench_b256_fp8_forward bench_b256_bf16_same_box_as_fp8; do
python - <<PY
import json
d=json.loads(open('$P/${R}_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}; q=d.get('roofline_ffn1') or {}
print('$f', d['value'], d['ms_per_step'], d.get('step_mfma_frac'), 'wgrad', r.get('frac'), r.get('avg_launch_ms'), r.get('traffic'), 'ffn1', q.get('frac'), q.get('traffic'))
PY
done
grep -v amdgpu $P/${R}_ragged_384x640.txt $P/${R}_preprocess_b256.txt
