#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c12; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mx8.py tests/test_gpu_gemm.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
python bench.py --config 5 --no-cpu-baseline --no-other-configs --no-h2d > $O/bench_fp8.json 2> $O/bench_fp8.err; echo rc=$?
python bench.py --no-cpu-baseline --no-other-configs --no-h2d > $O/bench_bf16.json 2> $O/bench_bf16.err; echo rc=$?
python - <<'PY'
import json
for f in ("bench_fp8","bench_bf16"):
    d=json.loads(open("gpurun_out/c12/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["vilt_block_frac"], d["lm_block_frac"], d["blocks"]["vilt"]["ms_forward"], d["blocks"]["lm"]["ms_forward"], d["parity"]["max_abs_dlogits"], d["final_loss"])
PY
