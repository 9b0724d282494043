#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c14; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log
python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -3
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c14/bench_default.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","step_mfma_frac","vilt_block_frac","lm_block_frac"): print(k,d[k])
print({k:v for k,v in d["roofline"].items() if k in ("frac","traffic","avg_launch_ms")})
print("fp16",d["parity"]["fp16_operands"]["ratio_to_the_bf16_line"], {k:(v.get("value")) for k,v in d["other_configs"].items()})
PY
