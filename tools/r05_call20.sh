#!/bin/bash
# config 5 with the 8-wave MXFP8 form (QKV + FFN-in, head-major, 8-bit gelu'): tests, then same-box pairs bf16 / fp8
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c20; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mx8.py -x -q > $O/t_mx8.txt 2>&1
rc=$?; tail -5 $O/t_mx8.txt
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do
  timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-other-configs --no-cpu-baseline --no-parity --no-h2d > $O/bf16_$r.json 2>$O/bf16_$r.err || exit 1
  timeout -k 10 300 python bench.py --steps 15 --warmup 4 --no-other-configs --no-cpu-baseline --no-parity --no-h2d --fp8-forward > $O/fp8_$r.json 2>$O/fp8_$r.err || exit 1
  python - <<PY
import json
a=json.load(open("$O/bf16_$r.json")); b=json.load(open("$O/fp8_$r.json"))
print("round $r bf16", a["value"], a["ms_per_step"], "fp8", b["value"], b["ms_per_step"], "ratio", round(b["value"]/a["value"],4))
PY
done
