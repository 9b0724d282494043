#!/bin/bash
# Round 6, call 22: the host->device loop of bench.py with the pixels copied straight into the staging buffer; the tests that read bench lines.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c22; mkdir -p $O
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity > $O/bench_$i.json 2> $O/bench_$i.err; tail -2 $O/bench_$i.err | grep -v amdgpu
python - <<PY
import json
d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print("value", d["value"], "f32 pixels", d["with_f32_pixel_values"]["value"], "h2d", d["with_h2d_input_copies"]["value"], "%+.2f %%" % (100*(d["with_h2d_input_copies"]["value"]/d["value"]-1)), "uint8", d["with_uint8_input_pipeline"]["value"], "%+.2f %%" % (100*(d["with_uint8_input_pipeline"]["value"]/d["value"]-1)))
PY
done
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -x -q -k "bench" > $O/test_bench.log 2>&1; echo "bench tests rc=$?"; tail -3 $O/test_bench.log
