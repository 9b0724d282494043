# one-launch preprocessing + adopted pixel_patches: tests, kernel timing, the bench's input-pipeline lines (two runs)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c25; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_preprocess.py tests/test_gpu_train.py -x -q -k "preprocess or adopted or patch or tape or one_launch or device_processor or packed" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/preprocess_bench.py 256 > $O/preprocess.txt 2>&1 || { tail -30 $O/preprocess.txt; exit 1; }
grep "480x480" $O/preprocess.txt
for i in 1 2; do
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity > $O/bench_$i.json 2> $O/bench_$i.err || { tail -20 $O/bench_$i.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print("value", d["value"], "f32 pixels", d["with_f32_pixel_values"]["value"], "h2d", d["with_h2d_input_copies"]["value"], "%+.2f %%" % (100*(d["with_h2d_input_copies"]["value"]/d["value"]-1)), "uint8", d["with_uint8_input_pipeline"]["value"], "%+.2f %%" % (100*(d["with_uint8_input_pipeline"]["value"]/d["value"]-1)))
PY
done
