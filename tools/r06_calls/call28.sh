# uint8 input pipeline with the step on a high-priority stream; + the GPU test files touched today
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c28; mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for hp in 0 1 0 1; do
BENCH_U8_HP=$hp timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity > $O/bench_hp$hp.json 2> $O/bench_hp$hp.err || { tail -20 $O/bench_hp$hp.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$O/bench_hp$hp.json").read().strip().splitlines()[-1])
print("hp=$hp value", d["value"], "h2d", d["with_h2d_input_copies"]["value"], "%+.2f %%" % (100*(d["with_h2d_input_copies"]["value"]/d["value"]-1)), "uint8", d["with_uint8_input_pipeline"]["value"], "%+.2f %%" % (100*(d["with_uint8_input_pipeline"]["value"]/d["value"]-1)))
PY
done
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_preprocess.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
