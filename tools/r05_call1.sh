#!/bin/bash
# round 5, first GPU call: the GPU suite at the new default + a short bench line + AdamW on both libraries
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c1; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -5 $O/pytest.log
for i in 1 2; do for f in bf16 fp16; do python tools/adamw_bench.py 222400000 $f 1; done; done 2>&1 | grep -v amdgpu.ids | tee $O/adamw.txt
timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-h2d --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 1500 $O/bench.json
