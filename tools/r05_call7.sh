#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c7; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/bench_under_rocprof.log 2>&1
grep '^{"metric' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
python tools/prof_summary.py $O/stats > $O/kernel_stats.txt
python tools/prof_by_shape.py $O/stats > $O/kernel_by_shape.txt 2>&1
rm -rf $O/stats
head -32 $O/kernel_by_shape.txt | cut -c1-150
python -c "
import json;d=json.loads(open('$O/bench_under_rocprof.json').read());print(d['value'],d['ms_per_step'],d['vilt_block_frac'],d['lm_block_frac'],d['blocks'])"
