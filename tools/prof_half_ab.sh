#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats of the timed steps on the bf16 and on the fp16 operand build, same box.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/half_ab; mkdir -p $O
for H in bf16 fp16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$H -o run -- python3 bench.py --half $H --steps 6 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/bench_$H.log 2>&1
  grep '^{"metric' $O/bench_$H.log | tail -1 > $O/bench_$H.json
  python tools/prof_summary.py $O/stats_$H > $O/kernel_stats_$H.txt
  rm -rf $O/stats_$H
done
for H in bf16 fp16 bf16 fp16; do
  python bench.py --half $H --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$H', d['value'], d['ms_per_step_median'])" >> $O/plain.txt
done
cat $O/plain.txt
