#!/bin/bash
# Run on the GPU box: same-box comparison of several values of one environment variable.
# usage: ab_values.sh VAR "v1 v2 ..." "bench args" [reps]
cd "$GRAFT_REPO_ROOT"
VAR=$1; VALS=$2; ARGS=$3; REPS=${4:-2}
for i in $(seq $REPS); do
  for V in $VALS; do
    env $VAR=$V python bench.py $ARGS --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$V', '$ARGS', d['value'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
  done
done
