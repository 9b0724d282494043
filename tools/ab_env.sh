#!/bin/bash
# Run on the GPU box: same-box A/B of an environment switch.  usage: ab_env.sh VAR "bench args" [reps]
# Alternates VAR=0 / VAR=1 runs of bench.py (timed steps only) and prints samples/s + median step ms for each.
cd "$GRAFT_REPO_ROOT"
VAR=$1; ARGS=$2; REPS=${3:-2}
for i in $(seq $REPS); do
  for V in 0 1; do
    env $VAR=$V python bench.py $ARGS --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$V', '$ARGS', d['value'], d['ms_per_step_median'])"
  done
done
