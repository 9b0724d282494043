#!/bin/bash
# Run on the GPU box: same-box comparison of several environment settings ("A=1 B=2" strings).  usage: ab_multi.sh "bench args" reps "env1" "env2" ...
cd "$GRAFT_REPO_ROOT"
ARGS=$1; REPS=$2; shift 2
for i in $(seq $REPS); do
  for E in "$@"; do
    env $E python bench.py $ARGS --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$E]', d['value'], d['ms_per_step_median'])"
  done
done
