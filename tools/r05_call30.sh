#!/bin/bash
# kernel table of the data-parallel step's code path on one rank (VAULT_FORCE_DP=1), against the plain step of the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c30; mkdir -p $O
export VAULT_FORCE_DP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o run -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/dp.log 2>&1 || exit 1
python tools/prof_by_shape.py $O/st > $O/by_shape_dp.txt 2>&1
rm -rf $O/st
unset VAULT_FORCE_DP
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['ms_per_step'])"
VAULT_FORCE_DP=1 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-parity --no-h2d --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dp1', d['value'], d['ms_per_step'])"
head -45 $O/by_shape_dp.txt
