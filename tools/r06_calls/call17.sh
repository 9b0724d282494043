#!/bin/bash
# Round 6, call 17: which share of the optimizer's elements is idle in the step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python tools/micro/adamw_idle_probe.py 32 2>&1 | grep -v amdgpu
