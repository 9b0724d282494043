#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c26; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/t_all.txt 2>&1
rc=$?; tail -5 $O/t_all.txt
exit $rc
