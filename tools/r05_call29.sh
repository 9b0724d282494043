#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c29; mkdir -p $O
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 || { tail -5 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/t_all.txt 2>&1
rc=$?; tail -3 $O/t_all.txt
exit $rc
