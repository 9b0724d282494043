#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c2; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -15 $O/pytest.log
