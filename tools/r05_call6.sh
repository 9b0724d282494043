#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c6; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -6 $O/pytest.log
