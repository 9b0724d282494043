#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c17; mkdir -p $O
python tools/ab_attr.py HEAD_MAJOR_MIN_ROWS 16384,8192 256 3 2>&1 | grep -v amdgpu | tee $O/hm.txt
python tools/ab_attr.py LM_BIAS_PARTIALS False,True 256 3 2>&1 | grep -v amdgpu | tee $O/lmpart.txt
