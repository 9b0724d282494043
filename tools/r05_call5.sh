#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c5; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -4 $O/pytest.log
for L in tree gelu_old; do if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi; echo "== $L"; VAULT_HIP_LIB=$P python tools/traj_seeds.py 600 700 800 900 1000 1100 1200 1300 2>&1 | grep -v amdgpu.ids; done | tee $O/traj_seeds.txt
