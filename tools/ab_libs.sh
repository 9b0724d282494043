#!/bin/bash
# Run on the GPU box: the same command under several builds of the library (same box, interleaved).
# usage: ab_libs.sh "lib1 lib2 ..." reps cmd...      (lib = name under build_ab/ without prefix / suffix, or "tree")
cd "$GRAFT_REPO_ROOT"
LIBS=$1; REPS=$2; shift 2
for i in $(seq $REPS); do
  for L in $LIBS; do
    if [ "$L" = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
    echo "== $L"
    VAULT_HIP_LIB=$P "$@" 2>&1 | grep -v amdgpu.ids
  done
done
