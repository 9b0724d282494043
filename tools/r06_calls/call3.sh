#!/bin/bash
# Round 6, call 3: split-K tests again, whole GPU suite, attention backward ablations, split-K in the step (same-process A/B), traces.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c3; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_gemm.py -x -q -k "split_k" > $O/test_splitk.log 2>&1; echo "split-k tests rc=$?"; tail -3 $O/test_splitk.log
for i in 1 2; do for L in tree attn_abl1 attn_abl3 attn_abl4 attn_abl5 attn_abl6; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  echo "== $L"; VAULT_HIP_LIB=$P timeout -k 10 120 python tools/attn_bench.py 256 185 2>&1 | grep -v amdgpu | grep "head-major"
done; done > $O/attn_ablate.txt 2>&1; cat $O/attn_ablate.txt
timeout -k 10 200 python tools/ab_attr.py SPLITK True,False 32 3 2>&1 | grep -v amdgpu > $O/ab_splitk_b32.txt; cat $O/ab_splitk_b32.txt
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/test_all.log 2>&1; echo "all tests rc=$?"; tail -8 $O/test_all.log
for B in 64 32; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t$B -o run -- python3 bench.py --batch $B --steps 6 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/t${B}_rocprof.log 2>&1
  python tools/timeline.py $O/t$B 3 > $O/timeline_b$B.txt 2>&1
  rm -rf $O/t$B
done
head -4 $O/timeline_b64.txt; head -4 $O/timeline_b32.txt
