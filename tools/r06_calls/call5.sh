#!/bin/bash
# Round 6, call 5: phase-1 unroll of the attention backward, prefetch distance of the ring kernel's residual epilogue,
# the GEMM scheduler alone (ops.GEMM_SCHED 0 / 3), the data-parallel code path on one rank.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c5; mkdir -p $O
for i in 1 2; do for L in tree attn_u2 attn_u3 attn_u6; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  echo "== $L"; VAULT_HIP_LIB=$P timeout -k 10 120 python tools/attn_bench.py 256 185 2>&1 | grep -v amdgpu | grep head-major
done; done > $O/attn_unroll.txt 2>&1; cat $O/attn_unroll.txt
for i in 1 2; do for L in tree epd3 epd4; do
  if [ $L = tree ]; then P=$GRAFT_REPO_ROOT/vault_amd/libvault_hip.so; else P=$GRAFT_REPO_ROOT/build_ab/libvault_hip_$L.so; fi
  echo "== $L"; VAULT_HIP_LIB=$P timeout -k 10 200 python tools/pf_bench.py 47360 res,lm 2>&1 | grep -v amdgpu | grep "fwd"
done; done > $O/epi_pd.txt 2>&1; cat $O/epi_pd.txt
timeout -k 10 300 python tools/ab_sched.py 256 2 2>&1 | grep -v amdgpu > $O/ab_sched.txt; cat $O/ab_sched.txt
python bench.py --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_default.json 2> $O/bench_default.err
VAULT_FORCE_DP=1 python bench.py --no-cpu-baseline --no-other-configs --no-h2d --no-parity > $O/bench_dp1_fp32_wire.json 2> $O/bench_dp1.err
python bench.py --no-cpu-baseline --no-other-configs --no-parity --no-h2d > $O/bench_default2.json 2> $O/bench_default2.err
for f in bench_default bench_dp1_fp32_wire bench_default2; do cut -c1-200 $O/$f.json; echo; done
