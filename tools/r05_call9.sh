#!/bin/bash
# round 5: in-kernel clocks of the three long kernels, their perfect-memory ablation times (one box), then the evidence set
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c9; mkdir -p $O
VAULT_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/libvault_hip_stamp.so python tools/clock_stamp.py 2>&1 | grep -v amdgpu.ids | tee $O/clock_stamp.txt
bash tools/ab_libs.sh "tree abl" 2 python tools/pf_bench.py 47360 dgrad,res,wgrad 2>&1 | tee $O/ring_ablation.txt
RASTER_BENCH_GNS=0 RASTER_BENCH_KIND=8w bash tools/ab_libs.sh "tree abl" 2 python tools/raster_bench.py 47360 2>&1 | tee $O/w8_ablation.txt
bash tools/collect_profiles.sh > $O/collect.log 2>&1; echo "collect rc=$?"
tail -3 gpurun_out/final/bench_default.json | cut -c1-600
