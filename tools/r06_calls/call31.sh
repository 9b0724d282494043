#!/bin/bash
# Round 6, call 31: HBM counters of the image-preprocessing kernels (one --pmc pass per counter; tools/micro/preprocess_ablation.py: the one-launch kernel, 256 images of 480 x 480 -> the 16-bit patch unfold)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c31; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -o run -- python3 tools/micro/preprocess_ablation.py > $O/pmc_$C.log 2>&1
  python tools/pmc_summary.py $O/pmc_$C $C > $O/pmc_$C.txt; grep "resize_" $O/pmc_$C.txt
  rm -rf $O/pmc_$C
done
