#!/bin/bash
# Run on the GPU box: LDS / wait counters of one bench step per kernel (development: what stalls the GEMM main loops)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_lds; mkdir -p $O
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  T=$(echo $SET | tr ' ' '_')
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/$T -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-h2d > $O/$T.log 2>&1
  for C in $SET; do python tools/pmc_summary.py $O/$T $C > $O/$C.txt; done
  rm -rf $O/$T
done
