#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c25; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o run -- python3 tools/micro/steps_attr.py fp8 FFN_OUT_FP8=True > $O/log.txt 2>&1 || exit 1
python tools/prof_by_shape.py $O/st > $O/by_shape_ffn_out.txt 2>&1
rm -rf $O/st
head -30 $O/by_shape_ffn_out.txt
