#!/bin/bash
# Round 6, call 8: per-kernel comparison of round 5's tree (_r05) and this tree at B = 256 and config 4 on ONE box (kernel traces).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c8; mkdir -p $O
for what in "" "--config 4"; do
  tag=$(echo "b256$what" | sed 's/b256--config 4/cfg4/')
  (cd _r05 && rocprofv3 --kernel-trace --output-format csv -d $O/t05$tag -o run -- python3 bench.py $what --steps 6 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/t05${tag}.log 2>&1)
  python tools/timeline.py $O/t05$tag 3 > $O/timeline_r05_$tag.txt 2>&1; rm -rf $O/t05$tag
  rocprofv3 --kernel-trace --output-format csv -d $O/t06$tag -o run -- python3 bench.py $what --steps 6 --warmup 3 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/t06${tag}.log 2>&1
  python tools/timeline.py $O/t06$tag 3 > $O/timeline_r06_$tag.txt 2>&1; rm -rf $O/t06$tag
done
head -3 $O/timeline_r05_b256.txt $O/timeline_r06_b256.txt $O/timeline_r05_cfg4.txt $O/timeline_r06_cfg4.txt | cut -c1-200
