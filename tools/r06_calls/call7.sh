#!/bin/bash
# Round 6, call 7: round 5's tree (_r05: git worktree of d3939e5, built) against this tree on ONE box, interleaved: B = 32, 64, 128
# (config 4), 256; the tail / split-K micro-benchmark; new ring-tail test.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c7; mkdir -p $O
timeout -k 10 200 python -m pytest tests/test_gpu_gemm.py -x -q -k "tail_rows" > $O/test_tail.log 2>&1; echo "tail tests rc=$?"; tail -3 $O/test_tail.log
for i in 1 2; do
  for what in "--batch 32" "--config 2" "--config 4" ""; do
    tag=$(echo "b256 $what" | sed 's/b256 --batch 32/b32/; s/b256 --config 2/b64/; s/b256 --config 4/cfg4/; s/ //g')
    (cd _r05 && python bench.py $what --steps 15 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-h2d 2>/dev/null | tail -1 > $O/r05_${tag}_$i.json)
    python bench.py $what --steps 15 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-h2d 2>/dev/null | tail -1 > $O/r06_${tag}_$i.json
    python - <<PY
import json
a=json.load(open("$O/r05_${tag}_$i.json")); b=json.load(open("$O/r06_${tag}_$i.json"))
print("$tag round $i: r05 %.1f  r06 %.1f samples/s  (%+.2f %%)   vilt %s -> %s  lm %s -> %s" % (a["value"], b["value"], 100*(b["value"]/a["value"]-1), a.get("vilt_block_frac"), b.get("vilt_block_frac"), a.get("lm_block_frac"), b.get("lm_block_frac")))
PY
  done
done 2>&1 | tee $O/r05_vs_r06_same_box.txt
timeout -k 10 300 python tools/splitk_bench.py 6144 23808 35584 2>&1 | grep -v amdgpu > $O/splitk_tail_bench.txt; cat $O/splitk_tail_bench.txt
