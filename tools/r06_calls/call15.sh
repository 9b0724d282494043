#!/bin/bash
# Round 6, call 15: AdamW without stores for never-touched elements: test, trajectory tests, same-box comparison with round 5's tree
# at B = 32 / 64 / 256 (final tree), kernel time of the optimizer pass in the step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c15; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q > $O/test_train.log 2>&1; echo "train tests rc=$?"; tail -3 $O/test_train.log
for i in 1 2; do
  for what in "--batch 32" "--config 2" ""; do
    tag=$(echo "b256 $what" | sed 's/b256 --batch 32/b32/; s/b256 --config 2/b64/; s/ //g')
    (cd _r05 && python bench.py $what --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-h2d 2>/dev/null | tail -1 > $O/r05_${tag}_$i.json)
    python bench.py $what --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-h2d 2>/dev/null | tail -1 > $O/r06_${tag}_$i.json
    python - <<PY
import json
a=json.load(open("$O/r05_${tag}_$i.json")); b=json.load(open("$O/r06_${tag}_$i.json"))
print("$tag round $i: r05 %.1f  r06 %.1f samples/s  (%+.2f %%)   vilt %s -> %s  lm %s -> %s" % (a["value"], b["value"], 100*(b["value"]/a["value"]-1), a.get("vilt_block_frac"), b.get("vilt_block_frac"), a.get("lm_block_frac"), b.get("lm_block_frac")))
PY
  done
done 2>&1 | tee $O/r05_vs_r06_same_box_final.txt
for B in 32 256; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t$B -o run -- python3 bench.py --batch $B --steps 25 --warmup 5 --no-cpu-baseline --no-parity --no-h2d --no-other-configs > $O/t$B.log 2>&1
  python tools/prof_seq.py $O/t$B adamw_kernel 30 > $O/adamw_seq_b$B.txt; cat $O/adamw_seq_b$B.txt; rm -rf $O/t$B
done
