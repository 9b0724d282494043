#!/bin/bash
# Round 6, call 10: new LayerNorm backward test + LM kernels; per-step timeline of the three bench loops (resident inputs, host->device
# copies, uint8 pipeline) in ONE run - where the input pipeline's 2 % goes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c10; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q > $O/test_ops.log 2>&1; echo "ops tests rc=$?"; tail -3 $O/test_ops.log
timeout -k 10 300 python tools/ln_bench.py > $O/ln_bench.txt 2>&1; grep -v amdgpu $O/ln_bench.txt | tail -12
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o run -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-other-configs > $O/bench_traced.log 2>&1
python tools/timeline.py $O/tr 3 adamw_kernel --per-step > $O/per_step.txt 2>&1; cat $O/per_step.txt | cut -c1-260
ls $O/tr/*/ | head; python - <<PY
import csv, glob
for f in glob.glob("$O/tr/**/*memory_copy_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows))
    import collections
    agg=collections.defaultdict(lambda:[0,0,0.0])
    for r in rows:
        k=(r.get("Direction") or r.get("Kind") or "?")
        agg[k][0]+=1; agg[k][1]+=int(r.get("Size",0) or 0); agg[k][2]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    for k,v in agg.items(): print(k, v[0], "copies", round(v[1]/1e6,1), "MB", round(v[2]/1e6,2), "ms", round(v[1]/max(v[2],1),2), "GB/s")
PY
rm -rf $O/tr
