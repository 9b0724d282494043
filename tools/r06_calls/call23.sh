#!/bin/bash
# Round 6, call 23: per-step timeline of one bench.py run with the new host->device loop (as call 10).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06c23; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o run -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-other-configs > $O/bench_traced.log 2>&1
python tools/timeline.py $O/tr 3 adamw_kernel --per-step > $O/per_step.txt 2>&1; cat $O/per_step.txt | cut -c1-250
python - <<PY
import csv, glob, collections
rows=[]
for f in glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks=[e for s,e,n in rows if "adamw_kernel" in n]
def table(i0,i1):
    agg=collections.defaultdict(float)
    for s,e,n in rows:
        if marks[i0] <= s and e <= marks[i1]: agg[n[:70]] += (e-s)/1e3/(i1-i0)
    return agg
# steps: 1..9 resident patches (2 warmup + 8 - 1), then f32 pixels (max(4, 4) = 4 steps + 2 untimed), then h2d (1 untimed + 8)
n=len(marks); print("marker kernels:", n)
a=table(12,15); b=table(n-12-8, n-12-3) if n > 30 else None
PY
rm -rf $O/tr
