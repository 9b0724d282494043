"""Summarise rocprofv3 --pmc csv output per (kernel, grid): mean counter value per launch (development tool)."""
import csv, glob, sys, collections, re
d, counter = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)[:60]
        key = (n, r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
print("%-62s %10s %7s %16s" % ("kernel", "grid", "calls", "mean_" + counter))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-62s %10s %7d %16.1f" % (k[0], k[1], c, t / c))
