"""Group a rocprofv3 kernel-trace csv by (kernel, grid) -> calls / avg us (development tool)."""
import csv, glob, sys, collections, re
d = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)[:60]
        key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
        agg[key][0] += 1
        agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("%-62s %10s %4s %5s %7s %11s %9s %6s" % ("kernel", "gridX", "gZ", "wg", "calls", "total_us", "avg_us", "%"))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-62s %10s %4s %5s %7d %11.1f %9.2f %6.2f" % (k[0], k[1], k[2], k[3], c, t / 1e3, t / 1e3 / c, 100 * t / tot))
