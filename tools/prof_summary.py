"""Summarise a rocprofv3 --kernel-trace --stats csv dir into a short per-kernel table (development tool)."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not files:
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]; dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            agg[n][0] += 1; agg[n][1] += dt
    tot = sum(v[1] for v in agg.values())
    print("%-90s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-90s %8d %12.1f %10.2f %6.2f" % (n[:90], c, t / 1e3, t / 1e3 / c, 100 * t / tot))
else:
    for f in files:
        rows = list(csv.DictReader(open(f)))
        print(f)
        print("%-90s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
        for r in rows[:40]:
            print("%-90s %8s %12.1f %10.2f %6s" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                   float(r["AverageNs"]) / 1e3, r["Percentage"]))
