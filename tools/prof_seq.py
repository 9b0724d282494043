"""Durations of every launch of one kernel (name substring) in launch order, for the LAST `n` launches of a rocprofv3
kernel-trace csv (development tool): python tools/prof_seq.py <dir> <substring> [n]"""
import csv, glob, sys
d, sub = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
rows.sort()
print(" ".join(f"{t / 1e3:.0f}({g})" for _, t, g in rows[-n:]))
print("sum of the last", n, ":", round(sum(t for _, t, _ in rows[-n:]) / 1e3, 1), "us")
