"""MFMA-pipe utilisation and effective clock per kernel from one rocprofv3 --pmc pass (development tool):
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o run -- python3 bench.py ...
  python tools/pmc_mfma.py DIR
util = MFMA busy cycles / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8
XCDs, MI355X_MICROARCH.md); effective clock = kernel cycles / kernel wall time."""
import csv, glob, sys, collections, re
d = sys.argv[1]
rows = collections.defaultdict(dict)
meta = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            meta[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size", "?"))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for key, c in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", key[1]); n = re.sub(r"^void ", "", n)[:58]
    ns, grid = meta.get(key, (0, "?"))
    a = agg[(n, grid)]
    a[0] += 1; a[1] += c["SQ_VALU_MFMA_BUSY_CYCLES"]; a[2] += c["GRBM_GUI_ACTIVE"]; a[3] += ns
print("%-60s %9s %6s %10s %9s %8s" % ("kernel", "grid", "calls", "mfma_util", "clk_GHz", "avg_us"))
for (n, grid), (cnt, busy, gui, ns) in sorted(agg.items(), key=lambda kv: -kv[1][3])[:24]:
    cyc = gui / 8.0
    util = busy / (1024.0 * cyc) if cyc else 0.0
    clk = cyc / ns if ns else 0.0
    print("%-60s %9s %6d %9.1f%% %9.2f %8.1f" % (n, grid, cnt, 100 * util, clk, ns / cnt / 1e3))
