"""Build profiles/<name>.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py (development tool).

  python tools/make_pmc_json.py <fetch_dir> <write_dir> <kernel substring[|substring...]> <out.json> <batch> <lm> "<description>"

Averages the counters over ALL launches of the kernel (like bench.py's live timing and rocprofv3 --stats do), and per
grid size.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-counts wide coalesced reads by 2x.
"""
import csv, glob, json, sys, collections
fd, wd, pat, out, batch, lm, desc = sys.argv[1:8]


def collect(d, counter):
    tot, n = 0.0, 0
    per = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter or not any(q in r["Kernel_Name"] for q in pat.split("|")):
                continue
            v = float(r["Counter_Value"])
            tot += v; n += 1
            g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
            per[g][0] += 1; per[g][1] += v
    return (tot / max(n, 1), n, {g: {"launches": c, "mean_KB": round(t / c, 1)} for g, (c, t) in per.items()})


f_mean, f_n, f_per = collect(fd, "FETCH_SIZE")
w_mean, w_n, w_per = collect(wd, "WRITE_SIZE")
res = {
    "kernel": desc, "batch": int(batch), "lm": lm,
    "command": "rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-parity --no-h2d "
               f"--batch {batch} --no-cpu-baseline   (one pass per counter: FETCH_SIZE, WRITE_SIZE)",
    "launches": {"FETCH_SIZE": f_n, "WRITE_SIZE": w_n},
    "FETCH_SIZE_KB_mean_over_launches": round(f_mean, 1), "WRITE_SIZE_KB_mean_over_launches": round(w_mean, 1),
    "per_grid": {"FETCH_SIZE": f_per, "WRITE_SIZE": w_per},
    "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2 "
                  "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; both in KB",
    "traffic_bytes_per_launch_avg": int(round((2.0 * f_mean + w_mean) * 1024)),
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])
