"""Timeline of the last full steps in a rocprofv3 --kernel-trace csv (development tool): wall time of a step, GPU-busy time
(union of all kernels), time with two kernels' queues busy at once, idle gaps, and per kernel name the time on the critical
union - where a small-batch step spends its wall time when two streams share the chip.

  python tools/timeline.py <dir> [steps=3] [marker substring = adamw_kernel]
A step = from the end of one marker kernel (the fused optimizer pass: one per step) to the end of the next."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marker = sys.argv[3] if len(sys.argv) > 3 else "adamw_kernel"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
marks = [e for s, e, n, q in rows if marker in n]
if "--per-step" in sys.argv:
    # one line per step of the whole trace: wall, union-busy time, time of the longest kernels - to see where steps of different
    # loops of one run (bench.py: resident inputs, host->device copies, uint8 pipeline) differ
    import bisect
    starts = [r[0] for r in rows]
    for i in range(1, len(marks)):
        a, b = marks[i - 1], marks[i]
        win_ = rows[bisect.bisect_left(starts, a):bisect.bisect_left(starts, b)]
        win_ = [(s, min(e, b), n, q) for s, e, n, q in win_ if e > a]
        ev_ = sorted([(s, 1) for s, e, n, q in win_] + [(e, -1) for s, e, n, q in win_])
        busy_, depth_, last_ = 0, 0, a
        for t, k in ev_:
            if depth_ >= 1:
                busy_ += t - last_
            depth_ += k; last_ = t
        byq = collections.defaultdict(float)
        for s, e, n, q in win_:
            byq[q] += e - s
        extra = {n_[:40]: round(sum(e - s for s, e, n, q in win_ if n_ in n) / 1e3, 1) for n_ in ("resize_", "im2col", "copyBuffer", "elementwise")}
        print(f"step {i:3d}: wall {(b - a) / 1e3:9.1f} us  busy {busy_ / 1e3:9.1f}  idle {(b - a - busy_) / 1e3:7.1f}  launches {len(win_):4d}  per queue "
              f"{ {q: round(v / 1e3, 1) for q, v in byq.items()} }  {extra}")
    sys.exit(0)
if len(marks) < nsteps + 1:
    sys.exit(f"only {len(marks)} marker kernels")
t0, t1 = marks[-nsteps - 1], marks[-1]
win = [(s, e, n, q) for s, e, n, q in rows if s >= t0 and e <= t1]
wall = (t1 - t0) / nsteps
# union / overlap by a sweep over the start and end events
ev = []
for s, e, n, q in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0
depth, last = 0, t0
for t, k in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += k
    last = t
tot = sum(e - s for s, e, n, q in win)
print(f"steps {nsteps}: wall {wall / 1e3:.1f} us/step, busy (union) {busy / nsteps / 1e3:.1f}, idle {(t1 - t0 - busy) / nsteps / 1e3:.1f}, "
      f">= 2 kernels at once {over / nsteps / 1e3:.1f}, sum of kernel durations {tot / nsteps / 1e3:.1f}, launches {len(win) / nsteps:.0f}")
perq = collections.defaultdict(float)
for s, e, n, q in win:
    perq[q] += e - s
print("per queue (us/step):", {q: round(v / nsteps / 1e3, 1) for q, v in perq.items()})
# gaps on the union timeline by size class
gaps = []
depth, last = 0, t0
for t, k in ev:
    if depth == 0 and t > last:
        gaps.append(t - last)
    depth += k
    last = t
cls = collections.Counter()
for g in gaps:
    cls["<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else ">=20us"] += g
print("idle by gap size (us/step):", {k: round(v / nsteps / 1e3, 1) for k, v in cls.items()}, "gaps/step", len(gaps) // nsteps)
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in win:
    agg[n][0] += 1; agg[n][1] += e - s
print("%-100s %7s %10s %9s" % ("kernel", "calls", "us/step", "avg_us"))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-100s %7.1f %10.1f %9.2f" % (n[:100], c / nsteps, t / nsteps / 1e3, t / c / 1e3))
