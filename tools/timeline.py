"""Timeline of one training step from a rocprofv3 --kernel-trace CSV (default streams mode: teacher pass and filter-gradient
GEMMs beside the main stream):
    python tools/timeline.py <..._kernel_trace.csv> [steps-from-the-end to analyse = 3]
Splits the trace at the adam_kernel launches (one per step) and reports, per step: span, the time at least one kernel was
running (union), idle time, per-queue busy time, the number of launches and of gaps > 2 us on the union, and the kernels
with the largest total time.  Answers "is the step bound by kernel time or by gaps / dependencies between kernels?"."""
import csv, re, sys
from collections import defaultdict

path = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void suhpe::", "").replace("suhpe::", "")))
rows.sort()
cuts = [i for i, r in enumerate(rows) if r[3].startswith("adam_kernel")]
print(f"{len(rows)} dispatches, {len(cuts)} steps")
for s in range(max(1, len(cuts) - last), len(cuts)):
    seg = rows[cuts[s - 1] + 1: cuts[s] + 1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    ev = sorted((a, b) for a, b, _, _ in seg)
    busy, gaps, cur_a, cur_b, big = 0, 0, ev[0][0], ev[0][1], 0
    for a, b in ev[1:]:
        if a > cur_b:
            busy += cur_b - cur_a
            if a - cur_b > 2000:
                big += 1
            gaps += a - cur_b
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    perq = defaultdict(int)
    for a, b, q, _ in seg:
        perq[q] += b - a
    print(f"step {s}: span {(t1 - t0) / 1e6:.2f} ms, union busy {busy / 1e6:.2f} ms, idle {gaps / 1e6:.2f} ms ({big} gaps > 2 us), launches {len(seg)}, "
          f"sum of kernel time {sum(b - a for a, b, _, _ in seg) / 1e6:.2f} ms, per queue " + ", ".join(f"q{q}: {v / 1e6:.2f}" for q, v in sorted(perq.items())))
seg = rows[cuts[-2] + 1: cuts[-1] + 1]
agg = defaultdict(lambda: [0, 0])
for a, b, q, name in seg:
    agg[name][0] += b - a
    agg[name][1] += 1
for name, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"  {t / 1e6:7.3f} ms  x{c:4d}  {name[:110]}")
