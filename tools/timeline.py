"""Timeline view of a rocprofv3 --kernel-trace CSV: over the LAST `frac` of the trace (steady state), per-queue busy time, the union
of busy intervals (time at least one kernel runs), the span, and the kernels by total duration.
usage: timeline.py <kernel_trace.csv> [frac=0.5] [top=40]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = list(csv.DictReader(open(path)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows]
ks.sort()
t0, t1 = ks[0][0], max(k[1] for k in ks)
cut = t1 - (t1 - t0) * frac
ks = [k for k in ks if k[0] >= cut]
span = max(k[1] for k in ks) - ks[0][0]
per_q = defaultdict(int)
for s, e, n, q in ks:
    per_q[q] += e - s
# union of intervals
busy, cur_s, cur_e = 0, None, None
for s, e, n, q in ks:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"window {span / 1e6:.3f} ms, {len(ks)} kernels; some kernel running {busy / 1e6:.3f} ms ({100.0 * busy / span:.1f} %), idle {(span - busy) / 1e6:.3f} ms")
for q, t in sorted(per_q.items(), key=lambda kv: -kv[1]):
    print(f"  queue {q}: busy {t / 1e6:.3f} ms ({100.0 * t / span:.1f} % of the window)")
agg = defaultdict(lambda: [0, 0])
for s, e, n, q in ks:
    name = n.replace("void ", "").split("(")[0][:70]
    agg[name][0] += 1
    agg[name][1] += e - s
print("| kernel | calls | total ms | avg us | % of window |")
for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"| {name} | {c} | {t / 1e6:.3f} | {t / c / 1e3:.1f} | {100.0 * t / span:.1f} |")
