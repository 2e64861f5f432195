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
import os
step_kernel = os.environ.get("STEP_KERNEL")   # e.g. STEP_KERNEL=adamw_kernel:3:4 -> the window is ONE step: from the end of the
# 3rd-launch-per-step kernel of step 3 to its end in step 4 (the optimizer's last launch closes a step)
if step_kernel:
    name, per, idx = step_kernel.split(":")
    ends = [k[1] for k in ks if name in k[2]]
    marks = ends[int(per) - 1::int(per)]
    lo, hi = marks[int(idx) - 1], marks[int(idx)]
    ks = [k for k in ks if k[0] >= lo and k[1] <= hi]
else:
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

# gaps on the busiest queue: where the main chain waits (for another queue's event, or for the host)
main_q = max(per_q.items(), key=lambda kv: kv[1])[0]
mk = [k for k in ks if k[3] == main_q]
gaps = []
for a, b in zip(mk[:-1], mk[1:]):
    g = b[0] - a[1]
    if g > 0:
        gaps.append((g, a[2].replace("void ", "").split("(")[0][:48], b[2].replace("void ", "").split("(")[0][:48]))
tot = sum(g[0] for g in gaps)
print(f"\nqueue {main_q}: {len(mk)} kernels, {tot / 1e6:.3f} ms of gaps between consecutive kernels ({sum(1 for g in gaps if g[0] > 5000)} gaps > 5 us = "
      f"{sum(g[0] for g in gaps if g[0] > 5000) / 1e6:.3f} ms)")
pair = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    pair[(a, b)][0] += 1
    pair[(a, b)][1] += g
print("| after | before | count | total us | avg us |")
for (a, b), (c, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"| {a} | {b} | {c} | {t / 1e3:.1f} | {t / c / 1e3:.2f} |")

# GANTT=<ms>: the window in buckets of that many ms -- per queue the busy share of the bucket and the kernel that took most of it
if os.environ.get("GANTT"):
    step = float(os.environ["GANTT"]) * 1e6
    lo = ks[0][0]
    nb = int(span / step) + 1
    qs = sorted(per_q, key=lambda q: -per_q[q])
    table = {q: [defaultdict(int) for _ in range(nb)] for q in qs}
    for s, e, n, q in ks:
        name = n.replace("void ", "").split("(")[0].split("<")[0][:22]
        b0, b1 = int((s - lo) / step), int((e - lo) / step)
        for b in range(b0, min(b1, nb - 1) + 1):
            a, z = max(s, lo + b * step), min(e, lo + (b + 1) * step)
            if z > a:
                table[q][b][name] += z - a
    print("\n| ms | " + " | ".join(f"queue {q}" for q in qs) + " |")
    for b in range(nb):
        cells = []
        for q in qs:
            d = table[q][b]
            if not d:
                cells.append("")
                continue
            tot_b = sum(d.values())
            name = max(d.items(), key=lambda kv: kv[1])[0]
            cells.append(f"{100.0 * tot_b / step:3.0f}% {name}")
        print(f"| {b * step / 1e6:5.1f} | " + " | ".join(cells) + " |")

# WAKE=<ms>: every kernel that starts a queue's work after that queue sat idle for at least <ms>, with the kernels that ended last on
# the other queues before it -- what the queue was waiting for
if os.environ.get("WAKE"):
    thr = float(os.environ["WAKE"]) * 1e6
    lo = ks[0][0]
    last_end = {}
    hist = []
    print("\n| queue | wakes at ms | idle ms | first kernel | ended just before (queue: kernel @ ms) |")
    for s, e, n, q in ks:
        if q in last_end and s - last_end[q] >= thr:
            before = sorted([(ee, qq, nn) for ss, ee, nn, qq in hist if qq != q and ee <= s + 2000], reverse=True)[:3]
            desc = "; ".join(f"{qq}: {nn.replace('void ', '').split('(')[0][:28]} @ {(ee - lo) / 1e6:.3f}" for ee, qq, nn in before)
            print(f"| {q} | {(s - lo) / 1e6:.3f} | {(s - last_end[q]) / 1e6:.2f} | {n.replace('void ', '').split('(')[0][:36]} | {desc} |")
        last_end[q] = max(last_end.get(q, 0), e)
        hist.append((s, e, n, q))
        if len(hist) > 400:
            hist = hist[-200:]

# LIST=<from_ms>:<to_ms>: every kernel that starts in that part of the window (queue, start, duration)
if os.environ.get("LIST"):
    a_ms, b_ms = (float(v) for v in os.environ["LIST"].split(":"))
    lo = ks[0][0]
    print(f"\n| queue | start ms | us | kernel |   ({a_ms} .. {b_ms} ms of the window)")
    for s, e, n, q in ks:
        t = (s - lo) / 1e6
        if a_ms <= t < b_ms:
            print(f"| {q} | {t:7.3f} | {(e - s) / 1e3:7.1f} | {n.replace('void ', '')[:90]} |")
