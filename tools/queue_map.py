"""Which hardware queue does each stream's work land on, and what sits in front of a late kernel?  From a rocprofv3 --kernel-trace CSV of
the bench step: per Queue_Id the kernels it carried in ONE step window (names, first start / last end relative to the window), and for
the kernels matching NEEDLE (default: the text tower's first backward kernels are hard to name, so: every queue's timeline of big
events).  usage: queue_map.py <kernel_trace.csv> [step index from the end = 2]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0][:60], r["Queue_Id"]) for r in rows)
ends = [k[1] for k in ks if "adamw_kernel" in k[2]]
per = 3
marks = ends[per - 1::per]
lo, hi = marks[-back - 1], marks[-back]
win = [k for k in ks if k[0] >= lo and k[1] <= hi]
print(f"step window {(hi - lo) / 1e6:.2f} ms, {len(win)} kernels")
byq = defaultdict(list)
for k in win:
    byq[k[3]].append(k)
for q, lst in sorted(byq.items()):
    busy = sum(e - s for s, e, _, _ in lst)
    print(f"\n== queue {q}: {len(lst)} kernels, busy {busy / 1e6:.2f} ms, first start {(lst[0][0] - lo) / 1e6:.2f} ms, last end {(lst[-1][1] - lo) / 1e6:.2f} ms")
    # compress into runs of the same kernel family
    runs = []
    for s, e, n, _ in lst:
        fam = n.split("<")[0]
        if runs and runs[-1][0] == fam and s - runs[-1][2] < 200000:
            runs[-1][2] = e
            runs[-1][3] += 1
        else:
            runs.append([fam, s, e, 1])
    # print the runs that are long or preceded by a long gap on this queue
    prev_end = lo
    for fam, s, e, c in runs:
        gap = (s - prev_end) / 1e6
        if gap > 0.5 or (e - s) > 300000 or "oneRank" in fam or "ccl" in fam.lower():
            print(f"   {(s - lo) / 1e6:7.2f} -> {(e - lo) / 1e6:7.2f} ms  x{c:3d} {fam}   (queue idle {gap:.2f} ms before)")
        prev_end = e
