"""Average one PMC counter (and the kernel durations) per kernel over a rocprofv3 --pmc --kernel-trace output directory.
usage: pmc_one.py <dir> <counter> [name filter]"""
import collections, csv, glob, sys

d, counter = sys.argv[1], sys.argv[2]
filt = sys.argv[3] if len(sys.argv) > 3 else ""
tot, n = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and filt in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
dur, dn = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if filt in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            dn[k] += 1
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f"{k[:60]:60s} launches {n[k]:4d}  {counter}/launch {tot[k] / n[k]:12.1f}  avg us {dur[k] / max(dn[k], 1) / 1e3:8.1f}")
