# FETCH_SIZE / WRITE_SIZE per kernel of any command (two PMC passes): bash tools/pmc_fetch_one.sh FILTER python3 tools/x.py args
FILTER=$1; shift
R=$GRAFT_REPO_ROOT
CMD=()
for a in "$@"; do case "$a" in tools/*|bench.py) CMD+=("$R/$a");; *) CMD+=("$a");; esac; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_a /tmp/pf_b
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_a -- "${CMD[@]}" > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pf_b -- "${CMD[@]}" > /dev/null 2>&1
cd $R
FILTER="$FILTER" python3 - <<'PY'
import csv, glob, collections, os
filt = os.environ["FILTER"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
for d in ("a", "b"):
    for f in glob.glob(f"/tmp/pf_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if filt not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
for k, c in acc.items():
    lf, lw = max(n[k]["FETCH_SIZE"], 1), max(n[k]["WRITE_SIZE"], 1)
    # (the counters are in KB; reads: 2 x FETCH_SIZE on gfx950 for wide coalesced loads, MI355X_MICROARCH.md)
    print(f"{k}: launches {lf}  read beyond L2 per launch {2 * c['FETCH_SIZE'] / lf * 1024 / 1e6:.1f} MB (2 x FETCH_SIZE)  written {c['WRITE_SIZE'] / lw * 1024 / 1e6:.1f} MB")
PY
