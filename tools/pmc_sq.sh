# SQ counters (two PMC passes) + kernel durations of the kernels whose name contains FILTER, for any command.
# usage (on the GPU box): [ENV=...] bash tools/pmc_sq.sh FILTER python3 tools/some_bench.py args > gpurun_out/x.txt
# (the program itself follows the filter: no env / bash -c hop between rocprofv3 and python3)
FILTER=$1; shift
R=$GRAFT_REPO_ROOT
CMD=()
for a in "$@"; do case "$a" in tools/*|bench.py) CMD+=("$R/$a");; *) CMD+=("$a");; esac; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq_a /tmp/pmc_sq_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_sq_a -- "${CMD[@]}" > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/pmc_sq_b -- "${CMD[@]}" > /dev/null 2>&1
cd $R
FILTER="$FILTER" python3 - <<'PY'
import csv, glob, collections, os
filt = os.environ["FILTER"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
dur = collections.defaultdict(list)
for d in ("a", "b"):
    for f in glob.glob(f"/tmp/pmc_sq_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if filt not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"): n[(k, r["Counter_Name"])] += 1
    for f in glob.glob(f"/tmp/pmc_sq_{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if filt in k: dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, c in acc.items():
    la = max(n[(k, "SQ_WAVE_CYCLES")], 1)
    d = sorted(dur[k])
    print(f"== {k}  launches {la}  median {d[len(d) // 2]:.1f} us (profiled)")
    wc = c["SQ_WAVE_CYCLES"]
    for name in sorted(c):
        extra = f"  ({c[name] / wc:.3f} of wave cycles)" if name.startswith("SQ_WAIT") or name.startswith("SQ_ACTIVE") else ""
        print(f"   {name:28s} {c[name] / la:14.0f} per launch{extra}")
PY
