# SQ counters of every kernel whose name contains FILTER, for any bench script; three PMC passes, sums per kernel and launch.
# usage (on the GPU box): bash tools/pmc_kernels.sh FILTER tools/some_bench.py [args...] > gpurun_out/pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=$1; shift
S=$R/$1; shift
rm -rf $R/gpurun_out/pmc_k_a $R/gpurun_out/pmc_k_b $R/gpurun_out/pmc_k_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k_a -- python3 $S "$@" > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k_b -- python3 $S "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k_c -- python3 $S "$@" > /dev/null 2>&1
cd $R
FILTER=$F python3 - <<'PY'
import csv, glob, collections, os
flt = os.environ["FILTER"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for d in ("a", "b", "c"):
    for f in glob.glob(f"gpurun_out/pmc_k_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if flt not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"): n[(k, r["Counter_Name"])] += 1
for k, c in acc.items():
    la = max(n[(k, "SQ_WAVE_CYCLES")], 1)
    print(f"== {k}  launches {la}")
    wc = c["SQ_WAVE_CYCLES"]
    for name in sorted(c):
        extra = f"  ({c[name] / wc:.3f} of wave cycles)" if name.startswith("SQ_WAIT") or name.startswith("SQ_ACTIVE") or name.startswith("SQ_INST_") else ""
        print(f"   {name:28s} {c[name] / la:14.0f} per launch{extra}")
PY
find gpurun_out/pmc_k_a gpurun_out/pmc_k_b gpurun_out/pmc_k_c -type f -delete
